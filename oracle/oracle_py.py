"""ctypes binding of the CPU ORACLE (oracle/libygz_oracle.so) and of the reference's libfast (oracle/_ref).

TEST INFRASTRUCTURE ONLY: imported by tests/, by bench.py's cpu_baseline leg and by __graft_entry__.smoke().
The product package (orb_ygz_slam_amd) never imports this module.
"""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))


class KeyPoint(C.Structure):  # cv::KeyPoint layout, 28 bytes
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("size", C.c_float), ("angle", C.c_float),
                ("response", C.c_float), ("octave", C.c_int), ("class_id", C.c_int)]


KP_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"), ("response", "<f4"),
                     ("octave", "<i4"), ("class_id", "<i4")])


def build(force=False):
    so = os.path.join(_HERE, "libygz_oracle.so")
    if force or not os.path.exists(so) or not os.path.exists(os.path.join(_HERE, "_ref", "libfast_ref.so")):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return so


_lib = None
_ref = None

# OpenCV arithmetic mode of the 8-bit GaussianBlur (oracle_cvprims.cpp): process-wide, applied to the oracle and to every oracle/_ref
# build of the reference's own sources (their cv::GaussianBlur is the same restatement)
CV_MODE_LEGACY_SSE2, CV_MODE_LEGACY_INT, CV_MODE_CV4 = 0, 1, 2
_cv_mode = CV_MODE_LEGACY_SSE2
_cv_libs = []


def _track_cv(L):
    if L is not None and hasattr(L, "yo_set_cv_mode") and all(L is not x for x in _cv_libs):
        _cv_libs.append(L)
        L.yo_set_cv_mode(_cv_mode)
    return L


def set_cv_mode(mode):
    global _cv_mode
    _cv_mode = int(mode)
    for L in _cv_libs:
        L.yo_set_cv_mode(_cv_mode)


class cv_mode:
    """with cv_mode(CV_MODE_CV4): ... -- the oracle (and the reference builds) blur like that OpenCV generation inside the block."""

    def __init__(self, mode):
        self.mode = mode

    def __enter__(self):
        self.prev = _cv_mode
        set_cv_mode(self.mode)
        return self

    def __exit__(self, *a):
        set_cv_mode(self.prev)



_lib_override = None


def ref_matcher_lib():
    """The reference's own src/ORBmatcher.cc behind the oracle's flat matcher API (oracle/_ref/libref_orbmatcher.so), or None."""
    build()
    p = os.path.join(_HERE, "_ref", "libref_orbmatcher.so")
    return _track_cv(C.CDLL(p)) if os.path.exists(p) else None


def ref_mappoint_lib():
    """The reference's own src/MapPoint.cc (oracle/_ref/libref_mappoint.so) behind yo_distinctive_descriptors / yo_predict_scale, or None."""
    build()
    p = os.path.join(_HERE, "_ref", "libref_mappoint.so")
    return C.CDLL(p) if os.path.exists(p) else None


class reference_mappoint:
    """with reference_mappoint(): distinctive_descriptors() and predict_scale() run the reference's MapPoint code."""

    def __enter__(self):
        global _lib_override
        _lib_override = ref_mappoint_lib()
        assert _lib_override is not None, "oracle/_ref/libref_mappoint.so not built"
        return self

    def __exit__(self, *a):
        global _lib_override
        _lib_override = None


def ref_frame_lib():
    """The reference's own src/Frame.cc (oracle/_ref/libref_frame.so) behind yo_features_in_area / yo_is_in_frustum / yo_compute_stereo_matches."""
    build()
    p = os.path.join(_HERE, "_ref", "libref_frame.so")
    return _track_cv(C.CDLL(p)) if os.path.exists(p) else None


class reference_frame:
    """with reference_frame(): features_in_area() and is_in_frustum() run the reference's Frame code."""

    def __enter__(self):
        global _lib_override
        _lib_override = ref_frame_lib()
        assert _lib_override is not None, "oracle/_ref/libref_frame.so not built"
        return self

    def __exit__(self, *a):
        global _lib_override
        _lib_override = None


class reference_matcher:
    """with reference_matcher(): the matcher wrappers of this module (search_by_projection_*, search_for_initialization, search_by_bow)
    run the REFERENCE's code instead of the oracle's restatement -- same flat inputs, same outputs (a slot that was matched and then
    culled by the rotation check reads -1 there, -2 from the oracle: the reference leaves no trace of it)."""

    def __enter__(self):
        global _lib_override
        _lib_override = ref_matcher_lib()
        assert _lib_override is not None, "oracle/_ref/libref_orbmatcher.so not built"
        return self

    def __exit__(self, *a):
        global _lib_override
        _lib_override = None


def lib():
    global _lib
    if _lib_override is not None:
        return _lib_override
    if _lib is None:
        _lib = C.CDLL(os.environ.get("YGZ_ORACLE_LIB") or build())   # YGZ_ORACLE_LIB: the sanitizer build (tests/test_oracle_sanitizers.py)
        L = _lib
        _track_cv(L)
        L.yo_extractor_create.restype = C.c_void_p
        L.yo_extractor_create.argtypes = [C.c_int, C.c_float, C.c_int, C.c_int, C.c_int]
        L.yo_extractor_destroy.argtypes = [C.c_void_p]
        L.yo_fast_atan2.restype = C.c_float
        L.yo_fast_atan2.argtypes = [C.c_float, C.c_float]
        L.yo_sincos_deg.argtypes = [C.c_float, C.POINTER(C.c_float), C.POINTER(C.c_float)]
        L.yo_cv_round.argtypes = [C.c_double]
        L.yo_ic_angle.restype = C.c_float
        L.yo_ic_angle.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_float]
        L.yo_descriptor.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, C.c_void_p]
        L.yo_sparse_img_align.restype = C.c_size_t
    return _lib


def ref_fast():
    """The reference's own Thirdparty/fast, or None when oracle/_ref was never built (no reference checkout)."""
    global _ref
    if _ref is None:
        build()
        p = os.path.join(_HERE, "_ref", "libfast_ref.so")
        if not os.path.exists(p):
            return None
        _ref = C.CDLL(p)
    return _ref


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


_ref_ex = None


def ref_extractor_lib():
    """The reference's own src/ORBextractor.cc (oracle/_ref/libref_orbextractor.so, built by oracle/Makefile from the reference checkout
    against the OpenCV stand-in of oracle/ref_shim/), or None when it was never built."""
    global _ref_ex
    if _ref_ex is None:
        build()
        p = os.path.join(_HERE, "_ref", "libref_orbextractor.so")
        if not os.path.exists(p):
            return None
        _ref_ex = _track_cv(C.CDLL(p))
        _ref_ex.yr_extract.restype = C.c_int
        _ref_ex.yr_extract.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
        _ref_ex.yr_pyramid_level.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        _ref_ex.yr_frame_extractor_create.restype = C.c_void_p
        _ref_ex.yr_frame_extractor_create.argtypes = [C.c_int, C.c_float, C.c_int, C.c_int, C.c_int]
        _ref_ex.yr_frame_extractor_destroy.argtypes = [C.c_void_p]
        _ref_ex.yr_frame_extract.restype = C.c_int
        _ref_ex.yr_frame_extract.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
    return _ref_ex


def _kp7_to_struct(a):
    k = np.zeros(len(a), KP_DTYPE)
    for i, f in enumerate(("x", "y", "size", "angle", "response")):
        k[f] = a[:, i]
    k["octave"] = a[:, 5].astype(np.int32)
    k["class_id"] = a[:, 6].astype(np.int32)
    return k


def _struct_to_kp7(k):
    a = np.zeros((len(k), 7), np.float32)
    for i, f in enumerate(("x", "y", "size", "angle", "response", "octave", "class_id")):
        a[:, i] = k[f]
    return a


def ref_extract(img, nfeatures=1000, scale_factor=1.2, nlevels=8, ini_th=20, min_th=7, cap=60000):
    """ygz::ORBextractor(...)(image, mask, keypoints, descriptors) run by the REFERENCE's own code -> (keys, desc)."""
    L = ref_extractor_lib()
    img = np.ascontiguousarray(img, np.uint8)
    h, w = img.shape
    kp = np.zeros((cap, 7), np.float32)
    d = np.zeros((cap, 32), np.uint8)
    n = L.yr_extract(_p(img), w, h, w, nfeatures, scale_factor, nlevels, ini_th, min_th, _p(kp), _p(d), cap)
    assert n >= 0
    return _kp7_to_struct(kp[:n]), d[:n].copy()


def ref_pyramid(img, scale_factor=1.2, nlevels=8):
    L = ref_extractor_lib()
    img = np.ascontiguousarray(img, np.uint8)
    h, w = img.shape
    out = []
    for lvl in range(nlevels):
        lw, lh = C.c_int(), C.c_int()
        L.yr_pyramid_level(_p(img), w, h, w, scale_factor, nlevels, lvl, None, C.byref(lw), C.byref(lh))
        o = np.zeros((lh.value, lw.value), np.uint8)
        L.yr_pyramid_level(_p(img), w, h, w, scale_factor, nlevels, lvl, _p(o), C.byref(lw), C.byref(lh))
        out.append(o)
    return out


class RefFrameExtractor:
    """The reference's ORBextractor object kept across frames (mnGridSize of the DSO detector is state), driven through
    operator()(Frame*, keypoints, descriptors, method, leftEye = true).  method: 0 ORBSLAM_KEYPOINT, 2 DSO_KEYPOINT."""

    def __init__(self, nfeatures=1000, scale_factor=1.2, nlevels=8, ini_th=20, min_th=7):
        self.L = ref_extractor_lib()
        self.h = self.L.yr_frame_extractor_create(nfeatures, scale_factor, nlevels, ini_th, min_th)

    def close(self):
        if self.h:
            self.L.yr_frame_extractor_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def extract(self, img, method, existing=None, cap=60000):
        img = np.ascontiguousarray(img, np.uint8)
        h, w = img.shape
        ex = _struct_to_kp7(existing) if existing is not None and len(existing) else np.zeros((0, 7), np.float32)
        kp = np.zeros((cap, 7), np.float32)
        d = np.zeros((cap, 32), np.uint8)
        n = self.L.yr_frame_extract(self.h, _p(img), w, h, w, method, _p(ex) if len(ex) else None, len(ex), _p(kp), _p(d), cap)
        assert n >= 0
        return _kp7_to_struct(kp[:n]), d[:n].copy()

    def dso_multilevel(self, img, existing=None, cap=60000):
        """ComputeKeyPointsDSO (the multi-level grid detector) -> (existing keys with fresh angles, new keys in LEVEL coordinates, mnGridSize)."""
        img = np.ascontiguousarray(img, np.uint8)
        h, w = img.shape
        ex = _struct_to_kp7(existing) if existing is not None and len(existing) else np.zeros((0, 7), np.float32)
        kp = np.zeros((cap, 7), np.float32)
        g = C.c_int(-1)
        self.L.yr_dso_multilevel.restype = C.c_int
        self.L.yr_dso_multilevel.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_int)]
        n = self.L.yr_dso_multilevel(self.h, _p(img), w, h, w, _p(ex) if len(ex) else None, len(ex), _p(kp), cap, C.byref(g))
        assert n >= 0
        return _kp7_to_struct(ex), _kp7_to_struct(kp[:n]), g.value


class Extractor:
    def __init__(self, nfeatures=1000, scale_factor=1.2, nlevels=8, ini_th=20, min_th=7):
        self.L = lib()
        self.nlevels = nlevels
        self.nfeatures = nfeatures
        self.h = C.c_void_p(self.L.yo_extractor_create(nfeatures, scale_factor, nlevels, ini_th, min_th))

    def __del__(self):
        try:
            self.L.yo_extractor_destroy(self.h)
        except Exception:
            pass

    def tables(self):
        n = self.nlevels
        sc, inv, s2, is2 = (np.zeros(n, np.float32) for _ in range(4))
        nf = np.zeros(n, np.int32)
        um = np.zeros(16, np.int32)
        self.L.yo_extractor_tables(self.h, _p(sc), _p(inv), _p(s2), _p(is2), _p(nf), _p(um))
        return dict(scale=sc, inv_scale=inv, sigma2=s2, inv_sigma2=is2, nfeat=nf, umax=um)

    def level_size(self, w, h, level):
        lw, lh = C.c_int(), C.c_int()
        self.L.yo_level_size(self.h, w, h, level, C.byref(lw), C.byref(lh))
        return lw.value, lh.value

    def pyramid(self, img):
        img = np.ascontiguousarray(img, np.uint8)
        h, w = img.shape
        self.L.yo_pyramid(self.h, _p(img), w, h, w)
        out = []
        for l in range(self.nlevels):
            lw, lh = self.level_size(w, h, l)
            a = np.zeros((lh, lw), np.uint8)
            self.L.yo_get_level(self.h, l, _p(a))
            out.append(a)
        return out

    def cell_candidates(self, level, cap=400000):
        xs, ys, sc = (np.zeros(cap, np.int32) for _ in range(3))
        n = self.L.yo_cell_candidates(self.h, level, _p(xs), _p(ys), _p(sc), cap)
        assert n <= cap
        return xs[:n].copy(), ys[:n].copy(), sc[:n].copy()

    def octree(self, xs, ys, resp, minX, maxX, minY, maxY, N):
        xs, ys, resp = (np.ascontiguousarray(a, np.int32) for a in (xs, ys, resp))
        cap = len(xs) + 8
        out = np.zeros(cap, np.int32)
        n = self.L.yo_octree(self.h, _p(xs), _p(ys), _p(resp), len(xs), minX, maxX, minY, maxY, N, _p(out), cap)
        return out[:n].copy()

    def level_keypoints(self, level, cap=100000):
        k = np.zeros(cap, KP_DTYPE)
        n = self.L.yo_level_keypoints(self.h, level, _p(k), cap)
        return k[:n].copy()

    def extract(self, img, cap=None):
        img = np.ascontiguousarray(img, np.uint8)
        h, w = img.shape
        cap = cap or (self.nfeatures * 2 + 64 * self.nlevels)
        k = np.zeros(cap, KP_DTYPE)
        d = np.zeros((cap, 32), np.uint8)
        n = self.L.yo_extract(self.h, _p(img), w, h, w, _p(k), cap, _p(d))
        assert n >= 0, "oracle extract: capacity too small"
        return k[:n].copy(), d[:n].copy()

    def extract_dso(self, img, existing=None, grid_size=-1, cap=None):
        """operator()(Frame*, ..., DSO_KEYPOINT) -> (keys (existing with updated angles + new), desc, new mnGridSize)."""
        img = np.ascontiguousarray(img, np.uint8)
        h, w = img.shape
        existing = np.zeros(0, KP_DTYPE) if existing is None else np.ascontiguousarray(existing, KP_DTYPE)
        g0 = grid_size if grid_size > 0 else max(1, int(np.sqrt(h * w / max(self.cfg.nfeatures if hasattr(self, 'cfg') else self.nfeatures, 1))))
        gm = max(1, min(7, g0))
        cap = cap or (len(existing) + 3 * (w // gm) * (h // gm) + 16)
        k = np.zeros(cap, KP_DTYPE)
        k[:len(existing)] = existing
        d = np.zeros((cap, 32), np.uint8)
        g = C.c_int(grid_size)
        self.L.yo_extract_dso.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p,
                                          C.POINTER(C.c_int)]
        n = self.L.yo_extract_dso(self.h, _p(img), w, h, w, _p(k), len(existing), cap, _p(d), C.byref(g))
        assert n >= 0
        return k[:n].copy(), d[:n].copy(), g.value

    def _extract_grid(self, mode, img, existing, grid_size, cap):
        img = np.ascontiguousarray(img, np.uint8)
        h, w = img.shape
        existing = np.zeros(0, KP_DTYPE) if existing is None else np.ascontiguousarray(existing, KP_DTYPE)
        cap = cap or (len(existing) + (w // 5) * (h // 5) * 2 + 16)
        k = np.zeros(cap, KP_DTYPE)
        k[:len(existing)] = existing
        d = np.zeros((cap, 32), np.uint8)
        g = C.c_int(grid_size)
        self.L.yo_extract_grid.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p,
                                           C.POINTER(C.c_int)]
        n = self.L.yo_extract_grid(self.h, mode, _p(img), w, h, w, _p(k), len(existing), cap, _p(d), C.byref(g))
        assert n >= 0
        return k[:n].copy(), d[:n].copy(), g.value

    def extract_fast(self, img, existing=None, cap=None):
        """operator()(Frame*, ..., FAST_KEYPOINT) (ComputeKeyPointsFast) -> (keys (existing with fresh angles + new), desc)."""
        k, d, _ = self._extract_grid(1, img, existing, -1, cap)
        return k, d

    def extract_dso_multilevel(self, img, existing=None, grid_size=-1, cap=None):
        """The Frame overload over the multi-level ComputeKeyPointsDSO -> (keys, desc, mnGridSize after the call)."""
        return self._extract_grid(3, img, existing, grid_size, cap)

    def describe_keys(self, img, keys, recompute_angle=False):
        """Descriptors (and optionally fresh IC_Angle) of existing keys -> (keys, desc)."""
        img = np.ascontiguousarray(img, np.uint8)
        h, w = img.shape
        k = np.ascontiguousarray(keys, KP_DTYPE).copy()
        d = np.zeros((len(k), 32), np.uint8)
        self.L.yo_describe_keys.restype = None
        self.L.yo_describe_keys.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        self.L.yo_describe_keys(self.h, _p(img), w, h, w, _p(k), len(k), int(recompute_angle), _p(d))
        return k, d

    def compute_stereo_matches(self, img_l, img_r, keys_l, desc_l, keys_r, desc_r, mb, mbf):
        """Frame::ComputeStereoMatches -> (mvuRight, mvDepth)."""
        il, ir = np.ascontiguousarray(img_l, np.uint8), np.ascontiguousarray(img_r, np.uint8)
        h, w = il.shape
        kl, kr = np.ascontiguousarray(keys_l, KP_DTYPE), np.ascontiguousarray(keys_r, KP_DTYPE)
        dl, dr = np.ascontiguousarray(desc_l, np.uint8), np.ascontiguousarray(desc_r, np.uint8)
        ur, dp = np.zeros(max(len(kl), 1), np.float32), np.zeros(max(len(kl), 1), np.float32)
        self.L.yo_compute_stereo_matches.restype = None
        self.L.yo_compute_stereo_matches.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int,
                                                     C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_void_p, C.c_void_p]
        self.L.yo_compute_stereo_matches(self.h, _p(il), _p(ir), w, h, len(kl), _p(kl), _p(dl), len(kr), _p(kr), _p(dr), mb, mbf, _p(ur), _p(dp))
        return ur[:len(kl)], dp[:len(kl)]

    def find_direct_projection_batch(self, ref_imgs, cur_img, cur_Tcw7, cam, ref_slot, ref_Tcw7, ref_kp, mp_world, px_curr):
        """ORBmatcher::FindDirectProjection for n candidates -> (px_curr n x 2, search_level, success, patch_with_border n x 100)."""
        refs = [np.ascontiguousarray(r, np.uint8) for r in ref_imgs]
        cur = np.ascontiguousarray(cur_img, np.uint8)
        h, w = cur.shape
        ptrs = (C.c_void_p * len(refs))(*[r.ctypes.data for r in refs])
        ct = np.ascontiguousarray(cur_Tcw7, np.float32)
        rs = np.ascontiguousarray(ref_slot, np.int32)
        rt = np.ascontiguousarray(ref_Tcw7, np.float32)
        rk = np.ascontiguousarray(ref_kp, KP_DTYPE)
        mw = np.ascontiguousarray(mp_world, np.float32)
        px = np.array(px_curr, np.float32).reshape(-1, 2).copy()
        n = len(rs)
        sl = np.zeros(max(n, 1), np.int32)
        ok = np.zeros(max(n, 1), np.uint8)
        pt = np.zeros((max(n, 1), 100), np.uint8)
        self.L.yo_find_direct_projection_batch.restype = None
        self.L.yo_find_direct_projection_batch.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_float,
                                                           C.c_float, C.c_float, C.c_float, C.c_int] + [C.c_void_p] * 8
        self.L.yo_find_direct_projection_batch(self.h, len(refs), ptrs, _p(cur), w, h, _p(ct), cam["fx"], cam["fy"], cam["cx"], cam["cy"], n,
                                               _p(rs), _p(rt), _p(rk), _p(mw), _p(px), _p(sl), _p(ok), _p(pt))
        return px, sl[:n], ok[:n], pt[:n]

    def shi_tomasi(self, img, u, v):
        img = np.ascontiguousarray(img, np.uint8)
        self.L.yo_shi_tomasi.restype = C.c_float
        self.L.yo_shi_tomasi.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]
        return float(self.L.yo_shi_tomasi(self.h, _p(img), img.shape[1], img.shape[0], u, v))

    def ic_angle(self, img, x, y):
        img = np.ascontiguousarray(img, np.uint8)
        return float(self.L.yo_ic_angle(self.h, _p(img), img.shape[1], img.shape[0], x, y))

    def descriptor(self, blurred, x, y, angle):
        blurred = np.ascontiguousarray(blurred, np.uint8)
        d = np.zeros(32, np.uint8)
        self.L.yo_descriptor(self.h, _p(blurred), blurred.shape[1], blurred.shape[0], x, y, angle, _p(d))
        return d


def blur(img):
    """cv::GaussianBlur 7x7 sigma 2 REFLECT_101 in the current cv mode."""
    img = np.ascontiguousarray(img, np.uint8)
    out = np.zeros_like(img)
    lib().yo_blur(_p(img), img.shape[1], img.shape[0], _p(out))
    return out


def blur_kernel(mode):
    k = np.zeros(7, np.int32)
    lib().yo_blur_kernel(mode, _p(k))
    return k


def resize(img, dw, dh):
    img = np.ascontiguousarray(img, np.uint8)
    out = np.zeros((dh, dw), np.uint8)
    lib().yo_resize(_p(img), img.shape[1], img.shape[0], _p(out), dw, dh)
    return out


def fast9(img, threshold, nonmax=True):
    img = np.ascontiguousarray(img, np.uint8)
    h, w = img.shape
    cap = w * h
    xs, ys, sc = (np.zeros(cap, np.int32) for _ in range(3))
    n = lib().yo_fast9(_p(img), w, w, h, threshold, int(nonmax), _p(xs), _p(ys), _p(sc), cap)
    return xs[:n].copy(), ys[:n].copy(), sc[:n].copy()


def fast_atan2(y, x):
    return float(lib().yo_fast_atan2(y, x))


def sincos_deg(a):
    c, s = C.c_float(), C.c_float()
    lib().yo_sincos_deg(a, C.byref(c), C.byref(s))
    return c.value, s.value


def hamming(a, b):
    a = np.ascontiguousarray(a, np.uint8)
    b = np.ascontiguousarray(b, np.uint8)
    return int(lib().yo_hamming(_p(a), _p(b)))


def fast10(img, barrier, stride=None):
    img = np.ascontiguousarray(img, np.uint8)
    h, w = img.shape
    cap = w * h
    xy = np.zeros((cap, 2), np.int16)
    n = lib().yo_fast10_detect(_p(img), w, h, stride or w, barrier, _p(xy), cap)
    xy = xy[:n].copy()
    sc = np.zeros(n, np.int32)
    lib().yo_fast10_score(_p(img), stride or w, _p(xy), n, _p(sc))
    nm = np.zeros(max(n, 1), np.int32)
    m = lib().yo_fast_nonmax_3x3(_p(xy), _p(sc), n, _p(nm), n)
    return xy, sc, nm[:m].copy()


def ref_fast10(img, barrier, which=1):
    R = ref_fast()
    img = np.ascontiguousarray(img, np.uint8)
    h, w = img.shape
    cap = w * h
    xy = np.zeros((cap, 2), np.int16)
    n = R.ref_fast10_detect(which, _p(img), w, h, w, barrier, _p(xy), cap)
    xy = xy[:n].copy()
    sc = np.zeros(n, np.int32)
    R.ref_fast10_score(_p(img), w, _p(xy), n, barrier, _p(sc))
    nm = np.zeros(max(n, 1), np.int32)
    m = R.ref_fast_nonmax_3x3(_p(xy), _p(sc), n, _p(nm), n)
    return xy, sc, nm[:m].copy()


def bench_extract_match(frames, nfeatures=1000, scale_factor=1.2, nlevels=8, ini_th=20, min_th=7, threads=1,
                        frames_per_thread=None, max_seconds=0.0, fx=458.654, fy=457.296, cx=367.215, cy=248.375, match=True, align=False, stereo=False):
    """CPU baseline ('port'): seconds for `threads` workers each doing extract (+ frame-to-frame projection match, + SparseImgAlign of the same
    pair, + ComputeStereoMatches on (left, right) pairs) on `frames_per_thread` consecutive frames of the clip `frames` (n,h,w) u8.  Stops early at
    `max_seconds` (0 = no limit; a frame / pair that was started is finished).  Returns (seconds, keypoints, matches, frames_done)."""
    frames = np.ascontiguousarray(frames, np.uint8)
    n, h, w = frames.shape
    if frames_per_thread is None:
        frames_per_thread = max(1, n // threads)
    nk, nm, nf = C.c_long(), C.c_long(), C.c_long()
    L = lib()
    L.yo_bench_extract_match.restype = C.c_double
    L.yo_bench_extract_match.argtypes = [C.c_int, C.c_float, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double,
                                         C.c_float, C.c_float, C.c_float, C.c_float, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    mode = (1 if match else 0) | (2 if align else 0) | (4 if stereo else 0)
    sec = L.yo_bench_extract_match(nfeatures, scale_factor, nlevels, ini_th, min_th, _p(frames), n, w, h, threads, frames_per_thread, max_seconds,
                                   fx, fy, cx, cy, mode, C.byref(nk), C.byref(nm), C.byref(nf))
    return sec, nk.value, nm.value, nf.value


class _YoFrame(C.Structure):
    _fields_ = [("N", C.c_int), ("keys", C.c_void_p), ("desc", C.c_void_p), ("uRight", C.c_void_p), ("minX", C.c_float),
                ("minY", C.c_float), ("maxX", C.c_float), ("maxY", C.c_float), ("fx", C.c_float), ("fy", C.c_float),
                ("cx", C.c_float), ("cy", C.c_float), ("mb", C.c_float), ("mbf", C.c_float), ("scaleFactors", C.c_void_p),
                ("nlevels", C.c_int)]


def _yo_frame(keys, desc, scale_factors, w, h, fx, fy, cx, cy, mb=0.0, mbf=0.0, u_right=None, keep=None):
    keys = np.ascontiguousarray(keys, KP_DTYPE)
    desc = np.ascontiguousarray(desc, np.uint8)
    sf = np.ascontiguousarray(scale_factors, np.float32)
    keep.extend([keys, desc, sf])
    ur = None
    if u_right is not None:
        ur = np.ascontiguousarray(u_right, np.float32)
        keep.append(ur)
    return _YoFrame(len(keys), keys.ctypes.data, desc.ctypes.data, ur.ctypes.data if ur is not None else None, 0.0, 0.0, float(w),
                    float(h), fx, fy, cx, cy, mb, mbf, sf.ctypes.data, len(sf))


def search_by_projection_last(cur_keys, cur_desc, scale_factors, w, h, cam, last_keys, mp_world, mp_desc, Rcw, tcw, Rlw, tlw, th,
                              mono=True, check_level=True, check_ori=True, mp_valid=None, outlier=None, mp_has_obs=None,
                              u_right=None, cur_owner=None):
    """Oracle ORBmatcher::SearchByProjection(Cur, Last, ...) -> (nmatches, cur_match, cur_owner).  cam = dict(fx,fy,cx,cy[,mb,mbf])."""
    keep = []
    fr = _yo_frame(cur_keys, cur_desc, scale_factors, w, h, cam["fx"], cam["fy"], cam["cx"], cam["cy"], cam.get("mb", 0.0),
                   cam.get("mbf", 0.0), u_right, keep)
    lk = np.ascontiguousarray(last_keys, KP_DTYPE)
    n = len(lk)
    valid = np.ones(n, np.uint8) if mp_valid is None else np.ascontiguousarray(mp_valid, np.uint8)
    outl = np.zeros(n, np.uint8) if outlier is None else np.ascontiguousarray(outlier, np.uint8)
    obs = np.ones(n, np.uint8) if mp_has_obs is None else np.ascontiguousarray(mp_has_obs, np.uint8)
    mw = np.ascontiguousarray(mp_world, np.float32)
    md = np.ascontiguousarray(mp_desc, np.uint8)
    mats = [np.ascontiguousarray(a, np.float32) for a in (Rcw, tcw, Rlw, tlw)]
    nt = fr.N
    owner = np.zeros(max(nt, 1), np.uint8) if cur_owner is None else np.array(cur_owner, np.uint8)
    match = np.full(max(nt, 1), -1, np.int32)
    L = lib()
    L.yo_search_by_projection_last.argtypes = [C.POINTER(_YoFrame), C.c_int] + [C.c_void_p] * 10 + [C.c_float, C.c_int, C.c_int,
                                                                                                   C.c_int, C.c_void_p, C.c_void_p]
    r = L.yo_search_by_projection_last(C.byref(fr), n, _p(lk), _p(valid), _p(outl), _p(obs), _p(mw), _p(md), _p(mats[0]),
                                       _p(mats[1]), _p(mats[2]), _p(mats[3]), th, int(mono), int(check_level), int(check_ori),
                                       _p(owner), _p(match))
    return r, match[:nt], owner[:nt]


class _YoAlignFrame(C.Structure):
    _fields_ = [("N", C.c_int), ("keys", C.c_void_p), ("mp_valid", C.c_void_p), ("outlier", C.c_void_p), ("mp_world", C.c_void_p),
                ("Tcw", C.c_float * 7), ("nlevels", C.c_int), ("levels", C.POINTER(C.c_void_p)), ("level_w", C.c_void_p),
                ("level_h", C.c_void_p), ("invScaleFactors", C.c_void_p), ("fx", C.c_float), ("fy", C.c_float), ("cx", C.c_float),
                ("cy", C.c_float)]


def _align_frame(keys, mp_valid, outlier, mp_world, Tcw7, pyramid, inv_scale, cam, keep):
    keys = np.ascontiguousarray(keys, KP_DTYPE)
    n = len(keys)
    mv = np.ones(n, np.uint8) if mp_valid is None else np.ascontiguousarray(mp_valid, np.uint8)
    ol = np.zeros(n, np.uint8) if outlier is None else np.ascontiguousarray(outlier, np.uint8)
    mw = np.zeros((max(n, 1), 3), np.float32) if mp_world is None else np.ascontiguousarray(mp_world, np.float32)
    pyr = [np.ascontiguousarray(p, np.uint8) for p in pyramid]
    lw = np.array([p.shape[1] for p in pyr], np.int32)
    lh = np.array([p.shape[0] for p in pyr], np.int32)
    isf = np.ascontiguousarray(inv_scale, np.float32)
    arr = (C.c_void_p * len(pyr))(*[p.ctypes.data for p in pyr])
    keep.extend([keys, mv, ol, mw, pyr, lw, lh, isf, arr])
    f = _YoAlignFrame()
    f.N = n
    f.keys, f.mp_valid, f.outlier, f.mp_world = keys.ctypes.data, mv.ctypes.data, ol.ctypes.data, mw.ctypes.data
    for i in range(7):
        f.Tcw[i] = float(Tcw7[i])
    f.nlevels = len(pyr)
    f.levels = arr
    f.level_w, f.level_h, f.invScaleFactors = lw.ctypes.data, lh.ctypes.data, isf.ctypes.data
    f.fx, f.fy, f.cx, f.cy = cam["fx"], cam["fy"], cam["cx"], cam["cy"]
    return f


def sparse_img_align(ref_keys, ref_world, ref_Tcw7, ref_pyr, cur_Tcw7, cur_pyr, inv_scale, cam, max_level, min_level, n_iter=10,
                     mp_valid=None, outlier=None, device_order=False):
    """Oracle SparseImgAlign(max_level, min_level, n_iter).run(ref, cur, TCR) -> (ret, TCR7 (qx qy qz qw tx ty tz), info, H).
    device_order: the normal equations in the HIP kernel's formulation / fused multiply-adds / reduction tree (bit-for-bit comparison mode)."""
    keep = []
    R = _align_frame(ref_keys, mp_valid, outlier, ref_world, ref_Tcw7, ref_pyr, inv_scale, cam, keep)
    Cf = _align_frame(np.zeros(0, KP_DTYPE), None, None, None, cur_Tcw7, cur_pyr, inv_scale, cam, keep)
    out7 = np.zeros(7, np.float32)
    info = np.zeros(2, np.float32)
    H = np.zeros(36, np.float32)
    L = lib()
    if device_order:
        L.yo_sparse_img_align_mode.restype = C.c_size_t
        L.yo_sparse_img_align_mode.argtypes = [C.POINTER(_YoAlignFrame), C.POINTER(_YoAlignFrame), C.c_int, C.c_int, C.c_int, C.c_void_p,
                                               C.c_void_p, C.c_void_p, C.c_int]
        ret = L.yo_sparse_img_align_mode(C.byref(R), C.byref(Cf), max_level, min_level, n_iter, _p(out7), _p(info), _p(H), 1)
        return int(ret), out7, info, H.reshape(6, 6)
    L.yo_sparse_img_align.argtypes = [C.POINTER(_YoAlignFrame), C.POINTER(_YoAlignFrame), C.c_int, C.c_int, C.c_int, C.c_void_p,
                                      C.c_void_p, C.c_void_p]
    ret = L.yo_sparse_img_align(C.byref(R), C.byref(Cf), max_level, min_level, n_iter, _p(out7), _p(info), _p(H))
    return int(ret), out7, info, H.reshape(6, 6)


def sparse_img_align_f64(ref_keys, ref_world, ref_Tcw7, ref_pyr, cur_Tcw7, cur_pyr, inv_scale, cam, max_level, min_level, n_iter=10,
                         mp_valid=None, outlier=None):
    """The same Gauss-Newton with every quantity in double (oracle_align.cpp, sparse_img_align_f64): (ret, TCR7 as float64, (iterations, chi2)).
    The third party of the aligner's tolerance argument: tests report |device - this| beside |reference_order - this|."""
    keep = []
    R = _align_frame(ref_keys, mp_valid, outlier, ref_world, ref_Tcw7, ref_pyr, inv_scale, cam, keep)
    Cf = _align_frame(np.zeros(0, KP_DTYPE), None, None, None, cur_Tcw7, cur_pyr, inv_scale, cam, keep)
    out7 = np.zeros(7, np.float64)
    info = np.zeros(2, np.float64)
    L = lib()
    L.yo_sparse_img_align_f64.restype = C.c_size_t
    L.yo_sparse_img_align_f64.argtypes = [C.POINTER(_YoAlignFrame), C.POINTER(_YoAlignFrame), C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    ret = L.yo_sparse_img_align_f64(C.byref(R), C.byref(Cf), max_level, min_level, n_iter, _p(out7), _p(info))
    return int(ret), out7, info


_ref_sophus = None


def ref_sophus_lib():
    """oracle/_ref/libref_sophus.so: the reference's own Thirdparty/sophus headers over ref_shim/eigen_min (None when oracle/_ref was never built)."""
    global _ref_sophus
    if _ref_sophus is None:
        build()
        p = os.path.join(_HERE, "_ref", "libref_sophus.so")
        if not os.path.exists(p):
            return None
        _ref_sophus = C.CDLL(p)
    return _ref_sophus


def se3_act(a7, p3):
    a, p = np.ascontiguousarray(a7, np.float32), np.ascontiguousarray(p3, np.float32)
    out = np.zeros(3, np.float32)
    lib().yo_se3_act(_p(a), _p(p), _p(out))
    return out


def se3_exp(a6):
    a = np.ascontiguousarray(a6, np.float32)
    out = np.zeros(7, np.float32)
    lib().yo_se3_exp(_p(a), _p(out))
    return out


def se3_mul(a7, b7):
    a, b = np.ascontiguousarray(a7, np.float32), np.ascontiguousarray(b7, np.float32)
    out = np.zeros(7, np.float32)
    lib().yo_se3_mul(_p(a), _p(b), _p(out))
    return out


def se3_inverse(a7):
    a = np.ascontiguousarray(a7, np.float32)
    out = np.zeros(7, np.float32)
    lib().yo_se3_inverse(_p(a), _p(out))
    return out


def features_in_area(keys, scale_factors, w, h, x, y, r, min_level=-1, max_level=-1):
    keep = []
    fr = _yo_frame(keys, np.zeros((len(keys), 32), np.uint8), scale_factors, w, h, 1.0, 1.0, 0.0, 0.0, keep=keep)
    out = np.zeros(max(len(keys), 1), np.int32)
    L = lib()
    L.yo_features_in_area.argtypes = [C.POINTER(_YoFrame), C.c_float, C.c_float, C.c_float, C.c_int, C.c_int, C.c_void_p, C.c_int]
    n = L.yo_features_in_area(C.byref(fr), x, y, r, min_level, max_level, _p(out), len(out))
    return out[:n].copy()


def search_by_projection_kf(keys, desc, scale_factors, w, h, cam, usable, world, max_dist_inv, min_dist_inv, mf_max_distance, kf_angle,
                            mp_desc, Rcw, tcw, log_scale_factor, th, orb_dist, check_ori=True, owner=None):
    """Oracle ORBmatcher::SearchByProjection(Cur, KF, found, th, ORBdist) -> (nmatches, match, owner, (valid, u, v, level))."""
    keep = []
    fr = _yo_frame(keys, desc, scale_factors, w, h, cam["fx"], cam["fy"], cam["cx"], cam["cy"], cam.get("mb", 0.0), cam.get("mbf", 0.0),
                   None, keep)
    M = len(usable)
    us = np.ascontiguousarray(usable, np.uint8)
    wd, mx, mn, mf, ka = (np.ascontiguousarray(a, np.float32) for a in (world, max_dist_inv, min_dist_inv, mf_max_distance, kf_angle))
    md = np.ascontiguousarray(mp_desc, np.uint8)
    R = np.ascontiguousarray(Rcw, np.float32)
    t = np.ascontiguousarray(tcw, np.float32)
    nt = fr.N
    own = np.zeros(max(nt, 1), np.uint8) if owner is None else np.array(owner, np.uint8)
    match = np.full(max(nt, 1), -1, np.int32)
    ov = np.zeros(max(M, 1), np.uint8)
    ou, ovv = np.zeros(max(M, 1), np.float32), np.zeros(max(M, 1), np.float32)
    ol = np.zeros(max(M, 1), np.int32)
    L = lib()
    L.yo_search_by_projection_kf.argtypes = [C.POINTER(_YoFrame), C.c_int] + [C.c_void_p] * 9 + [C.c_float, C.c_int, C.c_float, C.c_int, C.c_int] + \
        [C.c_void_p] * 6
    r = L.yo_search_by_projection_kf(C.byref(fr), M, _p(us), _p(wd), _p(mx), _p(mn), _p(mf), _p(ka), _p(md), _p(R), _p(t),
                                     float(log_scale_factor), len(scale_factors), th, orb_dist, int(check_ori), _p(own), _p(match), _p(ov), _p(ou),
                                     _p(ovv), _p(ol))
    return r, match[:nt], own[:nt], (ov[:M], ou[:M], ovv[:M], ol[:M])


def search_for_initialization(keys1, desc1, keys2, desc2, scale_factors, w, h, cam, prev_matched_xy, window=100, nnratio=0.9, check_ori=True):
    """Oracle ORBmatcher::SearchForInitialization -> (nmatches, matches12, updated prev_matched_xy)."""
    keep = []
    f1 = _yo_frame(keys1, desc1, scale_factors, w, h, cam["fx"], cam["fy"], cam["cx"], cam["cy"], 0.0, 0.0, None, keep)
    f2 = _yo_frame(keys2, desc2, scale_factors, w, h, cam["fx"], cam["fy"], cam["cx"], cam["cy"], 0.0, 0.0, None, keep)
    pm = np.array(prev_matched_xy, np.float32).reshape(-1, 2).copy()
    m12 = np.full(max(f1.N, 1), -1, np.int32)
    L = lib()
    L.yo_search_for_initialization.argtypes = [C.POINTER(_YoFrame), C.POINTER(_YoFrame), C.c_void_p, C.c_int, C.c_float, C.c_int, C.c_void_p]
    r = L.yo_search_for_initialization(C.byref(f1), C.byref(f2), _p(pm), int(window), nnratio, int(check_ori), _p(m12))
    return r, m12[:f1.N], pm


def search_by_bow(kf_off, kf_idx, f_off, f_idx, kf_valid, kf_keys, kf_desc, f_keys, f_desc, nnratio=0.7, check_ori=True):
    """Oracle ORBmatcher::SearchByBoW(KF, F) on a joined node list -> (nmatches, match per Frame feature)."""
    ko, ki, fo, fi = (np.ascontiguousarray(a, np.int32) for a in (kf_off, kf_idx, f_off, f_idx))
    kv = np.ascontiguousarray(kf_valid, np.uint8)
    kk, fk = np.ascontiguousarray(kf_keys, KP_DTYPE), np.ascontiguousarray(f_keys, KP_DTYPE)
    kd, fd = np.ascontiguousarray(kf_desc, np.uint8), np.ascontiguousarray(f_desc, np.uint8)
    match = np.full(max(len(fk), 1), -1, np.int32)
    L = lib()
    L.yo_search_by_bow.argtypes = [C.c_int] + [C.c_void_p] * 7 + [C.c_int, C.c_void_p, C.c_void_p, C.c_float, C.c_int, C.c_void_p]
    r = L.yo_search_by_bow(len(ko) - 1, _p(ko), _p(ki), _p(fo), _p(fi), _p(kv), _p(kk), _p(kd), len(fk), _p(fk), _p(fd), nnratio, int(check_ori),
                           _p(match))
    return r, match[:len(fk)]


def search_for_triangulation(off1, idx1, off2, idx2, kf1, kf2, scale_factors2, level_sigma2_2, F12, Cw1, R2w, t2w, cam2,
                             only_stereo=False, check_ori=True):
    """Oracle ORBmatcher::SearchForTriangulation on a joined node list.  kf1 / kf2: dicts with keys (KP_DTYPE), desc (n x 32), has_mp
    (n bytes), u_right (n floats or None); cam2 = (fx, fy, cx, cy) of KF2 -> (nmatches, match12 per KF1 feature)."""
    o1, i1, o2, i2 = (np.ascontiguousarray(a, np.int32) for a in (off1, idx1, off2, idx2))
    keep = []

    def side(kf):
        k = np.ascontiguousarray(kf["keys"], KP_DTYPE)
        d = np.ascontiguousarray(kf["desc"], np.uint8)
        m = np.ascontiguousarray(kf["has_mp"], np.uint8)
        u = None if kf.get("u_right") is None else np.ascontiguousarray(kf["u_right"], np.float32)
        keep.extend([k, d, m, u])
        return len(k), _p(k), _p(d), _p(m), (_p(u) if u is not None else None)

    sf, sg = np.ascontiguousarray(scale_factors2, np.float32), np.ascontiguousarray(level_sigma2_2, np.float32)
    Fm, Cw, Rm, tm, cam = (np.ascontiguousarray(a, np.float32).reshape(-1) for a in (F12, Cw1, R2w, t2w, cam2))
    a, b = side(kf1), side(kf2)
    match = np.full(max(a[0], 1), -1, np.int32)
    L = lib()
    L.yo_search_for_triangulation.argtypes = ([C.c_int] + [C.c_void_p] * 4 + [C.c_int] + [C.c_void_p] * 4 + [C.c_int] + [C.c_void_p] * 4 + [C.c_int] +
                                              [C.c_void_p] * 7 + [C.c_int, C.c_int, C.c_void_p])
    r = L.yo_search_for_triangulation(len(o1) - 1, _p(o1), _p(i1), _p(o2), _p(i2), *a, *b, len(sf), _p(sf), _p(sg), _p(Fm), _p(Cw), _p(Rm), _p(tm),
                                      _p(cam), int(only_stereo), int(check_ori), _p(match))
    return r, match[:a[0]]


def is_in_frustum(keys, desc, scale_factors, w, h, cam, world, normal, max_dist_inv, min_dist_inv, mf_max_distance, Rcw, tcw, Ow,
                  log_scale_factor, viewing_cos_limit=0.5):
    """Oracle Frame::isInFrustum for M points -> (in_view, projX, projY, projXR, level, viewCos)."""
    keep = []
    fr = _yo_frame(keys, desc, scale_factors, w, h, cam["fx"], cam["fy"], cam["cx"], cam["cy"], cam.get("mb", 0.0), cam.get("mbf", 0.0),
                   None, keep)
    wd, nm, mx, mn, mf = (np.ascontiguousarray(a, np.float32) for a in (world, normal, max_dist_inv, min_dist_inv, mf_max_distance))
    R, t, O = (np.ascontiguousarray(a, np.float32) for a in (Rcw, tcw, Ow))
    M = len(mx)
    iv = np.zeros(max(M, 1), np.uint8)
    px, py, pxr, vc = (np.zeros(max(M, 1), np.float32) for _ in range(4))
    lv = np.zeros(max(M, 1), np.int32)
    L = lib()
    L.yo_is_in_frustum.restype = None
    L.yo_is_in_frustum.argtypes = [C.POINTER(_YoFrame), C.c_int] + [C.c_void_p] * 8 + [C.c_float, C.c_int, C.c_float] + [C.c_void_p] * 6
    L.yo_is_in_frustum(C.byref(fr), M, _p(wd), _p(nm), _p(mx), _p(mn), _p(mf), _p(R), _p(t), _p(O), float(log_scale_factor), len(scale_factors),
                       viewing_cos_limit, _p(iv), _p(px), _p(py), _p(pxr), _p(lv), _p(vc))
    return iv[:M], px[:M], py[:M], pxr[:M], lv[:M], vc[:M]


def predict_scale(ratio, log_scale_factor, nlevels):
    r = np.ascontiguousarray(ratio, np.float32)
    out = np.zeros(max(len(r), 1), np.int32)
    L = lib()
    L.yo_predict_scale.restype = None
    L.yo_predict_scale.argtypes = [C.c_void_p, C.c_int, C.c_float, C.c_int, C.c_void_p]
    L.yo_predict_scale(_p(r), len(r), float(log_scale_factor), nlevels, _p(out))
    return out[:len(r)]


def distinctive_descriptors(obs_off, desc):
    """Oracle MapPoint::ComputeDistinctiveDescriptors for a batch of points -> index (within the point's observations) of the winner."""
    oo = np.ascontiguousarray(obs_off, np.int32)
    d = np.ascontiguousarray(desc, np.uint8)
    best = np.zeros(max(len(oo) - 1, 1), np.int32)
    L = lib()
    L.yo_distinctive_descriptors.restype = None
    L.yo_distinctive_descriptors.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    L.yo_distinctive_descriptors(len(oo) - 1, _p(oo), _p(d), _p(best))
    return best[:len(oo) - 1]


def search_by_projection_mappoints(keys, desc, scale_factors, w, h, cam, track_in_view, proj_x, proj_y, view_cos, scale_level, mp_desc,
                                   th, check_level=True, nnratio=0.8, is_bad=None, mp_has_obs=None, proj_xr=None, u_right=None,
                                   owner=None):
    """Oracle ORBmatcher::SearchByProjection(F, MapPoints, th, checkLevel) -> (nmatches, match, owner)."""
    keep = []
    fr = _yo_frame(keys, desc, scale_factors, w, h, cam["fx"], cam["fy"], cam["cx"], cam["cy"], cam.get("mb", 0.0), cam.get("mbf", 0.0),
                   u_right, keep)
    M = len(proj_x)
    tiv = np.ascontiguousarray(track_in_view, np.uint8)
    bad = np.zeros(M, np.uint8) if is_bad is None else np.ascontiguousarray(is_bad, np.uint8)
    obs = np.ones(M, np.uint8) if mp_has_obs is None else np.ascontiguousarray(mp_has_obs, np.uint8)
    px, py, vc = (np.ascontiguousarray(a, np.float32) for a in (proj_x, proj_y, view_cos))
    pxr = np.zeros(M, np.float32) if proj_xr is None else np.ascontiguousarray(proj_xr, np.float32)
    lv = np.ascontiguousarray(scale_level, np.int32)
    md = np.ascontiguousarray(mp_desc, np.uint8)
    nt = fr.N
    own = np.zeros(max(nt, 1), np.uint8) if owner is None else np.array(owner, np.uint8)
    match = np.full(max(nt, 1), -1, np.int32)
    L = lib()
    L.yo_search_by_projection_mappoints.argtypes = [C.POINTER(_YoFrame), C.c_int] + [C.c_void_p] * 9 + [C.c_float, C.c_int, C.c_float,
                                                                                                        C.c_void_p, C.c_void_p]
    r = L.yo_search_by_projection_mappoints(C.byref(fr), M, _p(tiv), _p(bad), _p(obs), _p(px), _p(py), _p(pxr), _p(vc), _p(lv), _p(md), th,
                                            int(check_level), nnratio, _p(own), _p(match))
    return r, match[:nt], own[:nt]


# ---- Frame::ComputeBoW / DBoW2 vocabulary -------------------------------------------------------------------------------------
# The oracle of the tree descent is this numpy restatement of TemplatedVocabulary::transform (Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:
# 1151-1283) + BowVector::addWeight / normalize (BowVector.cpp:34-84) + FeatureVector::addFeature; tests/test_ref_dbow2.py pins it to the
# reference's own DBoW2 compiled into oracle/_ref/libref_dbow2.so.  ORBvoc.bin itself is a blob the reference does not ship
# (.MISSING_LARGE_BLOBS): vocabularies here are generated (same k-ary layout, random centroids and idf weights).

def make_vocabulary(seed, k=10, L=3, stop_fraction=0.02):
    """A full k-ary tree of depth L as the loaders build it: (parent, is_leaf, desc n x 32, weight) in node-id order, node 0 = root;
    the children of a node are consecutive ids (HKmeans creation order).  A few words get weight 0 (stopped words)."""
    rng = np.random.default_rng(seed)
    parent, level = [-1], [0]
    frontier = [0]
    for lv in range(1, L + 1):
        nxt = []
        for p in frontier:
            for _ in range(k):
                parent.append(p); level.append(lv); nxt.append(len(parent) - 1)
        frontier = nxt
    n = len(parent)
    parent = np.array(parent, np.int32)
    level = np.array(level)
    is_leaf = (level == L).astype(np.uint8)
    desc = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    # children resemble their parent (a few flipped bits per level), so that descents are decided by small margins and ties occur
    for i in range(1, n):
        flip = np.zeros(32, np.uint8)
        for b in rng.integers(0, 256, 24 - 3 * min(level[i], 6)):
            flip[b >> 3] ^= np.uint8(1 << (b & 7))
        desc[i] = desc[parent[i]] ^ flip
    weight = np.where(is_leaf > 0, np.round(rng.uniform(0.5, 9.0, n), 6), 0.0)
    weight[(is_leaf > 0) & (rng.random(n) < stop_fraction)] = 0.0
    return dict(k=k, L=L, parent=parent, is_leaf=is_leaf, desc=desc, weight=weight)


def write_vocabulary_text(voc, path, scoring=0, weighting=0):
    """The text format of TemplatedVocabulary::saveToTextFile / loadFromTextFile (:1362-1470): "k L scoring weighting", then one line per
    node 1..n-1: parent isLeaf 32 descriptor bytes weight."""
    lines = ["%d %d %d %d" % (voc["k"], voc["L"], scoring, weighting)]
    for i in range(1, len(voc["parent"])):
        lines.append("%d %d %s %.6f" % (voc["parent"][i], voc["is_leaf"][i], " ".join(str(int(b)) for b in voc["desc"][i]), voc["weight"][i]))
    with open(path, "w") as f:
        f.write("\n".join(lines))     # no trailing newline: the loader's `while(!f.eof())` loop would turn it into one more (bogus) node


_POP8 = np.array([bin(i).count("1") for i in range(256)], np.int32)


def bow_descend(voc, desc, levelsup=4):
    """Per descriptor: (leaf node id, node id at level L - levelsup) -- the descent of :1240-1283, first nearest child wins."""
    parent = voc["parent"]
    n_nodes = len(parent)
    order = np.argsort(parent[1:], kind="stable") + 1            # children in ascending node id, grouped by parent
    counts = np.bincount(parent[1:], minlength=n_nodes)
    off = np.concatenate([[0], np.cumsum(counts)])
    d = np.ascontiguousarray(desc, np.uint8)
    n = len(d)
    node = np.zeros(n, np.int64)
    nid = np.zeros(n, np.int64)
    nid_level = voc["L"] - levelsup
    active = np.ones(n, bool)
    level = 0
    while active.any():
        level += 1
        for i in np.nonzero(active)[0]:
            c0, c1 = off[node[i]], off[node[i] + 1]
            if c1 == c0:
                active[i] = False
                if nid_level > 0 and level - 1 < nid_level:
                    nid[i] = node[i]
                continue
            ch = order[c0:c1]
            dist = _POP8[voc["desc"][ch] ^ d[i]].sum(1)
            node[i] = ch[int(np.argmin(dist))]                      # argmin returns the first minimum
            if level == nid_level:
                nid[i] = node[i]
    return node.astype(np.int32), nid.astype(np.int32)


def bow_vectors(voc, leaf, nid, scoring_l1=True):
    """BowVector (ascending word id, tf-idf summed in feature order, L1-normalised) and FeatureVector (node -> feature indices) from the
    descent -- what transform(features, v, fv, levelsup) assembles (:1151-1238)."""
    word_of = np.cumsum(voc["is_leaf"]) - 1                         # WordId = rank among the leaves in node order (loadFromTextFile :1433-1440)
    bow, fv = {}, {}
    for i, (lf, nd) in enumerate(zip(leaf, nid)):
        w = float(voc["weight"][lf])
        if w > 0:
            wid = int(word_of[lf])
            bow[wid] = bow.get(wid, 0.0) + w
            fv.setdefault(int(nd), []).append(i)
    ids = sorted(bow)
    vals = [bow[i] for i in ids]
    if scoring_l1:
        norm = 0.0
        for v in vals:
            norm += abs(v)
        if norm > 0:
            vals = [v / norm for v in vals]
    return np.array(ids, np.int32), np.array(vals, np.float64), {k: np.array(v, np.int32) for k, v in sorted(fv.items())}


def ref_dbow2_lib():
    build()
    p = os.path.join(_HERE, "_ref", "libref_dbow2.so")
    if not os.path.exists(p):
        return None
    L = C.CDLL(p)
    L.yr_voc_load_text.restype = C.c_void_p
    L.yr_voc_load_text.argtypes = [C.c_char_p]
    L.yr_voc_free.argtypes = [C.c_void_p]
    L.yr_voc_size.argtypes = [C.c_void_p]
    L.yr_voc_transform.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_int), C.c_void_p, C.c_void_p,
                                   C.c_void_p, C.c_int, C.POINTER(C.c_int)]
    return L


class RefVocabulary:
    """The reference's ORBVocabulary (DBoW2::TemplatedVocabulary<FORB::TDescriptor, FORB>) loaded from a text file by its own loader."""

    def __init__(self, path):
        self.L = ref_dbow2_lib()
        assert self.L is not None, "oracle/_ref/libref_dbow2.so not built"
        self.h = self.L.yr_voc_load_text(path.encode())
        assert self.h, "loadFromTextFile failed"

    def size(self):
        return self.L.yr_voc_size(self.h)

    def transform(self, desc, levelsup=4):
        d = np.ascontiguousarray(desc, np.uint8)
        n = len(d)
        ids, vals = np.zeros(max(n, 1), np.int32), np.zeros(max(n, 1), np.float64)
        nodes, off, idx = np.zeros(max(n, 1), np.int32), np.zeros(n + 2, np.int32), np.zeros(max(n, 1), np.int32)
        nb, nf = C.c_int(), C.c_int()
        rc = self.L.yr_voc_transform(self.h, _p(d), n, levelsup, _p(ids), _p(vals), len(ids), C.byref(nb), _p(nodes), _p(off), _p(idx), len(nodes), C.byref(nf))
        assert rc == 0
        fv = {int(nodes[k]): idx[off[k]:off[k + 1]].copy() for k in range(nf.value)}
        return ids[:nb.value].copy(), vals[:nb.value].copy(), fv

    def __del__(self):
        try:
            self.L.yr_voc_free(self.h)
        except Exception:
            pass


def ref_fast9_corners(img, barrier):
    """The REFERENCE's own FAST-9 decision tree (Thirdparty/fast/include/fast/corner_9.h:1, is_corner_9<Less|Greater>) over the interior
    [3, w-3) x [3, h-3): (xs, ys) of the corners in raster order.  None when oracle/_ref was never built."""
    R = ref_fast()
    if R is None or not hasattr(R, "ref_fast9_corners"):
        return None
    img = np.ascontiguousarray(img, np.uint8)
    h, w = img.shape
    flags = np.zeros((h, w), np.uint8)
    n = R.ref_fast9_corners(_p(img), w, h, w, int(barrier), _p(flags))
    ys, xs = np.nonzero(flags)
    assert len(xs) == n
    return xs.astype(np.int32), ys.astype(np.int32)


def ref_fast9_max_barrier(img, xs, ys):
    """Largest barrier at which the reference's tree still calls (x, y) a corner (-1: not even at 0)."""
    R = ref_fast()
    img = np.ascontiguousarray(img, np.uint8)
    xs = np.ascontiguousarray(xs, np.int32)
    ys = np.ascontiguousarray(ys, np.int32)
    out = np.zeros(len(xs), np.int32)
    R.ref_fast9_max_barrier(_p(img), img.shape[1], _p(xs), _p(ys), len(xs), _p(out))
    return out
