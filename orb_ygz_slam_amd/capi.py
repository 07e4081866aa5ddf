"""ctypes binding of include/ygzf.h (libygzf.so)."""
import ctypes as C
import os

import numpy as np

from . import build as _build

KP_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"), ("response", "<f4"),
                     ("octave", "<i4"), ("class_id", "<i4")])
assert KP_DTYPE.itemsize == 28


class YgzfError(RuntimeError):
    pass


class ExtractorCfg(C.Structure):
    _fields_ = [("nfeatures", C.c_int), ("scale_factor", C.c_float), ("nlevels", C.c_int), ("ini_th_fast", C.c_int),
                ("min_th_fast", C.c_int), ("cv_mode", C.c_int)]


CV_LEGACY_SSE2, CV_LEGACY_INT, CV_4 = 0, 1, 2   # ygzf_cv_mode


class Camera(C.Structure):
    _fields_ = [("fx", C.c_float), ("fy", C.c_float), ("cx", C.c_float), ("cy", C.c_float), ("mb", C.c_float),
                ("mbf", C.c_float), ("min_x", C.c_float), ("min_y", C.c_float), ("max_x", C.c_float), ("max_y", C.c_float)]


class FrameView(C.Structure):
    _fields_ = [("n", C.c_int), ("keys", C.c_void_p), ("desc", C.c_void_p), ("u_right", C.c_void_p),
                ("scale_factors", C.c_void_p), ("nlevels", C.c_int)]


class FrustumIn(C.Structure):
    _fields_ = [("world", C.c_void_p), ("normal", C.c_void_p), ("max_dist_inv", C.c_void_p), ("min_dist_inv", C.c_void_p),
                ("mf_max_distance", C.c_void_p), ("candidate", C.c_void_p), ("Rcw", C.c_float * 9), ("tcw", C.c_float * 3), ("Ow", C.c_float * 3),
                ("log_scale_factor", C.c_float), ("viewing_cos_limit", C.c_float)]


class SiaFrame(C.Structure):
    _fields_ = [("n", C.c_int), ("keys", C.c_void_p), ("mp_valid", C.c_void_p), ("outlier", C.c_void_p), ("mp_world", C.c_void_p),
                ("Tcw", C.c_float * 7), ("nlevels", C.c_int), ("levels", C.POINTER(C.c_void_p)), ("level_w", C.c_void_p),
                ("level_h", C.c_void_p)]


# EuRoC cam0 intrinsics (reference Examples/Monocular/EuRoC.yaml:8-11), used by bench.py / tests
EUROC = dict(fx=458.654, fy=457.296, cx=367.215, cy=248.375)


def make_camera(w, h, fx=EUROC["fx"], fy=EUROC["fy"], cx=EUROC["cx"], cy=EUROC["cy"], mb=0.0, mbf=0.0):
    return Camera(fx, fy, cx, cy, mb, mbf, 0.0, 0.0, float(w), float(h))


_lib = None


def host_stream_probe(device, threads, bytes_per_thread, seconds):
    """GB/s that `threads` host threads (bound to the device's NUMA node) stream-read from page-locked buffers; no GPU work (include/ygzf.h)."""
    L = load_library()
    L.ygzf_host_stream_probe.argtypes = [C.c_int, C.c_int, C.c_size_t, C.c_double, C.POINTER(C.c_double)]
    out = C.c_double(0.0)
    rc = L.ygzf_host_stream_probe(int(device), int(threads), int(bytes_per_thread), float(seconds), C.byref(out))
    if rc != 0:
        raise YgzfError("ygzf_host_stream_probe failed (%d)" % rc)
    return out.value


def force_env(base=None, **kv):
    """YGZF_FORCE string (csrc/ygzf_internal.h: key=value,... pins a plan the library otherwise picks itself; read when a context is created):
    `base` (default: the current environment's) with the given keys set, or removed when their value is None."""
    cur = os.environ.get("YGZF_FORCE", "") if base is None else base
    items = dict(p.split("=", 1) for p in cur.split(",") if "=" in p)
    for k, v in kv.items():
        if v is None:
            items.pop(k, None)
        else:
            items[k] = str(v)
    return ",".join("%s=%s" % kv for kv in items.items())


def load_library(build_if_missing=True):
    """Loads libygzf.so (building it with hipcc first when it is missing/stale).  Fails loudly: there is no fallback."""
    global _lib
    if _lib is not None:
        return _lib
    path = _build.lib_path()
    alt = os.environ.get("YGZF_LIBRARY")   # A/B measurements: another build of the same ABI (tools/ab_libs.sh)
    if alt:
        if not os.path.exists(alt):
            raise YgzfError("YGZF_LIBRARY=%s does not exist" % alt)
        path, build_if_missing = alt, False
    if build_if_missing and _build.needs_build():
        try:
            _build.build()
        except Exception as e:  # stale-but-present library is still usable on a box without hipcc sources changes
            if not os.path.exists(path):
                raise YgzfError("libygzf.so is missing and could not be built: %r" % (e,))
    if not os.path.exists(path):
        raise YgzfError("libygzf.so not found at %s (run python -m orb_ygz_slam_amd.build)" % path)
    L = C.CDLL(path)
    vp, ip, fp = C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_float)
    L.ygzf_create.argtypes = [C.c_int, C.POINTER(ExtractorCfg), C.c_int, C.c_int, C.c_int, C.POINTER(vp)]
    L.ygzf_destroy.argtypes = [vp]
    L.ygzf_destroy.restype = None
    L.ygzf_last_error.argtypes = [vp]
    L.ygzf_last_error.restype = C.c_char_p
    L.ygzf_scale_tables_host.argtypes = [C.POINTER(ExtractorCfg), vp, vp, vp, vp, vp]
    L.ygzf_pyramid_plan_host.argtypes = [C.POINTER(ExtractorCfg), C.c_int, C.c_int, ip, ip, vp, vp, C.c_int]
    L.ygzf_get_levels.argtypes = [vp]
    L.ygzf_get_scale_tables.argtypes = [vp, vp, vp, vp, vp]
    L.ygzf_get_features_per_level.argtypes = [vp, vp]
    L.ygzf_level_size.argtypes = [vp, C.c_int, C.c_int, C.c_int, ip, ip]
    L.ygzf_max_keypoints.argtypes = [vp, C.c_int, C.c_int]
    L.ygzf_compute_pyramid.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, C.POINTER(vp)]
    L.ygzf_extract.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, vp, vp, C.c_int, ip]
    L.ygzf_extract_batch_device.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_size_t]
    L.ygzf_extract_batch_host.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_size_t]
    L.ygzf_batch_counts.argtypes = [vp, vp]
    L.ygzf_batch_fetch.argtypes = [vp, C.c_int, vp, vp, C.c_int, ip]
    L.ygzf_batch_fetch_all.argtypes = [vp, vp, vp, vp, C.c_int]
    L.ygzf_batch_fetch_level.argtypes = [vp, C.c_int, C.c_int, vp]
    L.ygzf_sync.argtypes = [vp]
    L.ygzf_batch_fetch_candidates.argtypes = [vp, C.c_int, C.c_int, vp, vp, vp, C.c_int, ip]
    L.ygzf_batch_fetch_level_keypoints.argtypes = [vp, C.c_int, C.c_int, vp, vp, vp, C.c_int, ip]
    L.ygzf_descriptor_distance.argtypes = [vp, vp, vp, C.c_int, vp]
    L.ygzf_match_batch_prev.argtypes = [vp, C.POINTER(Camera), C.c_float, C.c_int, C.c_int, C.c_int]
    L.ygzf_match_counts.argtypes = [vp, vp]
    L.ygzf_match_fetch.argtypes = [vp, C.c_int, vp, vp, C.c_int]
    L.ygzf_search_by_projection_last.argtypes = [vp, C.POINTER(FrameView), C.POINTER(Camera), C.c_int, vp, vp, vp, vp, vp, vp, vp, vp,
                                                 vp, vp, C.c_float, C.c_int, C.c_int, C.c_int, vp, vp, ip]
    L.ygzf_search_by_projection_mappoints.argtypes = [vp, C.POINTER(FrameView), C.POINTER(Camera), C.c_int] + [vp] * 9 + [
        C.c_float, C.c_int, C.c_float, vp, vp, ip]
    L.ygzf_search_by_projection_kf.argtypes = [vp, C.POINTER(FrameView), C.POINTER(Camera), C.c_int] + [vp] * 6 + [
        C.c_float, C.c_int, C.c_int, vp, vp, ip]
    L.ygzf_search_for_initialization.argtypes = [vp, C.POINTER(FrameView), C.POINTER(FrameView), C.POINTER(Camera), vp, C.c_int, C.c_float, C.c_int,
                                                 vp, ip]
    L.ygzf_search_by_bow.argtypes = [vp, C.c_int, vp, vp, vp, vp, C.c_int, vp, vp, vp, C.c_int, vp, vp, C.c_float, C.c_int, vp, ip]
    L.ygzf_search_for_triangulation.argtypes = [vp, C.c_int, vp, vp, vp, vp, C.POINTER(FrameView), vp, C.POINTER(FrameView), vp, vp, vp, vp, vp, vp,
                                                C.POINTER(Camera), C.c_int, C.c_int, vp, ip]
    L.ygzf_compute_stereo_matches.argtypes = [vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, C.c_int, vp, vp, C.c_float, C.c_float, vp, vp]
    L.ygzf_stereo_batch.argtypes = [vp, C.c_float, C.c_float]
    L.ygzf_stereo_fetch.argtypes = [vp, C.c_int, vp, vp, C.c_int]
    L.ygzf_image_cache_reserve.argtypes = [vp, C.c_int, C.c_int, C.c_int]
    L.ygzf_image_cache_put.argtypes = [vp, C.c_int, vp, C.c_int, C.c_int, C.c_int]
    L.ygzf_image_cache_put_resident.argtypes = [vp, C.c_int, vp]
    L.ygzf_has_resident_image.argtypes = [vp, C.c_int, C.c_int]
    L.ygzf_find_direct_projection_batch.argtypes = [vp, C.POINTER(Camera), C.c_int, vp, C.c_int, vp, vp, vp, vp, vp, vp, vp, vp]
    L.ygzf_vocabulary_set.argtypes = [vp, C.c_int, C.c_int, vp, vp]
    L.ygzf_bow_transform.argtypes = [vp, C.c_int, vp, C.c_int, vp, vp]
    L.ygzf_predict_scale_steps.argtypes = [C.c_float, C.c_int, vp]
    L.ygzf_is_in_frustum_batch.argtypes = [vp, C.POINTER(Camera), C.c_int, C.c_int, C.POINTER(FrustumIn), vp, vp, vp, vp, vp, vp]
    L.ygzf_search_local_points.argtypes = [vp, C.POINTER(FrameView), C.POINTER(Camera), C.c_int, C.POINTER(FrustumIn), vp, vp, C.c_float, C.c_int,
                                           C.c_float, vp, vp, ip, vp, vp, vp, vp, vp, vp]
    L.ygzf_distinctive_descriptors_batch.argtypes = [vp, C.c_int, vp, vp, vp]
    L.ygzf_extract_resident.argtypes = [vp, vp, vp, C.c_int, vp]
    L.ygzf_device_mem_info.argtypes = [vp, vp, vp]
    L.ygzf_set_fast_plan.argtypes = [vp, C.c_int]
    L.ygzf_get_fast_plan.argtypes = [vp, vp]
    L.ygzf_set_fast_kernel.argtypes = [vp, C.c_int]
    L.ygzf_set_extract_ahead.argtypes = [vp, C.c_int]
    L.ygzf_set_carry_previous.argtypes = [vp, C.c_int]
    L.ygzf_features_in_area.argtypes = [vp, vp, C.c_int, vp, C.c_int, vp, vp, C.c_int, vp, vp]
    L.ygzf_sia_run.argtypes = [vp, C.POINTER(SiaFrame), C.POINTER(SiaFrame), C.POINTER(Camera), vp, C.c_int, C.c_int, C.c_int, vp,
                               C.POINTER(C.c_size_t), vp, vp]
    L.ygzf_sia_run_cached.argtypes = [vp, C.c_int, C.c_int, C.POINTER(SiaFrame), vp, C.POINTER(Camera), vp, C.c_int, C.c_int, C.c_int, vp,
                                      C.POINTER(C.c_size_t), vp, vp]
    L.ygzf_align_batch_prev.argtypes = [vp, C.POINTER(Camera), C.c_int, C.c_int, C.c_int]
    L.ygzf_align_fetch.argtypes = [vp, C.c_int, vp, C.POINTER(C.c_size_t), vp]
    L.ygzf_fast10.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, vp, C.c_int, ip, ip]
    L.ygzf_describe_keys.argtypes = [vp, C.c_int, vp, C.c_int, C.c_int, vp, vp]
    L.ygzf_extract_dso.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, vp, C.c_int, C.c_int, vp, ip, ip]
    L.ygzf_extract_dso_multilevel.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, vp, C.c_int, C.c_int, vp, ip, ip]
    L.ygzf_extract_fast_keypoint.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, vp, C.c_int, C.c_int, vp, ip]
    L.ygzf_timer_start.argtypes = [vp]
    L.ygzf_timer_stop.argtypes = [vp, fp]
    L.ygzf_mgpu_create.argtypes = [ip, C.c_int, C.POINTER(ExtractorCfg), C.c_int, C.c_int, C.c_int, C.POINTER(vp)]
    L.ygzf_mgpu_destroy.argtypes = [vp]
    L.ygzf_mgpu_destroy.restype = None
    L.ygzf_mgpu_last_error.argtypes = [vp]
    L.ygzf_mgpu_last_error.restype = C.c_char_p
    L.ygzf_mgpu_device_count.argtypes = [vp]
    L.ygzf_mgpu_keypoint_stride.argtypes = [vp]
    L.ygzf_mgpu_slot_of_frame.argtypes = [vp, C.c_int, C.c_int]
    L.ygzf_mgpu_extract_match.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_size_t, C.c_int, C.POINTER(Camera), C.c_float, C.c_int, C.c_int,
                                          C.c_int, vp, vp, vp, C.c_int, vp, vp]
    L.ygzf_profile_enable.argtypes = [vp, C.c_int]
    L.ygzf_profile_read.argtypes = [vp, C.POINTER(C.c_char_p), fp, ip, C.c_int]
    L.ygzf_profile_reset.argtypes = [vp]
    L.ygzf_stream.argtypes = [vp]
    L.ygzf_stream.restype = vp
    _lib = L
    return L


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def bind_host_thread_to_device(device):
    """ygzf_bind_host_thread_to_device: the calling thread onto the CPUs of the device's NUMA node; number of CPUs bound to (0: nothing done)."""
    return load_library().ygzf_bind_host_thread_to_device(int(device))


def host_row_pitch(w):
    """ygzf_host_row_pitch: the row pitch at which host frames take the whole-frame upload."""
    return load_library().ygzf_host_row_pitch(int(w))


def pyramid_plan_host(nfeatures, scale_factor, nlevels, w, h):
    """ygzf_pyramid_plan_host: how ComputePyramid of one w x h frame is launched (host arithmetic only, no device).  Returns a dict with
    `strips` (0 = one launch per level), `base` (the level the strips stage: 0 = the image), `lds_bytes`, `levels` = [(w, h)] and `rows`, an int
    array [strips, nlevels, 4] of (ca, cb, wa, wb)."""
    L = load_library()
    cfg = ExtractorCfg(nfeatures, scale_factor, nlevels, 20, 7, 0)
    n, lds = C.c_int(0), C.c_int(0)
    wh = np.zeros(2 * nlevels, np.int32)
    rows = np.zeros(64 * nlevels * 4, np.uint16)
    rc = L.ygzf_pyramid_plan_host(C.byref(cfg), w, h, C.byref(n), C.byref(lds), _p(wh), _p(rows), rows.size)
    if rc != 0:
        raise YgzfError("ygzf_pyramid_plan_host failed (%d)" % rc)
    L.ygzf_pyramid_plan_base_host.argtypes = [C.c_void_p, C.c_int, C.c_int]
    base = L.ygzf_pyramid_plan_base_host(C.byref(cfg), w, h)
    if base < 0:
        raise YgzfError("ygzf_pyramid_plan_base_host failed (%d)" % base)
    return {"strips": n.value, "base": base, "lds_bytes": lds.value, "levels": [(int(wh[2 * l]), int(wh[2 * l + 1])) for l in range(nlevels)],
            "rows": rows[:n.value * nlevels * 4].astype(np.int64).reshape(n.value, nlevels, 4)}


class Extractor:
    """Thin object wrapper over a ygzf_ctx (mirrors ygz::ORBextractor's constructor arguments)."""

    def __init__(self, nfeatures=1000, scale_factor=1.2, nlevels=8, ini_th=20, min_th=7, max_width=752, max_height=480,
                 max_batch=1, device=0, cv_mode=CV_LEGACY_SSE2):
        self.L = load_library()
        self.nlevels = nlevels
        self.cfg = ExtractorCfg(nfeatures, scale_factor, nlevels, ini_th, min_th, cv_mode)
        self._wh = (max_width, max_height, 0)   # (w, h, frames) of the last extracted batch
        self.max_width, self.max_height = max_width, max_height
        self._inflight = []                     # host arrays handed to asynchronous uploads, kept alive until the next synchronisation
        self.h = C.c_void_p()
        rc = self.L.ygzf_create(device, C.byref(self.cfg), max_width, max_height, max_batch, C.byref(self.h))
        if rc != 0:
            self.h = C.c_void_p()
            raise YgzfError("ygzf_create failed (%d): %s" % (rc, self.L.ygzf_last_error(None).decode()))

    def close(self):
        if getattr(self, "h", None) and self.h.value:
            self.L.ygzf_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc):
        if rc < 0:
            raise YgzfError("libygzf error %d: %s" % (rc, self.L.ygzf_last_error(self.h).decode()))
        return rc

    def tables(self):
        n = self.nlevels
        sc, inv, s2, is2 = (np.zeros(n, np.float32) for _ in range(4))
        nf = np.zeros(n, np.int32)
        self._ck(self.L.ygzf_get_scale_tables(self.h, _p(sc), _p(inv), _p(s2), _p(is2)))
        self._ck(self.L.ygzf_get_features_per_level(self.h, _p(nf)))
        return dict(scale=sc, inv_scale=inv, sigma2=s2, inv_sigma2=is2, nfeat=nf)

    def level_size(self, w, h, level):
        lw, lh = C.c_int(), C.c_int()
        self._ck(self.L.ygzf_level_size(self.h, w, h, level, C.byref(lw), C.byref(lh)))
        return lw.value, lh.value

    def max_keypoints(self, w, h):
        return self._ck(self.L.ygzf_max_keypoints(self.h, w, h))

    def compute_pyramid(self, img):
        img = np.ascontiguousarray(img, np.uint8)
        h, w = img.shape
        outs = [np.zeros(self.level_size(w, h, l)[::-1], np.uint8) for l in range(self.nlevels)]
        arr = (C.c_void_p * self.nlevels)(*[o.ctypes.data for o in outs])
        self._ck(self.L.ygzf_compute_pyramid(self.h, _p(img), w, h, w, arr))
        return outs

    def extract(self, img):
        img = np.asarray(img, np.uint8)
        if img.ndim != 2 or img.strides[1] != 1 or img.strides[0] < img.shape[1]:
            img = np.ascontiguousarray(img)
        h, w = img.shape                                   # a row-strided view (e.g. a crop) goes down as it is: stride = its row pitch
        cap = self.max_keypoints(w, h)
        k = np.zeros(max(cap, 1), KP_DTYPE)
        d = np.zeros((max(cap, 1), 32), np.uint8)
        n = C.c_int()
        self._ck(self.L.ygzf_extract(self.h, img.ctypes.data_as(C.c_void_p), w, h, int(img.strides[0]), _p(k), _p(d), cap, C.byref(n)))
        return k[:n.value].copy(), d[:n.value].copy()

    def extract_resident(self, w, h):
        """Keypoints + descriptors of the image whose pyramid the previous compute_pyramid left on the device (no second upload)."""
        cap = self.max_keypoints(w, h)
        k = np.zeros(max(cap, 1), KP_DTYPE)
        d = np.zeros((max(cap, 1), 32), np.uint8)
        n = C.c_int()
        self._ck(self.L.ygzf_extract_resident(self.h, _p(k), _p(d), cap, C.byref(n)))
        return k[:n.value].copy(), d[:n.value].copy()

    def extract_batch_host(self, imgs):
        """(n, h, w) uint8 frames; a view with padded rows / frames (unit stride along x) is handed over as it lies -- e.g. a [:, :, :w] view of
        frames allocated at ygzf_host_row_pitch(w), which take the long-row upload."""
        if not (imgs.dtype == np.uint8 and imgs.ndim == 3 and imgs.strides[2] == 1 and imgs.strides[1] >= imgs.shape[2] and
                (imgs.shape[0] == 1 or imgs.strides[0] >= imgs.strides[1] * imgs.shape[1])):
            imgs = np.ascontiguousarray(imgs, np.uint8)
        n, h, w = imgs.shape
        self._ck(self.L.ygzf_extract_batch_host(self.h, C.c_void_p(imgs.ctypes.data), n, w, h, imgs.strides[1], imgs.strides[0]))
        self._inflight.append(imgs)   # the H2D copy is asynchronous: the frames must stay alive until the next synchronising call
        self._wh = (w, h, n)

    def extract_batch_host_frames(self, frames, row_pitch=None):
        """ygzf_extract_batch_host_frames: a list of 2-D uint8 arrays (or views with a common row pitch) of one size, anywhere in host memory."""
        frames = [f if (f.dtype == np.uint8 and f.strides[1] == 1) else np.ascontiguousarray(f, np.uint8) for f in frames]
        h, w = frames[0].shape
        pitch = row_pitch or frames[0].strides[0]
        assert all(f.shape == (h, w) and f.strides[0] == pitch for f in frames)
        ptrs = (C.c_void_p * len(frames))(*[f.ctypes.data for f in frames])
        self.L.ygzf_extract_batch_host_frames.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]
        self._ck(self.L.ygzf_extract_batch_host_frames(self.h, ptrs, len(frames), w, h, pitch))
        self._inflight.append(frames)
        self._wh = (w, h, len(frames))

    def extract_batch_device(self, dptr, n, w, h, row_pitch=None, frame_stride=None):
        row_pitch = row_pitch or w
        frame_stride = frame_stride or row_pitch * h
        self._ck(self.L.ygzf_extract_batch_device(self.h, C.c_void_p(dptr), n, w, h, row_pitch, frame_stride))
        self._wh = (w, h, n)

    def sync(self):
        self._ck(self.L.ygzf_sync(self.h))
        self._inflight.clear()

    def batch_counts(self):
        n = np.zeros(self._wh[2], np.int32)
        self._ck(self.L.ygzf_batch_counts(self.h, _p(n)))
        self._inflight.clear()
        return n

    def batch_fetch(self, frame):
        w, h, _ = self._wh
        cap = self.max_keypoints(w, h)
        k = np.zeros(max(cap, 1), KP_DTYPE)
        d = np.zeros((max(cap, 1), 32), np.uint8)
        n = C.c_int()
        self._ck(self.L.ygzf_batch_fetch(self.h, frame, _p(k), _p(d), cap, C.byref(n)))
        self._inflight.clear()
        return k[:n.value].copy(), d[:n.value].copy()

    def batch_fetch_all(self, n_frames, out=None):
        """All frames' keypoints / descriptors with one synchronisation -> (kps (B, stride), desc (B, stride, 32), counts)."""
        w, h, _ = self._wh
        stride = max(self.max_keypoints(w, h), 1)
        if out is None:
            out = (np.zeros((n_frames, stride), KP_DTYPE), np.zeros((n_frames, stride, 32), np.uint8), np.zeros(n_frames, np.int32))
        k, d, n = out
        self._ck(self.L.ygzf_batch_fetch_all(self.h, _p(k), _p(d), _p(n), stride))
        self._inflight.clear()
        return k, d, n

    def batch_fetch_packed(self):
        """ygzf_batch_fetch_packed: the last batch's counts / keypoint rows / descriptor rows gathered on the device and brought over in ONE copy;
        returns (n_kp[B], kps[B, row], desc[B, row, 32]) as views of one host block."""
        B = self._last_frames()
        cap = 512 + B * (self._kp_stride_max() * (KP_DTYPE.itemsize + 32) + 512)
        host = np.zeros(cap, np.uint8)
        ok, od, by = C.c_size_t(0), C.c_size_t(0), C.c_size_t(0)
        re = C.c_int(0)
        self.L.ygzf_batch_fetch_packed.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        self._ck(self.L.ygzf_batch_fetch_packed(self.h, host.ctypes.data_as(C.c_void_p), cap, C.byref(ok), C.byref(od), C.byref(re), C.byref(by)))
        r = re.value
        n = host[:4 * B].view(np.int32).copy()
        kps = host[ok.value:ok.value + B * r * KP_DTYPE.itemsize].view(KP_DTYPE).reshape(B, r)
        desc = host[od.value:od.value + B * r * 32].reshape(B, r, 32)
        return n, kps, desc

    def _last_frames(self):
        return len(self.batch_counts())

    def _kp_stride_max(self):
        return self.max_keypoints(self.max_width, self.max_height)

    def batch_fetch_level(self, frame, level):
        w, h, _ = self._wh
        lw, lh = self.level_size(w, h, level)
        out = np.zeros((lh, lw), np.uint8)
        self._ck(self.L.ygzf_batch_fetch_level(self.h, frame, level, _p(out)))
        return out

    def batch_fetch_candidates(self, frame, level, cap=1 << 20):
        xs, ys, sc = (np.zeros(cap, np.int32) for _ in range(3))
        n = C.c_int()
        self._ck(self.L.ygzf_batch_fetch_candidates(self.h, frame, level, _p(xs), _p(ys), _p(sc), cap, C.byref(n)))
        return xs[:n.value].copy(), ys[:n.value].copy(), sc[:n.value].copy()

    def batch_fetch_level_keypoints(self, frame, level, cap=1 << 16):
        xs, ys, sc = (np.zeros(cap, np.int32) for _ in range(3))
        n = C.c_int()
        self._ck(self.L.ygzf_batch_fetch_level_keypoints(self.h, frame, level, _p(xs), _p(ys), _p(sc), cap, C.byref(n)))
        return xs[:n.value].copy(), ys[:n.value].copy(), sc[:n.value].copy()

    def descriptor_distance(self, a, b):
        a = np.ascontiguousarray(a, np.uint8).reshape(-1, 32)
        b = np.ascontiguousarray(b, np.uint8).reshape(-1, 32)
        out = np.zeros(len(a), np.int32)
        self._ck(self.L.ygzf_descriptor_distance(self.h, _p(a), _p(b), len(a), _p(out)))
        return out

    def match_batch_prev(self, cam, th=15.0, mono=True, check_level=True, check_ori=True):
        self._ck(self.L.ygzf_match_batch_prev(self.h, C.byref(cam), th, int(mono), int(check_level), int(check_ori)))

    def match_counts(self):
        n = np.zeros(self._wh[2], np.int32)
        self._ck(self.L.ygzf_match_counts(self.h, _p(n)))
        self._inflight.clear()
        return n

    def match_fallbacks(self):
        """pairs whose in-order resolution fell back to the one-wave serial pass since the context was created (include/ygzf.h)"""
        n = C.c_uint(0)
        self.L.ygzf_match_fallbacks.argtypes = [C.c_void_p, C.c_void_p]
        self._ck(self.L.ygzf_match_fallbacks(self.h, C.byref(n)))
        return n.value

    def match_fetch(self, frame):
        w, h, _ = self._wh
        cap = self.max_keypoints(w, h)
        m = np.zeros(max(cap, 1), np.int32)
        o = np.zeros(max(cap, 1), np.uint8)
        self._ck(self.L.ygzf_match_fetch(self.h, frame, _p(m), _p(o), cap))
        self._inflight.clear()
        return m, o

    def search_by_projection_last(self, cam, cur_keys, cur_desc, last_keys, mp_world, mp_desc, Rcw, tcw, Rlw, tlw, th, mono=True,
                                  check_level=True, check_ori=True, mp_valid=None, outlier=None, mp_has_obs=None, u_right=None,
                                  cur_owner=None, scale_factors=None):
        """ORBmatcher::SearchByProjection(Cur, Last, th, bMono, checkLevel) on host arrays -> (nmatches, cur_match, cur_owner)."""
        ck = np.ascontiguousarray(cur_keys, KP_DTYPE)
        cd = np.ascontiguousarray(cur_desc, np.uint8)
        lk = np.ascontiguousarray(last_keys, KP_DTYPE)
        mw = np.ascontiguousarray(mp_world, np.float32)
        md = np.ascontiguousarray(mp_desc, np.uint8)
        keep = [ck, cd, lk, mw, md]

        def opt(a, dt):
            if a is None:
                return None
            a = np.ascontiguousarray(a, dt)
            keep.append(a)
            return _p(a)
        fv = FrameView(len(ck), ck.ctypes.data, cd.ctypes.data, None, None, self.nlevels)
        ur = opt(u_right, np.float32)
        if ur is not None:
            fv.u_right = ur
        sf = opt(scale_factors, np.float32)
        if sf is not None:
            fv.scale_factors = sf
        owner = np.zeros(max(len(ck), 1), np.uint8) if cur_owner is None else np.array(cur_owner, np.uint8)
        match = np.full(max(len(ck), 1), -1, np.int32)
        mats = [np.ascontiguousarray(a, np.float32) for a in (Rcw, tcw, Rlw, tlw)]
        n = C.c_int()
        self._ck(self.L.ygzf_search_by_projection_last(self.h, C.byref(fv), C.byref(cam), len(lk), _p(lk), opt(mp_valid, np.uint8),
                                                       opt(outlier, np.uint8), opt(mp_has_obs, np.uint8), _p(mw), _p(md), _p(mats[0]),
                                                       _p(mats[1]), _p(mats[2]), _p(mats[3]), th, int(mono), int(check_level),
                                                       int(check_ori), _p(owner), _p(match), C.byref(n)))
        return n.value, match[:len(ck)], owner[:len(ck)]

    def search_by_projection_mappoints(self, cam, keys, desc, track_in_view, proj_x, proj_y, view_cos, scale_level, mp_desc, th,
                                       check_level=True, nnratio=0.8, is_bad=None, mp_has_obs=None, proj_xr=None, u_right=None,
                                       owner=None, scale_factors=None):
        """ORBmatcher::SearchByProjection(F, MapPoints, th, checkLevel) on host arrays -> (nmatches, match, owner)."""
        ck = np.ascontiguousarray(keys, KP_DTYPE)
        cd = np.ascontiguousarray(desc, np.uint8)
        keep = [ck, cd]

        def arr(a, dt):
            if a is None:
                return None
            a = np.ascontiguousarray(a, dt)
            keep.append(a)
            return _p(a)
        fv = FrameView(len(ck), ck.ctypes.data, cd.ctypes.data, None, None, self.nlevels)
        ur = arr(u_right, np.float32)
        if ur is not None:
            fv.u_right = ur
        sf = arr(scale_factors, np.float32)
        if sf is not None:
            fv.scale_factors = sf
        own = np.zeros(max(len(ck), 1), np.uint8) if owner is None else np.array(owner, np.uint8)
        match = np.full(max(len(ck), 1), -1, np.int32)
        n = C.c_int()
        self._ck(self.L.ygzf_search_by_projection_mappoints(
            self.h, C.byref(fv), C.byref(cam), len(proj_x), arr(track_in_view, np.uint8), arr(is_bad, np.uint8), arr(mp_has_obs, np.uint8),
            arr(proj_x, np.float32), arr(proj_y, np.float32), arr(proj_xr, np.float32), arr(view_cos, np.float32),
            arr(scale_level, np.int32), arr(mp_desc, np.uint8), th, int(check_level), nnratio, _p(own), _p(match), C.byref(n)))
        return n.value, match[:len(ck)], own[:len(ck)]

    def search_by_projection_kf(self, cam, keys, desc, valid, proj_x, proj_y, pred_level, kf_angle, mp_desc, th, orb_dist, check_ori=True,
                                owner=None, scale_factors=None):
        """ORBmatcher::SearchByProjection(Cur, KF, found, th, ORBdist), device part, on host arrays -> (nmatches, match, owner)."""
        ck = np.ascontiguousarray(keys, KP_DTYPE)
        cd = np.ascontiguousarray(desc, np.uint8)
        keep = [ck, cd]

        def arr(a, dt):
            a = np.ascontiguousarray(a, dt)
            keep.append(a)
            return _p(a)
        fv = FrameView(len(ck), ck.ctypes.data, cd.ctypes.data, None, None, self.nlevels)
        if scale_factors is not None:
            fv.scale_factors = arr(scale_factors, np.float32)
        own = np.zeros(max(len(ck), 1), np.uint8) if owner is None else np.array(owner, np.uint8)
        match = np.full(max(len(ck), 1), -1, np.int32)
        n = C.c_int()
        self._ck(self.L.ygzf_search_by_projection_kf(
            self.h, C.byref(fv), C.byref(cam), len(proj_x), arr(valid, np.uint8), arr(proj_x, np.float32), arr(proj_y, np.float32),
            arr(pred_level, np.int32), arr(kf_angle, np.float32), arr(mp_desc, np.uint8), th, orb_dist, int(check_ori), _p(own), _p(match),
            C.byref(n)))
        return n.value, match[:len(ck)], own[:len(ck)]

    def search_for_initialization(self, cam, keys1, desc1, keys2, desc2, prev_matched_xy, window=100, nnratio=0.9, check_ori=True,
                                  scale_factors=None):
        """ORBmatcher::SearchForInitialization -> (nmatches, matches12, updated prev_matched_xy)."""
        k1, k2 = np.ascontiguousarray(keys1, KP_DTYPE), np.ascontiguousarray(keys2, KP_DTYPE)
        d1, d2 = np.ascontiguousarray(desc1, np.uint8), np.ascontiguousarray(desc2, np.uint8)
        f1 = FrameView(len(k1), k1.ctypes.data, d1.ctypes.data, None, None, self.nlevels)
        f2 = FrameView(len(k2), k2.ctypes.data, d2.ctypes.data, None, None, self.nlevels)
        sf = None
        if scale_factors is not None:
            sf = np.ascontiguousarray(scale_factors, np.float32)
            f1.scale_factors = f2.scale_factors = _p(sf)
        pm = np.array(prev_matched_xy, np.float32).reshape(-1, 2).copy()
        m12 = np.full(max(len(k1), 1), -1, np.int32)
        n = C.c_int()
        self._ck(self.L.ygzf_search_for_initialization(self.h, C.byref(f1), C.byref(f2), C.byref(cam), _p(pm), int(window), nnratio, int(check_ori),
                                                       _p(m12), C.byref(n)))
        return n.value, m12[:len(k1)], pm

    def search_by_bow(self, kf_off, kf_idx, f_off, f_idx, kf_valid, kf_keys, kf_desc, f_keys, f_desc, nnratio=0.7, check_ori=True):
        """ORBmatcher::SearchByBoW(KF, F) on a joined node list -> (nmatches, match per Frame feature)."""
        ko, ki, fo, fi = (np.ascontiguousarray(a, np.int32) for a in (kf_off, kf_idx, f_off, f_idx))
        kv = np.ascontiguousarray(kf_valid, np.uint8)
        kk, fk = np.ascontiguousarray(kf_keys, KP_DTYPE), np.ascontiguousarray(f_keys, KP_DTYPE)
        kd, fd = np.ascontiguousarray(kf_desc, np.uint8), np.ascontiguousarray(f_desc, np.uint8)
        match = np.full(max(len(fk), 1), -1, np.int32)
        n = C.c_int()
        self._ck(self.L.ygzf_search_by_bow(self.h, len(ko) - 1, _p(ko), _p(ki), _p(fo), _p(fi), len(kk), _p(kv), _p(kk), _p(kd), len(fk), _p(fk),
                                           _p(fd), nnratio, int(check_ori), _p(match), C.byref(n)))
        return n.value, match[:len(fk)]

    def search_for_triangulation(self, off1, idx1, off2, idx2, kf1, kf2, scale_factors2, level_sigma2_2, F12, Cw1, R2w, t2w, cam2,
                                 only_stereo=False, check_ori=True):
        """ORBmatcher::SearchForTriangulation on a joined node list.  kf1 / kf2: dicts with keys (KP_DTYPE), desc, has_mp (bytes), u_right
        (floats or None); scale_factors2 / level_sigma2_2: KF2's tables or None (the context's); cam2 = (fx, fy, cx, cy) of KF2
        -> (nmatches, match12 per KF1 feature: KF2 feature, -1 none, -2 culled by the rotation check)."""
        o1, i1, o2, i2 = (np.ascontiguousarray(a, np.int32) for a in (off1, idx1, off2, idx2))
        keep = []

        def side(kf):
            k, d = np.ascontiguousarray(kf["keys"], KP_DTYPE), np.ascontiguousarray(kf["desc"], np.uint8)
            m = np.ascontiguousarray(kf["has_mp"], np.uint8)
            u = None if kf.get("u_right") is None else np.ascontiguousarray(kf["u_right"], np.float32)
            keep.extend([k, d, m, u])
            return FrameView(len(k), k.ctypes.data, d.ctypes.data, None if u is None else u.ctypes.data, None, self.nlevels), m

        f1, m1 = side(kf1)
        f2, m2 = side(kf2)
        sf = sg = None
        if scale_factors2 is not None:
            sf = np.ascontiguousarray(scale_factors2, np.float32)
            f2.scale_factors, f2.nlevels = sf.ctypes.data, len(sf)
        if level_sigma2_2 is not None:
            sg = np.ascontiguousarray(level_sigma2_2, np.float32)
        Fm, Cw, Rm, tm = (np.ascontiguousarray(a, np.float32).reshape(-1) for a in (F12, Cw1, R2w, t2w))
        cam = Camera(float(cam2[0]), float(cam2[1]), float(cam2[2]), float(cam2[3]), 0.0, 0.0, 0.0, 0.0, 0.0, 0.0)
        match = np.full(max(f1.n, 1), -1, np.int32)
        n = C.c_int()
        self._ck(self.L.ygzf_search_for_triangulation(self.h, len(o1) - 1, _p(o1), _p(i1), _p(o2), _p(i2), C.byref(f1), _p(m1), C.byref(f2), _p(m2),
                                                      None if sg is None else _p(sg), _p(Fm), _p(Cw), _p(Rm), _p(tm), C.byref(cam), int(only_stereo),
                                                      int(check_ori), _p(match), C.byref(n)))
        return n.value, match[:f1.n]

    @staticmethod
    def _frustum_in(world, normal, max_dist_inv, min_dist_inv, mf_max_distance, Rcw, tcw, Ow, log_scale_factor, viewing_cos_limit, candidate, keep):
        fi = FrustumIn()
        for name, arr, dt in (("world", world, np.float32), ("normal", normal, np.float32), ("max_dist_inv", max_dist_inv, np.float32),
                              ("min_dist_inv", min_dist_inv, np.float32), ("mf_max_distance", mf_max_distance, np.float32),
                              ("candidate", candidate, np.uint8)):
            if arr is None:
                continue
            a = np.ascontiguousarray(arr, dt)
            keep.append(a)
            setattr(fi, name, a.ctypes.data)
        fi.Rcw = (C.c_float * 9)(*np.asarray(Rcw, np.float32).ravel())
        fi.tcw = (C.c_float * 3)(*np.asarray(tcw, np.float32).ravel())
        fi.Ow = (C.c_float * 3)(*np.asarray(Ow, np.float32).ravel())
        fi.log_scale_factor = float(log_scale_factor)
        fi.viewing_cos_limit = float(viewing_cos_limit)
        return fi

    def is_in_frustum_batch(self, cam, world, normal, max_dist_inv, min_dist_inv, mf_max_distance, Rcw, tcw, Ow, log_scale_factor,
                            viewing_cos_limit=0.5, candidate=None):
        """Frame::isInFrustum over a MapPoint batch -> (in_view, projX, projY, projXR, level, viewCos)."""
        keep = []
        fi = self._frustum_in(world, normal, max_dist_inv, min_dist_inv, mf_max_distance, Rcw, tcw, Ow, log_scale_factor, viewing_cos_limit,
                              candidate, keep)
        n = len(keep[2])
        iv = np.zeros(max(n, 1), np.uint8)
        px, py, pxr, vc = (np.zeros(max(n, 1), np.float32) for _ in range(4))
        lv = np.zeros(max(n, 1), np.int32)
        self._ck(self.L.ygzf_is_in_frustum_batch(self.h, C.byref(cam), self.nlevels, n, C.byref(fi), _p(iv), _p(px), _p(py), _p(pxr), _p(lv), _p(vc)))
        return iv[:n], px[:n], py[:n], pxr[:n], lv[:n], vc[:n]

    def search_local_points(self, cam, keys, desc, world, normal, max_dist_inv, min_dist_inv, mf_max_distance, Rcw, tcw, Ow, log_scale_factor,
                            mp_desc, th, check_level=False, nnratio=0.8, viewing_cos_limit=0.5, candidate=None, mp_has_obs=None, owner=None,
                            scale_factors=None):
        """isInFrustum + SearchByProjection(F, MapPoints) fused on the device -> (nmatches, match, owner, in_view)."""
        ck = np.ascontiguousarray(keys, KP_DTYPE)
        cd = np.ascontiguousarray(desc, np.uint8)
        keep = [ck, cd]
        fi = self._frustum_in(world, normal, max_dist_inv, min_dist_inv, mf_max_distance, Rcw, tcw, Ow, log_scale_factor, viewing_cos_limit,
                              candidate, keep)
        fv = FrameView(len(ck), ck.ctypes.data, cd.ctypes.data, None, None, self.nlevels)
        if scale_factors is not None:
            sf = np.ascontiguousarray(scale_factors, np.float32)
            keep.append(sf)
            fv.scale_factors = _p(sf)
        md = np.ascontiguousarray(mp_desc, np.uint8)
        n = len(md)
        obs = None if mp_has_obs is None else np.ascontiguousarray(mp_has_obs, np.uint8)
        own = np.zeros(max(len(ck), 1), np.uint8) if owner is None else np.array(owner, np.uint8)
        match = np.full(max(len(ck), 1), -1, np.int32)
        iv = np.zeros(max(n, 1), np.uint8)
        nm = C.c_int()
        self._ck(self.L.ygzf_search_local_points(self.h, C.byref(fv), C.byref(cam), n, C.byref(fi), _p(obs) if obs is not None else None, _p(md), th,
                                                 int(check_level), nnratio, _p(own), _p(match), C.byref(nm), _p(iv), None, None, None, None, None))
        return nm.value, match[:len(ck)], own[:len(ck)], iv[:n]

    def device_mem_info(self):
        """(free, total) bytes of the context's device after draining its stream."""
        f, t = C.c_size_t(0), C.c_size_t(0)
        self._ck(self.L.ygzf_device_mem_info(self.h, C.byref(f), C.byref(t)))
        return f.value, t.value

    def set_fast_plan(self, plan):
        """0 auto (default), 1 one pass at minTh, 2 iniTh first -- same keypoints, different cost (include/ygzf.h)."""
        self._ck(self.L.ygzf_set_fast_plan(self.h, int(plan)))

    def set_fast_kernel(self, kernel):
        """0 auto (default), 1 register staging (k_fast_quads), 2 cell table + LDS-DMA staging (k_fast_tab) -- same results (include/ygzf.h)."""
        self._ck(self.L.ygzf_set_fast_kernel(self.h, int(kernel)))

    def fast_stats(self):
        """(corner-bearing quads per pass-1 run, pass-1 runs per cell) from the last sampled launch of the cell loop (include/ygzf.h)"""
        a, b = C.c_float(0), C.c_float(0)
        self.L.ygzf_get_fast_stats.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        self._ck(self.L.ygzf_get_fast_stats(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def phase_clocks(self, kernel, reset=True):
        """16 u64 counters of the phase-clock build (kernel 0 = k_fast_tab, 1 = k_describe); raises on the product build (include/ygzf.h)."""
        out = np.zeros(16, np.uint64)
        self.L.ygzf_phase_clocks.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]
        self._ck(self.L.ygzf_phase_clocks(self.h, int(kernel), out.ctypes.data_as(C.c_void_p), 1 if reset else 0))
        return out

    def set_extract_ahead(self, on):
        """compute_pyramid also queues the extraction of the same image; extract_resident then only collects it (include/ygzf.h)."""
        self._ck(self.L.ygzf_set_extract_ahead(self.h, 1 if on else 0))

    def set_carry_previous(self, on):
        """off: extractions no longer carry the previous batch's last frame into slot 0 (the batch matchers then raise; include/ygzf.h)."""
        self._ck(self.L.ygzf_set_carry_previous(self.h, 1 if on else 0))

    def fast_plan(self):
        p = C.c_int(0)
        self._ck(self.L.ygzf_get_fast_plan(self.h, C.byref(p)))
        return p.value

    def features_in_area(self, cam, keys, xyr, levels=None, cap=None):
        """Frame::GetFeaturesInArea for a batch of (x, y, r[, minLevel, maxLevel]) queries -> list of index arrays in the reference's order."""
        ck = np.ascontiguousarray(keys, KP_DTYPE)
        q = np.ascontiguousarray(xyr, np.float32).reshape(-1, 3)
        lv = None if levels is None else np.ascontiguousarray(levels, np.int32).reshape(-1, 2)
        cap = max(len(ck), 1) if cap is None else int(cap)
        out = np.full((max(len(q), 1), max(cap, 1)), -1, np.int32)
        cnt = np.zeros(max(len(q), 1), np.int32)
        self._ck(self.L.ygzf_features_in_area(self.h, C.byref(cam), len(ck), _p(ck), len(q), _p(q), None if lv is None else _p(lv), cap,
                                              _p(out), _p(cnt)))
        return [out[i, :min(int(cnt[i]), cap)].copy() for i in range(len(q))], cnt[:len(q)].copy()

    def distinctive_descriptors_batch(self, obs_off, desc):
        """MapPoint::ComputeDistinctiveDescriptors over a MapPoint batch -> winning observation index per point."""
        oo = np.ascontiguousarray(obs_off, np.int32)
        d = np.ascontiguousarray(desc, np.uint8)
        best = np.zeros(max(len(oo) - 1, 1), np.int32)
        self._ck(self.L.ygzf_distinctive_descriptors_batch(self.h, len(oo) - 1, _p(oo), _p(d), _p(best)))
        return best[:len(oo) - 1]

    def sia_run(self, cam, ref_keys, ref_world, ref_Tcw7, ref_pyr, cur_Tcw7, cur_pyr, inv_scale, max_level, min_level, n_iter=10,
                mp_valid=None, outlier=None):
        """SparseImgAlign(max_level, min_level, n_iter).run(ref, cur, TCR) -> (ret, TCR7, info[2], H 6x6)."""
        keep = []

        def frame(keys, world, Tcw7, pyr, valid, outl):
            f = SiaFrame()
            keys = np.ascontiguousarray(keys, KP_DTYPE)
            f.n = len(keys)
            keep.append(keys)
            f.keys = keys.ctypes.data
            if world is not None:
                world = np.ascontiguousarray(world, np.float32)
                keep.append(world)
                f.mp_world = world.ctypes.data
            for name, a in (("mp_valid", valid), ("outlier", outl)):
                if a is not None:
                    a = np.ascontiguousarray(a, np.uint8)
                    keep.append(a)
                    setattr(f, name, a.ctypes.data)
            for i in range(7):
                f.Tcw[i] = float(Tcw7[i])
            pyr = [np.ascontiguousarray(p, np.uint8) for p in pyr]
            lw = np.array([p.shape[1] for p in pyr], np.int32)
            lh = np.array([p.shape[0] for p in pyr], np.int32)
            arr = (C.c_void_p * len(pyr))(*[p.ctypes.data for p in pyr])
            keep.extend([pyr, lw, lh, arr])
            f.nlevels = len(pyr)
            f.levels = arr
            f.level_w, f.level_h = lw.ctypes.data, lh.ctypes.data
            return f
        R = frame(ref_keys, ref_world, ref_Tcw7, ref_pyr, mp_valid, outlier)
        Cf = frame(np.zeros(0, KP_DTYPE), None, cur_Tcw7, cur_pyr, None, None)
        isf = np.ascontiguousarray(inv_scale, np.float32)
        out7 = np.array([0, 0, 0, 1, 0, 0, 0], np.float32)
        info = np.zeros(2, np.float32)
        H = np.zeros(36, np.float32)
        ret = C.c_size_t()
        self._ck(self.L.ygzf_sia_run(self.h, C.byref(R), C.byref(Cf), C.byref(cam), _p(isf), max_level, min_level, n_iter, _p(out7),
                                     C.byref(ret), _p(info), _p(H)))
        return int(ret.value), out7, info, H.reshape(6, 6)

    def sia_run_cached(self, cam, ref_slot, cur_slot, ref_keys, ref_world, ref_Tcw7, cur_Tcw7, inv_scale, max_level, min_level, n_iter=10,
                       mp_valid=None, outlier=None):
        """SparseImgAlign::run on two image-cache slots (image_cache_put) -> (ret, TCR7, info[2], H 6x6)."""
        keys = np.ascontiguousarray(ref_keys, KP_DTYPE)
        world = np.ascontiguousarray(ref_world, np.float32)
        keep = [keys, world]
        f = SiaFrame()
        f.n = len(keys)
        f.keys = keys.ctypes.data
        f.mp_world = world.ctypes.data
        for name, a in (("mp_valid", mp_valid), ("outlier", outlier)):
            if a is not None:
                a = np.ascontiguousarray(a, np.uint8)
                keep.append(a)
                setattr(f, name, a.ctypes.data)
        for i in range(7):
            f.Tcw[i] = float(ref_Tcw7[i])
        ct = np.ascontiguousarray(cur_Tcw7, np.float32)
        isf = np.ascontiguousarray(inv_scale, np.float32)
        out7 = np.array([0, 0, 0, 1, 0, 0, 0], np.float32)
        info = np.zeros(2, np.float32)
        H = np.zeros(36, np.float32)
        ret = C.c_size_t()
        self._ck(self.L.ygzf_sia_run_cached(self.h, int(ref_slot), int(cur_slot), C.byref(f), _p(ct), C.byref(cam), _p(isf), max_level, min_level,
                                            n_iter, _p(out7), C.byref(ret), _p(info), _p(H)))
        return int(ret.value), out7, info, H.reshape(6, 6)

    def align_batch_prev(self, cam, max_level=None, min_level=1, n_iter=10):
        self._ck(self.L.ygzf_align_batch_prev(self.h, C.byref(cam), self.nlevels - 1 if max_level is None else max_level, min_level, n_iter))

    def align_fetch(self, frame):
        T = np.zeros(7, np.float32)
        info = np.zeros(2, np.float32)
        ret = C.c_size_t()
        self._ck(self.L.ygzf_align_fetch(self.h, frame, _p(T), C.byref(ret), _p(info)))
        return int(ret.value), T, info

    def fast10(self, img, barrier, window=None, cap=None):
        """libfast replacement: (xy int16 (n,2), scores, nonmax indices) of fast_corner_detect_10_sse2 / score / nonmax_3x3."""
        img = np.ascontiguousarray(img, np.uint8)
        ih, iw = img.shape
        x0, y0, w, h = window if window is not None else (0, 0, iw, ih)
        cap = cap or w * h
        xy = np.zeros((max(cap, 1), 2), np.int16)
        sc = np.zeros(max(cap, 1), np.int32)
        nm = np.zeros(max(cap, 1), np.int32)
        n, k = C.c_int(), C.c_int()
        self._ck(self.L.ygzf_fast10(self.h, _p(img), iw, ih, iw, x0, y0, w, h, barrier, _p(xy), _p(sc), _p(nm), cap, C.byref(n), C.byref(k)))
        return xy[:n.value].copy(), sc[:n.value].copy(), nm[:k.value].copy()

    def extract_dso(self, img, existing=None, grid_size=-1, cap=None):
        """operator()(Frame*, ..., DSO_KEYPOINT): (keys = existing (angles recomputed) + new level-0 keys, desc, new mnGridSize)."""
        img = np.ascontiguousarray(img, np.uint8)
        h, w = img.shape
        existing = np.zeros(0, KP_DTYPE) if existing is None else np.ascontiguousarray(existing, KP_DTYPE)
        g0 = grid_size if grid_size > 0 else max(1, int(np.sqrt(h * w / max(self.cfg.nfeatures if hasattr(self, 'cfg') else self.nfeatures, 1))))
        gm = max(1, min(7, g0))
        cap = cap or (len(existing) + 3 * (w // gm) * (h // gm) + 16)
        k = np.zeros(cap, KP_DTYPE)
        k[:len(existing)] = existing
        d = np.zeros((cap, 32), np.uint8)
        g, n = C.c_int(grid_size), C.c_int()
        self._ck(self.L.ygzf_extract_dso(self.h, _p(img), w, h, w, _p(k), len(existing), cap, _p(d), C.byref(g), C.byref(n)))
        return k[:n.value].copy(), d[:n.value].copy(), g.value

    def extract_fast_keypoint(self, img, existing=None, cap=None):
        """operator()(Frame*, ..., FAST_KEYPOINT) = ComputeKeyPointsFast: (keys = existing (re-oriented) + one per 5-px cell, desc)."""
        img = np.ascontiguousarray(img, np.uint8)
        h, w = img.shape
        existing = np.zeros(0, KP_DTYPE) if existing is None else np.ascontiguousarray(existing, KP_DTYPE)
        cap = cap or (len(existing) + (w // 5) * (h // 5) + 16)
        k = np.zeros(cap, KP_DTYPE)
        k[:len(existing)] = existing
        d = np.zeros((cap, 32), np.uint8)
        n = C.c_int()
        self._ck(self.L.ygzf_extract_fast_keypoint(self.h, _p(img), w, h, w, _p(k), len(existing), cap, _p(d), C.byref(n)))
        return k[:n.value].copy(), d[:n.value].copy()

    def extract_dso_multilevel(self, img, existing=None, grid_size=-1, cap=None):
        """The Frame overload over the multi-level ComputeKeyPointsDSO: (keys, desc, mnGridSize after the call)."""
        img = np.ascontiguousarray(img, np.uint8)
        h, w = img.shape
        existing = np.zeros(0, KP_DTYPE) if existing is None else np.ascontiguousarray(existing, KP_DTYPE)
        cap = cap or (len(existing) + 2 * (w // 5) * (h // 5) + 16)
        k = np.zeros(cap, KP_DTYPE)
        k[:len(existing)] = existing
        d = np.zeros((cap, 32), np.uint8)
        g, n = C.c_int(grid_size), C.c_int()
        self._ck(self.L.ygzf_extract_dso_multilevel(self.h, _p(img), w, h, w, _p(k), len(existing), cap, _p(d), C.byref(g), C.byref(n)))
        return k[:n.value].copy(), d[:n.value].copy(), g.value

    def vocabulary_set(self, parent, desc, depth_levels):
        """DBoW2 vocabulary tree onto the device: parent[i] of every node (node 0 = root), n_nodes x 32 centroids, m_L."""
        p = np.ascontiguousarray(parent, np.int32)
        d = np.ascontiguousarray(desc, np.uint8)
        self._ck(self.L.ygzf_vocabulary_set(self.h, len(p), depth_levels, _p(p), _p(d)))

    def bow_transform(self, desc, levelsup=4):
        """Tree descent of n descriptors -> (leaf node id, node id at level L - levelsup) per descriptor."""
        d = np.ascontiguousarray(desc, np.uint8)
        n = len(d)
        leaf, lvl = np.zeros(max(n, 1), np.int32), np.zeros(max(n, 1), np.int32)
        self._ck(self.L.ygzf_bow_transform(self.h, n, _p(d), levelsup, _p(leaf), _p(lvl)))
        return leaf[:n], lvl[:n]

    def describe_keys(self, keys, frame=0, recompute_angle=False):
        """Descriptors (+ optionally IC_Angle) of existing keys on the pyramid of `frame` of the last batch -> (angles, desc)."""
        k = np.ascontiguousarray(keys, KP_DTYPE)
        ang = np.zeros(len(k), np.float32)
        d = np.zeros((len(k), 32), np.uint8)
        self._ck(self.L.ygzf_describe_keys(self.h, frame, _p(k), len(k), int(recompute_angle), _p(ang), _p(d)))
        return ang, d

    def compute_stereo_matches(self, img_l, img_r, keys_l, desc_l, keys_r, desc_r, mb, mbf):
        """Frame::ComputeStereoMatches on host arrays -> (mvuRight, mvDepth)."""
        il, ir = np.ascontiguousarray(img_l, np.uint8), np.ascontiguousarray(img_r, np.uint8)
        h, w = il.shape
        kl, kr = np.ascontiguousarray(keys_l, KP_DTYPE), np.ascontiguousarray(keys_r, KP_DTYPE)
        dl, dr = np.ascontiguousarray(desc_l, np.uint8), np.ascontiguousarray(desc_r, np.uint8)
        ur, dp = np.zeros(max(len(kl), 1), np.float32), np.zeros(max(len(kl), 1), np.float32)
        self._ck(self.L.ygzf_compute_stereo_matches(self.h, _p(il), _p(ir), w, h, w, len(kl), _p(kl), _p(dl), len(kr), _p(kr), _p(dr), mb, mbf,
                                                    _p(ur), _p(dp)))
        return ur[:len(kl)], dp[:len(kl)]

    def stereo_batch(self, mb, mbf):
        self._ck(self.L.ygzf_stereo_batch(self.h, mb, mbf))

    def stereo_fetch(self, pair):
        w, h, _ = self._wh
        cap = max(self.max_keypoints(w, h), 1)
        ur, dp = np.zeros(cap, np.float32), np.zeros(cap, np.float32)
        self._ck(self.L.ygzf_stereo_fetch(self.h, pair, _p(ur), _p(dp), cap))
        return ur, dp

    def image_cache_reserve(self, n_slots, w, h):
        self._ck(self.L.ygzf_image_cache_reserve(self.h, n_slots, w, h))

    def image_cache_put(self, slot, img):
        img = np.ascontiguousarray(img, np.uint8)
        self._ck(self.L.ygzf_image_cache_put(self.h, slot, _p(img), img.shape[1], img.shape[0], img.shape[1]))
        self._inflight.append(img)
        if len(self._inflight) > 64:
            self.sync()

    def image_cache_put_resident(self, slot, src):
        """Fills the slot device to device from `src` (another Extractor that still holds the image and its pyramid)."""
        self._ck(self.L.ygzf_image_cache_put_resident(self.h, slot, src.h))

    def has_resident_image(self, w, h):
        return bool(self.L.ygzf_has_resident_image(self.h, w, h))

    def find_direct_projection_batch(self, cam, cur_slot, cur_Tcw7, ref_slot, ref_Tcw7, ref_kp, mp_world, px_curr, want_patches=False):
        """ORBmatcher::FindDirectProjection over a candidate batch -> (px_curr n x 2, search_level, success[, patches n x 100])."""
        ct = np.ascontiguousarray(cur_Tcw7, np.float32)
        rs = np.ascontiguousarray(ref_slot, np.int32)
        rt = np.ascontiguousarray(ref_Tcw7, np.float32)
        rk = np.ascontiguousarray(ref_kp, KP_DTYPE)
        mw = np.ascontiguousarray(mp_world, np.float32)
        px = np.array(px_curr, np.float32).reshape(-1, 2).copy()
        n = len(rs)
        sl = np.zeros(max(n, 1), np.int32)
        ok = np.zeros(max(n, 1), np.uint8)
        pt = np.zeros((max(n, 1), 100), np.uint8) if want_patches else None
        self._ck(self.L.ygzf_find_direct_projection_batch(self.h, C.byref(cam), cur_slot, _p(ct), n, _p(rs), _p(rt), _p(rk), _p(mw), _p(px), _p(sl),
                                                          _p(ok), _p(pt) if want_patches else None))
        self._inflight.clear()
        return (px, sl[:n], ok[:n], pt[:n]) if want_patches else (px, sl[:n], ok[:n])

    def timer_start(self):
        self._ck(self.L.ygzf_timer_start(self.h))

    def timer_stop(self):
        ms = C.c_float()
        self._ck(self.L.ygzf_timer_stop(self.h, C.byref(ms)))
        return ms.value

    def profile_enable(self, on=True):
        self._ck(self.L.ygzf_profile_enable(self.h, int(on)))

    def profile_reset(self):
        self._ck(self.L.ygzf_profile_reset(self.h))

    def profile_read(self):
        names = (C.c_char_p * 32)()
        ms = (C.c_float * 32)()
        n = (C.c_int * 32)()
        k = self._ck(self.L.ygzf_profile_read(self.h, names, ms, n, 32))
        return {names[i].decode(): (ms[i], n[i]) for i in range(k)}


def _frames_view(frames):
    """(n, h, w) uint8 frames as they lie when rows / frames are merely padded (unit stride along x), a contiguous copy otherwise;
    -> (array, row_pitch, frame_stride)"""
    f = frames
    if not (isinstance(f, np.ndarray) and f.dtype == np.uint8 and f.ndim == 3 and f.strides[2] == 1 and f.strides[1] >= f.shape[2] and
            (f.shape[0] == 1 or f.strides[0] >= f.strides[1] * f.shape[1])):
        f = np.ascontiguousarray(f, np.uint8)
    return f, f.strides[1], f.strides[0]


class MultiGpu:
    """ygzf_mgpu_*: frames (or units of consecutive frames) dealt round-robin over device slots, one host thread + context per slot, results
    in input order (include/ygzf.h).  `devices` may repeat a device index (several slots on one GPU)."""

    def __init__(self, devices, nfeatures=1000, scale_factor=1.2, nlevels=8, ini_th=20, min_th=7, max_width=752, max_height=480,
                 max_frames_per_device=64, cv_mode=CV_LEGACY_SSE2):
        self.L = load_library()
        self.cfg = ExtractorCfg(nfeatures, scale_factor, nlevels, ini_th, min_th, cv_mode)
        dv = (C.c_int * len(devices))(*devices)
        h = C.c_void_p()
        rc = self.L.ygzf_mgpu_create(dv, len(devices), C.byref(self.cfg), max_width, max_height, max_frames_per_device, C.byref(h))
        if rc != 0:
            raise YgzfError("ygzf_mgpu_create failed (%d): %s" % (rc, self.L.ygzf_last_error(None).decode()))
        self.h = h
        self.stride = self.L.ygzf_mgpu_keypoint_stride(self.h)

    def close(self):
        if getattr(self, "h", None):
            self.L.ygzf_mgpu_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def device_count(self):
        return self.L.ygzf_mgpu_device_count(self.h)

    def slot_of_frame(self, frame, unit=1):
        return self.L.ygzf_mgpu_slot_of_frame(self.h, frame, unit)

    def extract_match(self, frames, unit=1, cam=None, th=15.0, mono=True, check_level=True, check_ori=True, out=None):
        """-> (kps [n, stride], desc [n, stride, 32], n_kp [n], match [n, stride] or None, nmatches [n] or None); matching when cam is given"""
        frames, rp, fst = _frames_view(frames)
        n, h, w = frames.shape
        if out is None:
            out = (np.zeros((n, self.stride), KP_DTYPE), np.zeros((n, self.stride, 32), np.uint8), np.zeros(n, np.int32),
                   np.full((n, self.stride), -1, np.int32) if cam is not None else None, np.zeros(n, np.int32) if cam is not None else None)
        k, d, c, m, nm = out
        rc = self.L.ygzf_mgpu_extract_match(self.h, C.c_void_p(frames.ctypes.data), n, w, h, rp, fst, unit, C.byref(cam) if cam is not None else None, th, int(mono),
                                            int(check_level), int(check_ori), _p(k), _p(d), _p(c), self.stride, _p(m) if m is not None else None,
                                            _p(nm) if nm is not None else None)
        if rc != 0:
            raise YgzfError("ygzf_mgpu_extract_match failed (%d): %s" % (rc, self.L.ygzf_mgpu_last_error(self.h).decode()))
        return k, d, c, m, nm

    def chunk_frames(self):
        return self.L.ygzf_mgpu_chunk_frames(self.h)

    def extract_stereo(self, frames, mb, mbf, out=None):
        """frames = (left, right) pairs -> (kps [n, stride], desc [n, stride, 32], n_kp [n], u_right [n/2, stride], depth [n/2, stride])"""
        frames, rp, fst = _frames_view(frames)
        n, h, w = frames.shape
        if out is None:
            out = (np.zeros((n, self.stride), KP_DTYPE), np.zeros((n, self.stride, 32), np.uint8), np.zeros(n, np.int32),
                   np.full((n // 2, self.stride), -1, np.float32), np.full((n // 2, self.stride), -1, np.float32))
        k, d, c, ur, dp = out
        self.L.ygzf_mgpu_extract_stereo.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_size_t, C.c_float, C.c_float,
                                                    C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        rc = self.L.ygzf_mgpu_extract_stereo(self.h, C.c_void_p(frames.ctypes.data), n, w, h, rp, fst, mb, mbf, _p(k), _p(d), _p(c), self.stride, _p(ur), _p(dp))
        if rc != 0:
            raise YgzfError("ygzf_mgpu_extract_stereo failed (%d): %s" % (rc, self.L.ygzf_mgpu_last_error(self.h).decode()))
        return k, d, c, ur, dp
