"""The oracle against THE REFERENCE'S OWN src/Frame.cc with the real include/Frame.h (compiled where they lie into oracle/_ref/libref_frame.so
over oracle/ref_shim/; the reference's ORBextractor.cc and ORBmatcher.cc are linked in): the 64 x 48 feature grid (AssignFeaturesToGrid,
PosInGrid, GetFeaturesInArea: SURVEY 8a row a-15), Frame::isInFrustum (8f-3) and Frame::ComputeStereoMatches (8f-1).
CPU tier; skipped where the library was never built."""
import ctypes as C

import numpy as np
import pytest

from oracle import oracle_py as O
from orb_ygz_slam_amd.scene import stereo_scene, synth_frame

pytestmark = pytest.mark.skipif(O.ref_frame_lib() is None, reason="oracle/_ref/libref_frame.so not built (reference checkout absent)")

CAM = dict(fx=458.654, fy=457.296, cx=367.215, cy=248.375)
W, H = 752, 480


@pytest.fixture(scope="module")
def frame():
    oex = O.Extractor(1000, 1.2, 8, 20, 7)
    k, d = oex.extract(synth_frame(31, W, H))
    return oex, k, d, oex.tables()["scale"]


def test_features_in_area_equals_reference(frame):
    oex, k, d, sf = frame
    rng = np.random.default_rng(1)
    total = 0
    for _ in range(400):
        x, y = float(rng.uniform(-30, W + 30)), float(rng.uniform(-30, H + 30))
        r = float(rng.choice([3.0, 8.0, 15.0, 40.0, 120.0]))
        lo, hi = (-1, -1) if rng.uniform() < 0.3 else (int(rng.integers(-1, 7)), int(rng.integers(-1, 8)))
        e = O.features_in_area(k, sf, W, H, x, y, r, lo, hi)
        with O.reference_frame():
            g = O.features_in_area(k, sf, W, H, x, y, r, lo, hi)
        assert len(g) == len(e) and (g == e).all(), (x, y, r, lo, hi)
        total += len(e)
    assert total > 2000


def test_is_in_frustum_equals_reference(frame):
    oex, ka, da, sf = frame
    rng = np.random.default_rng(90)
    n = len(ka)
    depth = rng.uniform(2.0, 8.0, n).astype(np.float32)
    world = np.stack([(ka["x"] - np.float32(CAM["cx"])) / np.float32(CAM["fx"]) * depth, (ka["y"] - np.float32(CAM["cy"])) / np.float32(CAM["fy"]) * depth,
                      depth], -1).astype(np.float32)
    world[::17, 2] *= -1                                  # behind the camera
    world[5::23, 0] += 50                                 # outside the image
    normal = (world / np.linalg.norm(world, axis=1, keepdims=True)).astype(np.float32)
    normal[3::19] *= -1                                   # seen from behind
    dist = np.linalg.norm(world, axis=1).astype(np.float32)
    mf_max = (dist * sf[ka["octave"]]).astype(np.float32)
    mf_max[7::29] *= 0.3                                  # outside the scale-invariance range
    mf_min = (mf_max / sf[-1]).astype(np.float32)
    mx, mn = (np.float32(1.2) * mf_max).astype(np.float32), (np.float32(0.8) * mf_min).astype(np.float32)
    ang = np.float32(np.deg2rad(0.3))
    Rcw = np.array([[np.cos(ang), 0, np.sin(ang)], [0, 1, 0], [-np.sin(ang), 0, np.cos(ang)]], np.float32)
    tcw = np.array([0.02, -0.01, 0.03], np.float32)
    Ow = (-Rcw.T @ tcw).astype(np.float32)
    lsf = np.log(np.float32(1.2), dtype=np.float32)
    cam = dict(CAM, mb=0.11, mbf=47.9)
    for limit in (0.5, 0.9):
        e = O.is_in_frustum(ka, da, sf, W, H, cam, world, normal, mx, mn, mf_max, Rcw, tcw, Ow, lsf, limit)
        with O.reference_frame():
            g = O.is_in_frustum(ka, da, sf, W, H, cam, world, normal, mx, mn, mf_max, Rcw, tcw, Ow, lsf, limit)
        iv = e[0].astype(bool)
        assert (g[0] == e[0]).all() and 0.3 * len(iv) < iv.sum() < len(iv)
        for a, b in zip(g[1:], e[1:]):
            assert np.array_equal(a[iv].view(np.uint32), b[iv].view(np.uint32))


def test_compute_stereo_matches_equals_reference():
    L = O.ref_frame_lib()
    L.yr_stereo_config.argtypes = [C.c_int, C.c_float]
    L.yo_compute_stereo_matches.restype = None
    L.yo_compute_stereo_matches.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                                            C.c_void_p, C.c_float, C.c_float, C.c_void_p, C.c_void_p]
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    for seed, (w, h), nf, nl, sf, (mb, mbf) in ((3, (752, 480), 1200, 8, 1.2, (0.11, 47.9)), (5, (640, 480), 800, 6, 1.2, (0.12, 40.0)),
                                                (8, (752, 480), 2000, 8, 1.2, (0.11, 47.9))):
        left, right, _, _ = stereo_scene(seed, w, h)
        ex = O.Extractor(nf, sf, nl, 20, 7)
        kl, dl = ex.extract(left)
        kr, dr = ex.extract(right)
        e_ur, e_dp = ex.compute_stereo_matches(left, right, kl, dl, kr, dr, mb, mbf)
        L.yr_stereo_config(nl, sf)
        il, ir = np.ascontiguousarray(left), np.ascontiguousarray(right)
        ur, dp = np.zeros(len(kl), np.float32), np.zeros(len(kl), np.float32)
        L.yo_compute_stereo_matches(None, p(il), p(ir), w, h, len(kl), p(kl), p(dl), len(kr), p(kr), p(dr), mb, mbf, p(ur), p(dp))
        assert (e_ur >= 0).sum() > 200
        assert np.array_equal(ur.view(np.uint32), e_ur.view(np.uint32)) and np.array_equal(dp.view(np.uint32), e_dp.view(np.uint32)), seed
