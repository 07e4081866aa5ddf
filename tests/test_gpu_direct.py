"""GPU parity of ORBmatcher::FindDirectProjection + Align2D (src/ORBmatcher.cc:1525-1602, src/Align.cc:8-104) over candidate batches
against the oracle: warped reference patches, search level, success flag and the refined pixel are bit-identical."""
import numpy as np
import pytest

from orb_ygz_slam_amd.capi import EUROC
from orb_ygz_slam_amd.scene import rotvec_to_quat, two_view_scene

pytestmark = pytest.mark.gpu


def _scene(seed, rotvec, trans, w=752, h=480):
    A, B, (R, t), bp = two_view_scene(seed, w, h, EUROC, Z=4.0, rotvec=rotvec, trans=trans)
    q = rotvec_to_quat(rotvec)
    T7 = np.array([q[0], q[1], q[2], q[3], trans[0], trans[1], trans[2]], np.float32)
    return A, B, R, t, bp, T7


def test_find_direct_projection_batch(oracle):
    from orb_ygz_slam_amd import Extractor, make_camera
    w, h = 752, 480
    cam = make_camera(w, h)
    ex = Extractor(1000, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=1)
    oex = oracle.Extractor(1000, 1.2, 8, 20, 7)
    # two KeyFrames (A1, A2 = world frames of two scenes rendered from identity) and one current frame: candidates of both KFs in one batch
    A1, B1, R, t, bp, T7 = _scene(9, (0.01, -0.02, 0.03), (0.1, -0.05, 0.2))
    A2, _, _, _, _, _ = _scene(10, (0.0, 0.0, 0.0), (0.0, 0.0, 0.0))
    k1, _ = ex.extract(A1)
    k2, _ = ex.extract(A2)
    rng = np.random.default_rng(2)
    world1 = bp(k1["x"], k1["y"])
    world2 = bp(k2["x"], k2["y"]) * rng.uniform(0.5, 2.0, (len(k2), 1)).astype(np.float32)    # wrong depths / unrelated image: failures
    Xc = (R @ world1.T.astype(np.float64)).T + t
    u = EUROC["fx"] * Xc[:, 0] / Xc[:, 2] + EUROC["cx"]
    v = EUROC["fy"] * Xc[:, 1] / Xc[:, 2] + EUROC["cy"]
    px1 = np.stack([u, v], -1) + rng.uniform(-1.5, 1.5, (len(k1), 2))
    px2 = np.stack([k2["x"], k2["y"]], -1) + rng.uniform(-40, 40, (len(k2), 2))                 # some leave the image
    ref_kp = np.concatenate([k1, k2])
    world = np.concatenate([world1, world2])
    px0 = np.concatenate([px1, px2]).astype(np.float32)
    slot = np.concatenate([np.zeros(len(k1), np.int32), np.ones(len(k2), np.int32)])
    ident = np.array([0, 0, 0, 1, 0, 0, 0], np.float32)
    q2 = rotvec_to_quat((0.02, 0.01, -0.01))
    T2 = np.array([q2[0], q2[1], q2[2], q2[3], 0.05, 0.02, -0.03], np.float32)
    ref_T = np.concatenate([np.tile(ident, (len(k1), 1)), np.tile(T2, (len(k2), 1))])
    ex.image_cache_reserve(4, w, h)
    ex.image_cache_put(0, A1)
    ex.image_cache_put(1, A2)
    ex.image_cache_put(3, B1)
    gpx, gsl, gok, gpt = ex.find_direct_projection_batch(cam, 3, T7, slot, ref_T, ref_kp, world, px0, want_patches=True)
    opx, osl, ook, opt = oex.find_direct_projection_batch([A1, A2], B1, T7, EUROC, slot, ref_T, ref_kp, world, px0)
    assert (gpt == opt).all()
    assert (gsl == osl).all() and (gok == ook).all()
    assert np.array_equal(gpx.view(np.uint32), opx.view(np.uint32)) or np.allclose(gpx, opx, rtol=0, atol=0, equal_nan=True)
    # the first KeyFrame's candidates converge onto the true projections
    n1 = len(k1)
    inside = (u > 30) & (u < w - 30) & (v > 30) & (v < h - 30)
    good = ook[:n1].astype(bool) & inside
    assert good.sum() > 0.8 * inside.sum()
    err = np.hypot(opx[:n1, 0] - u, opx[:n1, 1] - v)
    assert np.median(err[good]) < 0.8
    assert len(np.unique(osl)) >= 4       # several search levels exercised
