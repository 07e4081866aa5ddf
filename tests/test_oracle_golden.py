"""The oracle against its committed golden vectors (tests/golden/extract_golden.npz, made by
tools/make_golden_extract.py).  PARITY UNPINNED w.r.t. the reference (no reference vectors exist for this path); the
FAST-10 vectors in test_oracle_fast10.py are the ones pinned to reference code."""
import hashlib
import os

import numpy as np
import pytest

from orb_ygz_slam_amd.synth import synth_frame

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "extract_golden.npz"))
CASES = [("vga_s0", 640, 480, 0, (1000, 1.2, 8, 20, 7)), ("vga_s1", 640, 480, 1, (1000, 1.2, 8, 20, 7)),
         ("euroc_s2", 752, 480, 2, (1000, 1.2, 8, 20, 7)), ("euroc_4lvl_s3", 752, 480, 3, (1000, 2.0, 4, 20, 7)),
         ("small_s4", 320, 240, 4, (500, 1.2, 8, 20, 7))]


@pytest.mark.parametrize("mode", [0, 1, 2])
@pytest.mark.parametrize("name,w,h,seed,cfg", CASES)
def test_oracle_matches_golden(oracle, name, w, h, seed, cfg, mode):
    img = synth_frame(seed, w, h)
    assert hashlib.sha256(img.tobytes()).digest() == GOLD[name + "_img_sha"].tobytes(), "synthetic generator drifted"
    with oracle.cv_mode(mode):
        k, d = oracle.Extractor(*cfg).extract(img)
    assert len(k) == int(GOLD[name + "_n"][0])
    assert hashlib.sha256(k.tobytes() + d.tobytes()).digest() == GOLD["%s_m%d_sha" % (name, mode)].tobytes()
    if mode == 0 and name + "_kps" in GOLD:
        assert (k == GOLD[name + "_kps"]).all() and (d == GOLD[name + "_desc"]).all()


def test_extract_invariants(oracle):
    img = synth_frame(0, 640, 480)
    ex = oracle.Extractor(1000, 1.2, 8, 20, 7)
    k, d = ex.extract(img)
    t = ex.tables()
    assert d.shape == (len(k), 32)
    assert (np.diff(k["octave"]) >= 0).all()                       # level-major order
    for l in range(8):
        kl = k[k["octave"] == l]
        n = len(kl)
        assert n <= t["nfeat"][l] + 3                               # octree may overshoot N by at most 2-3 nodes
        lw, lh = ex.level_size(640, 480, l)
        xs, ys = kl["x"] / t["scale"][l], kl["y"] / t["scale"][l]
        assert (xs >= 19 - 1e-3).all() and (xs <= lw - 19 + 1e-3).all() and (ys >= 19 - 1e-3).all() and (ys <= lh - 19 + 1e-3).all()
        assert (kl["size"] == np.float32(int(np.float32(31) * t["scale"][l]))).all()
    assert ((k["angle"] >= 0) & (k["angle"] < 360.0001)).all()
    assert (k["response"] >= 7).all() and (k["class_id"] == -1).all()
    # constant image: nothing; reference releases the descriptor matrix
    k0, d0 = ex.extract(np.full((480, 640), 77, np.uint8))
    assert len(k0) == 0 and d0.shape == (0, 32)
