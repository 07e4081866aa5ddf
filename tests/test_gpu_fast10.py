"""GPU parity of the Thirdparty/fast replacement (ygzf_fast10) -- the one part of the hot path that is PINNED to reference
code: golden vectors generated from the reference's own libfast (tests/golden/fast10_test1.npz, incl. the 167-corner
known answer of Thirdparty/fast/test/test.cpp:52), the live reference library when oracle/_ref is present, and the oracle."""
import os

import numpy as np
import pytest

from orb_ygz_slam_amd.synth import synth_frame

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden", "fast10_test1.npz")


def test_fast10_golden_167(oracle):
    from orb_ygz_slam_amd import Extractor
    ex = Extractor(max_width=64, max_height=64)
    g = np.load(GOLD)
    img = g["image"]
    for thr in (75, 20):
        xy, sc, nm = ex.fast10(img, thr)
        assert (xy == g["xy_%d" % thr]).all() and (sc == g["score_%d" % thr]).all() and (nm == g["nonmax_%d" % thr]).all()
    assert len(ex.fast10(img, 75)[0]) == 167


def test_fast10_vs_reference_and_oracle(oracle):
    from orb_ygz_slam_amd import Extractor
    ex = Extractor(max_width=64, max_height=64)
    rng = np.random.default_rng(1)
    imgs = [synth_frame(31, 160, 120), synth_frame(32, 97, 61), rng.integers(0, 256, (40, 64), dtype=np.uint8),
            rng.integers(0, 256, (30, 22), dtype=np.uint8), np.full((32, 32), 9, np.uint8)]
    for img in imgs:
        for thr in (5, 20, 40):
            got = ex.fast10(img, thr)
            exp = oracle.fast10(img, thr)
            for a, b in zip(got, exp):
                assert a.shape == b.shape and (a == b).all()
            if oracle.ref_fast() is not None:
                ref = oracle.ref_fast10(img, thr, which=1)
                for a, b in zip(got, ref):
                    assert a.shape == b.shape and (a == b).all()
    # grid-cell use (ComputeKeyPointsDSOSingleLevel, src/ORBextractor.cc:1326-1337): windows narrower than 22 px inside a frame
    big = synth_frame(33, 128, 96)
    import ctypes as C
    for (x0, y0, w, h) in ((10, 10, 19, 19), (40, 30, 21, 12), (5, 7, 30, 30)):
        xy, sc, nm = ex.fast10(big, 20, window=(x0, y0, w, h))
        sub = big[y0:, x0:]
        xo = np.zeros((w * h, 2), np.int16)
        n = oracle.lib().yo_fast10_detect(C.c_void_p(sub.ctypes.data), w, h, big.shape[1], 20, C.c_void_p(xo.ctypes.data), w * h)
        assert len(xy) == n and (xy == xo[:n]).all()
    # windows that would make the reference read outside the image are rejected, not emulated
    from orb_ygz_slam_amd import YgzfError
    with pytest.raises(YgzfError):
        ex.fast10(big, 20, window=(0, 0, 19, 19))
