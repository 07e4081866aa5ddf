"""The two remaining detectors of SURVEY 8a-3d on the device, bit-exact against the oracle (which tests/test_ref_extractor.py pins to the
reference's own code): ComputeKeyPointsFast -- the FAST_KEYPOINT branch of the Frame overload (src/ORBextractor.cc:1045-1051, 1189-1273) -- and
the multi-level ComputeKeyPointsDSO (:1388-1507).  Keypoints (position, size, angle, response, octave, order), descriptors, re-oriented
existing keys, mnGridSize."""
import numpy as np
import pytest

from orb_ygz_slam_amd.synth import synth_frame

pytestmark = pytest.mark.gpu

CASES = [(640, 480, 8, 1.2, 1000, 20, 7), (752, 480, 8, 1.2, 1200, 20, 7), (400, 300, 4, 1.5, 800, 12, 5), (515, 385, 6, 1.2, 600, 30, 10),
         (1241, 376, 8, 1.2, 2000, 20, 7), (322, 243, 3, 2.0, 500, 20, 7)]


def _same(k, d, ok, od, what):
    assert len(k) == len(ok), (what, len(k), len(ok))
    for f in ("x", "y", "size", "angle", "response", "octave", "class_id"):
        assert np.array_equal(k[f], ok[f]), (what, f, int((k[f] != ok[f]).sum()))
    assert np.array_equal(d, od), (what, int((d != od).any(axis=1).sum()))


@pytest.mark.parametrize("w,h,nl,sf,nf,ini,mn", CASES)
def test_fast_keypoint_branch(oracle, w, h, nl, sf, nf, ini, mn):
    from orb_ygz_slam_amd import Extractor
    ex = Extractor(nf, sf, nl, ini, mn, max_width=w, max_height=h, max_batch=1)
    oex = oracle.Extractor(nf, sf, nl, ini, mn)
    for seed in (31, 32):
        img = synth_frame(seed, w, h)
        k0, _ = oex.extract(img)
        for existing in (None, k0[::9].copy()):
            k, d = ex.extract_fast_keypoint(img, existing)
            ok, od = oex.extract_fast(img, existing)
            assert len(ok) > 100
            _same(k, d, ok, od, (w, h, seed, existing is not None))


def test_fast_keypoint_degenerate_images(oracle):
    from orb_ygz_slam_amd import Extractor
    w, h = 320, 240
    ex = Extractor(500, 1.2, 4, 20, 7, max_width=w, max_height=h, max_batch=1)
    oex = oracle.Extractor(500, 1.2, 4, 20, 7)
    rng = np.random.default_rng(3)
    for name, img in (("flat", np.full((h, w), 77, np.uint8)), ("noise", rng.integers(0, 256, (h, w), dtype=np.uint8)),
                      ("binary", (rng.integers(0, 2, (h, w)) * 255).astype(np.uint8))):
        k, d = ex.extract_fast_keypoint(img)
        ok, od = oex.extract_fast(img)
        _same(k, d, ok, od, name)


@pytest.mark.parametrize("w,h,nl,sf,nf,ini,mn", CASES)
def test_dso_multilevel(oracle, w, h, nl, sf, nf, ini, mn):
    from orb_ygz_slam_amd import Extractor
    ex = Extractor(nf, sf, nl, ini, mn, max_width=w, max_height=h, max_batch=1)
    oex = oracle.Extractor(nf, sf, nl, ini, mn)
    img = synth_frame(41, w, h)
    k0, _ = oex.extract(img)
    for existing in (None, k0[::6].copy()):
        k, d, g = ex.extract_dso_multilevel(img, existing)
        ok, od, og = oracle.Extractor(nf, sf, nl, ini, mn).extract_dso_multilevel(img, existing)
        assert g == og and len(ok) > 100
        _same(k, d, ok, od, (w, h, existing is not None))


def test_dso_multilevel_retry_passes(oracle):
    """a feature budget the first grids cannot satisfy: the retry passes (grid - 5 ... 7) with their persisting occupancy"""
    from orb_ygz_slam_amd import Extractor
    w, h = 480, 360
    img = np.full((h, w), 100, np.uint8)
    rng = np.random.default_rng(8)
    for _ in range(60):                                    # sparse blobs: most cells of the first passes stay empty
        x, y = int(rng.integers(25, w - 30)), int(rng.integers(25, h - 30))
        img[y:y + 4, x:x + 4] = int(rng.integers(150, 255))
    ex = Extractor(600, 1.2, 4, 20, 7, max_width=w, max_height=h, max_batch=1)
    oex = oracle.Extractor(600, 1.2, 4, 20, 7)
    k, d, g = ex.extract_dso_multilevel(img)
    ok, od, og = oex.extract_dso_multilevel(img)
    lw, lh = oex.level_size(w, h, 3)
    assert g == og and g < int(np.sqrt(lw * lh / oex.tables()["nfeat"][3]))     # the last level went through retry passes
    _same(k, d, ok, od, "retry")
