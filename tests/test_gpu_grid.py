"""Frame::AssignFeaturesToGrid + GetFeaturesInArea as a direct device query (ygzf_features_in_area) against the oracle, whose grid code is
pinned to the reference's own src/Frame.cc by tests/test_ref_frame.py: the index lists must be identical, order included."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _queries(rng, n, w, h):
    xyr = np.stack([rng.uniform(-60, w + 60, n), rng.uniform(-60, h + 60, n), rng.choice([0.5, 3.0, 15.0, 40.0, 120.0, 900.0], n)], 1).astype(np.float32)
    lv = np.stack([rng.integers(-1, 6, n), rng.integers(-1, 8, n)], 1).astype(np.int32)
    lv[::5] = -1                       # no level test at all
    lv[1::7, 1] = -1                   # only a minimum level
    return xyr, lv


@pytest.mark.parametrize("seed,w,h,nfeat", [(0, 752, 480, 1000), (1, 640, 480, 1000), (2, 1241, 376, 2000)])
def test_features_in_area_equals_oracle(oracle, seed, w, h, nfeat):
    from orb_ygz_slam_amd import Extractor, make_camera
    from orb_ygz_slam_amd.synth import synth_frame
    img = synth_frame(seed, w, h)
    ex = Extractor(nfeat, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=1)
    k, _ = ex.extract(img)
    sf = ex.tables()["scale"]
    cam = make_camera(w, h)
    rng = np.random.default_rng(100 + seed)
    xyr, lv = _queries(rng, 300, w, h)
    xyr[:len(k) // 8, :2] = np.stack([k["x"], k["y"]], 1)[:len(k) // 8]            # centred on keypoints: |dist| == 0 and exact-r edges
    got, cnt = ex.features_in_area(cam, k, xyr, lv)
    for q in range(len(xyr)):
        exp = oracle.features_in_area(k, sf, w, h, float(xyr[q, 0]), float(xyr[q, 1]), float(xyr[q, 2]), int(lv[q, 0]), int(lv[q, 1]))
        assert cnt[q] == len(exp), (q, xyr[q], lv[q])
        assert np.array_equal(got[q], exp), (q, xyr[q], lv[q])
    # without the level array: (-1, -1)
    got2, _ = ex.features_in_area(cam, k, xyr[:40])
    for q in range(40):
        assert np.array_equal(got2[q], oracle.features_in_area(k, sf, w, h, float(xyr[q, 0]), float(xyr[q, 1]), float(xyr[q, 2])))


def test_features_in_area_edges(oracle):
    from orb_ygz_slam_amd import Extractor, make_camera, KP_DTYPE
    w, h = 752, 480
    ex = Extractor(1000, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=1)
    cam = make_camera(w, h)
    sf = ex.tables()["scale"]
    rng = np.random.default_rng(9)
    # many keypoints in few cells (long cell lists, several 64-entry steps), on cell boundaries and on the image border
    k = np.zeros(5000, KP_DTYPE)
    k["x"] = np.concatenate([rng.uniform(100, 130, 3000), rng.choice([0.0, 11.75, 23.5, 751.0, 5.875], 2000)]).astype(np.float32)
    k["y"] = np.concatenate([rng.uniform(200, 215, 3000), rng.choice([0.0, 10.0, 20.0, 479.0, 5.0], 2000)]).astype(np.float32)
    k["octave"] = rng.integers(0, 8, 5000)
    xyr = np.array([[115, 207, 30], [115, 207, 8], [0, 0, 12], [751, 479, 12], [400, 240, 2000], [-500, 240, 100], [376, 900, 50], [115, 207, 0]], np.float32)
    lv = np.array([[-1, -1], [2, 5], [0, -1], [-1, 3], [-1, -1], [-1, -1], [-1, -1], [-1, -1]], np.int32)
    got, cnt = ex.features_in_area(cam, k, xyr, lv)
    for q in range(len(xyr)):
        exp = oracle.features_in_area(k, sf, w, h, float(xyr[q, 0]), float(xyr[q, 1]), float(xyr[q, 2]), int(lv[q, 0]), int(lv[q, 1]))
        assert cnt[q] == len(exp) and np.array_equal(got[q], exp), (q, cnt[q], len(exp))
    assert 4000 < cnt[4] < 5000 and cnt[5] == 0 and cnt[6] == 0 and cnt[7] == 0   # PosInGrid rounds: keys in the last half cell are in no cell
    # a cap below the result size: the first `cap` indices, the full count
    got_c, cnt_c = ex.features_in_area(cam, k, xyr, lv, cap=100)
    for q in range(len(xyr)):
        assert cnt_c[q] == cnt[q] and np.array_equal(got_c[q], got[q][:100])
    # no keypoints, no queries
    g0, c0 = ex.features_in_area(cam, k[:0], xyr, lv)
    assert all(len(a) == 0 for a in g0) and not c0.any()
    g1, c1 = ex.features_in_area(cam, k, xyr[:0])
    assert g1 == [] and len(c1) == 0
