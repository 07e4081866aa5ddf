"""GPU parity of the DSO_KEYPOINT path (ORBextractor::operator()(Frame*, ..., DSO_KEYPOINT), src/ORBextractor.cc:1031-1127,
:1275-1386, :1152-1187) through ygzf_extract_dso against the oracle: keypoints, angles and descriptors bit-exact, the persistent
grid size identical."""
import numpy as np
import pytest

from orb_ygz_slam_amd.synth import synth_frame

pytestmark = pytest.mark.gpu


def _cmp(kg, dg, gg, ko, do, go):
    assert gg == go
    assert len(kg) == len(ko)
    for f in ("x", "y", "size", "response", "octave", "class_id"):
        assert np.array_equal(kg[f], ko[f]), f
    assert np.array_equal(kg["angle"].view(np.uint32), ko["angle"].view(np.uint32))
    assert np.array_equal(dg, do)


@pytest.mark.parametrize("w,h,nfeat", [(752, 480, 1000), (640, 480, 400), (752, 480, 3000), (376, 240, 150)])
def test_dso_fresh_frame(oracle, w, h, nfeat):
    from orb_ygz_slam_amd import Extractor
    ex = Extractor(nfeat, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=1)
    oex = oracle.Extractor(nfeat, 1.2, 8, 20, 7)
    g_gpu, g_cpu = -1, -1
    for seed in (0, 1, 2):       # the grid size carries over from frame to frame
        img = synth_frame(seed, w, h)
        kg, dg, g_gpu = ex.extract_dso(img, grid_size=g_gpu)
        ko, do, g_cpu = oex.extract_dso(img, grid_size=g_cpu)
        assert len(ko) > 0
        _cmp(kg, dg, g_gpu, ko, do, g_cpu)


def test_dso_with_existing_keys(oracle):
    """Tracked frame: existing multi-level keys keep their slots, get fresh angles + descriptors, and block their pixels."""
    from orb_ygz_slam_amd import Extractor
    w, h = 752, 480
    ex = Extractor(1000, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=1)
    oex = oracle.Extractor(1000, 1.2, 8, 20, 7)
    img = synth_frame(5, w, h)
    k0, _ = ex.extract(img)                    # ORB-SLAM keys on all levels ...
    kd, _, g = ex.extract_dso(img)             # ... and DSO keys of the same frame (these sit exactly on FAST corners)
    existing = np.concatenate([k0[::3], kd[::4]])
    assert len(existing) > 300
    kg, dg, gg = ex.extract_dso(img, existing=existing, grid_size=g)
    ko, do, go = oex.extract_dso(img, existing=existing, grid_size=g)
    assert len(ko) > len(existing)
    _cmp(kg, dg, gg, ko, do, go)
    # none of the new keys sits on an occupied pixel
    occ = set(zip(np.rint(existing["x"]).astype(int), np.rint(existing["y"]).astype(int)))
    new = kg[len(existing):]
    assert not (set(zip(new["x"].astype(int), new["y"].astype(int))) & occ)


def test_dso_grid_shrinks_and_floor(oracle):
    """Few corners: the grid shrinks by 5 per pass down to the floor of 7 and the previous pass's keys stand."""
    from orb_ygz_slam_amd import Extractor
    w, h = 640, 480
    ex = Extractor(4000, 1.2, 4, 20, 7, max_width=w, max_height=h, max_batch=1)
    oex = oracle.Extractor(4000, 1.2, 4, 20, 7)
    rng = np.random.default_rng(3)
    img = np.full((h, w), 90, np.uint8)
    for _ in range(60):                                       # a sparse scene: 60 bright squares
        x, y = int(rng.integers(30, w - 40)), int(rng.integers(30, h - 40))
        img[y:y + 9, x:x + 9] = 200
    kg, dg, gg = ex.extract_dso(img, grid_size=40)
    ko, do, go = oex.extract_dso(img, grid_size=40)
    assert go == 7
    _cmp(kg, dg, gg, ko, do, go)


def test_dso_blank_frame(oracle):
    from orb_ygz_slam_amd import Extractor
    ex = Extractor(500, 1.2, 8, 20, 7, max_width=320, max_height=240, max_batch=1)
    oex = oracle.Extractor(500, 1.2, 8, 20, 7)
    img = np.full((240, 320), 128, np.uint8)
    kg, dg, gg = ex.extract_dso(img)
    ko, do, go = oex.extract_dso(img)
    assert len(kg) == 0 and len(ko) == 0 and gg == go


def test_describe_existing_keys(oracle):
    """ygzf_describe_keys: the 'existing ones' loop of the Frame overload (stored angle / recomputed angle), any frame of a batch."""
    from orb_ygz_slam_amd import Extractor
    w, h = 752, 480
    ex = Extractor(1000, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=2)
    oex = oracle.Extractor(1000, 1.2, 8, 20, 7)
    imgs = np.stack([synth_frame(7, w, h), synth_frame(8, w, h)])
    ex.extract_batch_host(imgs)
    k1, d1 = ex.batch_fetch(1)
    # keys of frame 1 re-described on frame 1 with their own angle reproduce their descriptors
    ang, d = ex.describe_keys(k1, frame=1)
    assert np.array_equal(d, d1)
    # sub-pixel (direct-tracked) positions with arbitrary angles, on frame 0, stored angle and recomputed angle
    rng = np.random.default_rng(1)
    keys = k1[::2].copy()
    inv = oex.tables()["inv_scale"]
    sizes = np.array([ex.level_size(w, h, l) for l in range(8)])
    lx, ly = keys["x"] * inv[keys["octave"]], keys["y"] * inv[keys["octave"]]
    keep = (lx >= 19) & (ly >= 19) & (lx < sizes[keys["octave"], 0] - 19) & (ly < sizes[keys["octave"], 1] - 19)
    keys = keys[keep]
    assert len(keys) > 200
    keys["x"] += rng.uniform(-2, 2, len(keys)).astype(np.float32)
    keys["y"] += rng.uniform(-2, 2, len(keys)).astype(np.float32)
    keys["angle"] = rng.uniform(0, 360, len(keys)).astype(np.float32)
    for rec in (False, True):
        ang, d = ex.describe_keys(keys, frame=0, recompute_angle=rec)
        ko, do = oex.describe_keys(imgs[0], keys, recompute_angle=rec)
        assert np.array_equal(d, do)
        if rec:
            assert np.array_equal(ang.view(np.uint32), ko["angle"].view(np.uint32))
