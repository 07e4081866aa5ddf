"""DistributeOctTree (src/ORBextractor.cc:533-723) on the device has two plans (extract_kernels.hip, k_octree): candidates SORTED by path key
(levels whose candidates fit LDS: the 752x480 class) and the HISTOGRAM plan (1920x1080 / 3840x2160: counts per key prefix + prefix sum instead
of a sort; a tree that splits below the histogram's depth sends its workgroup back to the sorting path).  Same bytes from every plan, from the
fall-back between them, and from launches grouped by level: octree list order per level, keypoints, descriptors -- against the oracle."""
import os

import numpy as np
import pytest

from orb_ygz_slam_amd.synth import synth_frame
from tests.test_gpu_extract import _cmp_frame

pytestmark = pytest.mark.gpu


class _env:
    """YGZF_FORCE=oct_plan=..,oct_hist_bins=.. (csrc/ygzf_internal.h) while a context is created"""
    def __init__(self, **kv):
        self.kv = kv

    def __enter__(self):
        from orb_ygz_slam_amd.capi import force_env
        self.old = os.environ.get("YGZF_FORCE")
        os.environ["YGZF_FORCE"] = force_env(**self.kv)

    def __exit__(self, *a):
        if self.old is None:
            os.environ.pop("YGZF_FORCE", None)
        else:
            os.environ["YGZF_FORCE"] = self.old


def _clustered(seed, w, h):
    """corners crowded into two small spots of an otherwise empty image: the tree subdivides far below any histogram depth"""
    rng = np.random.default_rng(seed)
    img = np.full((h, w), 120, np.uint8)
    for (cx, cy) in ((w // 5, h // 4), (w - 90, h - 80)):
        img[cy - 40:cy + 40, cx - 40:cx + 40] = rng.integers(0, 256, (80, 80), dtype=np.uint8)
    return img


CASES = [
    # (w, h, nlevels, nfeatures, image)                          what it exercises
    (752, 480, 8, 1000, lambda: synth_frame(41, 752, 480)),       # the default geometry, forced through the histogram plan
    (640, 480, 8, 1000, lambda: np.random.default_rng(3).integers(0, 256, (480, 640), dtype=np.uint8)),   # pure noise: ~20 k candidates on level 0
    (333, 517, 8, 1500, lambda: synth_frame(42, 333, 517)),       # tall: one octree root
    (1280, 360, 6, 1200, lambda: synth_frame(43, 1280, 360)),     # wide: four roots
    (752, 480, 8, 1000, lambda: _clustered(7, 752, 480)),         # deep trees -> overflow of every histogram depth
    (320, 240, 4, 3000, lambda: synth_frame(44, 320, 240)),       # quota above the number of corners: the tree runs until nothing divides
]


@pytest.mark.parametrize("case", range(len(CASES)))
@pytest.mark.parametrize("plan,bins", [("hist", 8192), ("hist", 512), ("hist", 16), ("sort", 0)])
def test_every_plan_gives_the_oracles_tree(oracle, case, plan, bins):
    from orb_ygz_slam_amd import Extractor
    w, h, nl, nf, make = CASES[case]
    img = make()
    with _env(oct_plan=plan, oct_hist_bins=bins if bins else None):
        ex = Extractor(nf, 1.2, nl, 20, 7, max_width=w, max_height=h, max_batch=3)
        oex = oracle.Extractor(nf, 1.2, nl, 20, 7)
        imgs = np.stack([img, np.ascontiguousarray(img[::-1]), img])
        ex.extract_batch_host(imgs)
        _cmp_frame(oracle, ex, oex, imgs[0], frame=0)
        _cmp_frame(oracle, ex, oex, imgs[1], frame=1)
        k0, d0 = ex.batch_fetch(0)
        k2, d2 = ex.batch_fetch(2)
        assert np.array_equal(k0, k2) and np.array_equal(d0, d2)
        ex.close()


def test_automatic_plan_of_the_large_configs(oracle):
    """1920x1080 / 8 / 4000 picks the histogram plan by itself (level groups with their own LDS allotment); the forced sorting plan returns the
    same bytes -- and both equal the oracle."""
    from orb_ygz_slam_amd import Extractor
    w, h, nl, nf = 1920, 1080, 8, 4000
    img = synth_frame(77, w, h)
    oex = oracle.Extractor(nf, 1.2, nl, 20, 7)
    res = []
    for plan in (None, "sort"):
        with _env(oct_plan=plan, oct_hist_bins=None):
            ex = Extractor(nf, 1.2, nl, 20, 7, max_width=w, max_height=h, max_batch=1)
            ex.extract_batch_host(img[None])
            _cmp_frame(oracle, ex, oex, img, frame=0)
            res.append(ex.batch_fetch(0))
            ex.close()
    assert np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1])


@pytest.mark.parametrize("case", [0, 1, 4, 5])
@pytest.mark.parametrize("helpers,spin", [(2, None), (8, None), (8, 0)])
def test_helper_workgroups_give_the_same_tree(oracle, case, helpers, spin):
    """Launches of a few frames give every (level, frame) of the histogram plan helper workgroups for the keys of its candidates (k_octree, gridDim.z;
    chosen by the library for 1920x1080 and up): forced here on small images, one and three frames per launch, twice in a row (the hand-over
    counters only grow by what a launch brings), including the level that overflows the histogram and restarts on the sorting path -- the oracle's tree.
    spin = 0: workgroup 0 gives up waiting at once (what it does when its helpers cannot start) and computes the level alone beside its late helpers."""
    from orb_ygz_slam_amd import Extractor
    w, h, nl, nf, make = CASES[case]
    img = make()
    with _env(oct_plan="hist", oct_helpers=helpers, oct_helper_spin=spin):
        ex = Extractor(nf, 1.2, nl, 20, 7, max_width=w, max_height=h, max_batch=3)
        oex = oracle.Extractor(nf, 1.2, nl, 20, 7)
        imgs = np.stack([img, np.ascontiguousarray(img[::-1]), img])
        for rep in range(2):
            ex.extract_batch_host(imgs[:1])
            _cmp_frame(oracle, ex, oex, imgs[0], frame=0)
            ex.extract_batch_host(imgs)
            _cmp_frame(oracle, ex, oex, imgs[0], frame=0)
            _cmp_frame(oracle, ex, oex, imgs[1], frame=1)
            k0, d0 = ex.batch_fetch(0)
            k2, d2 = ex.batch_fetch(2)
            assert np.array_equal(k0, k2) and np.array_equal(d0, d2)
        ex.close()


def test_helpers_are_what_the_large_frame_takes(oracle):
    """one 1920x1080 frame: the library's own choice (eight helpers per level) and no helpers return the same bytes"""
    from orb_ygz_slam_amd import Extractor
    w, h, nl, nf = 1920, 1080, 8, 4000
    img = synth_frame(78, w, h)
    res = []
    for helpers in (None, 1):
        with _env(oct_helpers=helpers):
            ex = Extractor(nf, 1.2, nl, 20, 7, max_width=w, max_height=h, max_batch=1)
            ex.extract_batch_host(img[None])
            res.append(ex.batch_fetch(0))
            ex.close()
    assert np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1])
    oex = oracle.Extractor(nf, 1.2, nl, 20, 7)
    ok, od = oex.extract(img)
    for fld in ("x", "y", "size", "angle", "response", "octave", "class_id"):
        assert (res[0][0][fld] == ok[fld]).all(), fld
    assert np.array_equal(od, res[0][1])
