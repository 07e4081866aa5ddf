"""Pins the FAST-9 predicate and score of SURVEY a-3b (what cv::FAST(..., TYPE_9_16) computes, restated in oracle/oracle_cvprims.cpp) to code the
REFERENCE itself holds: Rosten's machine-generated FAST-9 decision tree, /root/reference/Thirdparty/fast/include/fast/corner_9.h:1
(`is_corner_9<fast::Less>` / `is_corner_9<fast::Greater>`, comparison policies faster_corner_utilities.h:19-41), compiled where it lies into
oracle/_ref/libfast_ref.so (oracle/ref_fast9_capi.cpp).

 * corner SET: oracle fast9(nonmax=False) == {pixels the reference's tree accepts} on the one real image the reference ships (test1.png, via the
   committed fixture) and on synthetic / noise / degenerate images, at thresholds 5, 7, 12, 20, 40 (7 and 20 are the extractor's minTh / iniTh);
 * SCORE: the oracle's cornerScore<16> restatement == the largest barrier at which the reference's tree still says "corner" -- OpenCV's
   documented meaning of the FAST response, evaluated with the reference's predicate by bisection.

What stays unpinned after this (no reference-held code exists for it): cv::FAST's 3x3 non-maximum suppression rule (strict >), cv::resize,
cv::GaussianBlur, cv::fastAtan2."""
import os

import numpy as np
import pytest

from orb_ygz_slam_amd.synth import synth_frame

GOLD = os.path.join(os.path.dirname(__file__), "golden", "fast10_test1.npz")
THRESHOLDS = (5, 7, 12, 20, 40)


def _images():
    rng = np.random.default_rng(9)
    imgs = {"test1.png": np.load(GOLD)["image"],
            "synth_a": synth_frame(31, 320, 240), "synth_b": synth_frame(32, 197, 131),
            "noise": rng.integers(0, 256, (96, 128), dtype=np.uint8),
            "lowcontrast": (rng.integers(0, 24, (80, 90)) + 100).astype(np.uint8),
            "saturated": np.where(rng.random((70, 75)) < 0.5, 0, 255).astype(np.uint8),   # p +- t leaves [0, 255]: the short arithmetic of prep_t
            "tiny7": rng.integers(0, 256, (7, 7), dtype=np.uint8),
            "constant": np.full((40, 50), 77, np.uint8)}
    # a 2-level checkerboard of 5-px squares with a gradient: many exact ties |diff| == t
    yy, xx = np.mgrid[0:90, 0:110]
    imgs["ties"] = (((xx // 5 + yy // 5) % 2) * 20 + 100 + (xx % 3)).astype(np.uint8)
    return imgs


@pytest.fixture(scope="module")
def ref9(oracle):
    if oracle.ref_fast() is None or oracle.ref_fast9_corners(np.zeros((8, 8), np.uint8), 5) is None:
        pytest.skip("oracle/_ref/libfast_ref.so (with the FAST-9 shim) not built: no reference checkout")
    return oracle


@pytest.mark.parametrize("thr", THRESHOLDS)
def test_corner_set_equals_reference_tree(ref9, thr):
    total = 0
    for name, img in _images().items():
        xs, ys, sc = ref9.fast9(img, thr, nonmax=False)
        rx, ry = ref9.ref_fast9_corners(img, thr)
        assert len(xs) == len(rx), "%s @%d: %d corners, the reference's tree finds %d" % (name, thr, len(xs), len(rx))
        assert (xs == rx).all() and (ys == ry).all(), "%s @%d" % (name, thr)   # both raster order
        total += len(xs)
    assert total > 1000


def test_score_is_the_largest_barrier_of_the_reference_tree(ref9):
    checked = 0
    for name, img in _images().items():
        for thr in (0, 7, 20):
            xs, ys, sc = ref9.fast9(img, thr, nonmax=False)
            if not len(xs):
                continue
            mb = ref9.ref_fast9_max_barrier(img, xs, ys)
            assert (mb >= thr).all()
            assert (sc == mb).all(), "%s @%d: %d scores differ from the reference tree's largest passing barrier" % (name, thr, int((sc != mb).sum()))
            checked += len(xs)
    assert checked > 5000


def test_nonmax_output_is_a_subset_with_the_same_scores(ref9):
    """What the reference's tree cannot say anything about is the NMS rule itself; what it can: every keypoint cv::FAST(nonmax=true) keeps is a corner of
    the tree, carries the tree's score, and no 8-neighbour that is also a corner has a score >= its own (strict maximum)."""
    img = np.load(GOLD)["image"]
    for thr in (7, 20):
        xs, ys, sc = ref9.fast9(img, thr, nonmax=True)
        ax, ay, asc = ref9.fast9(img, thr, nonmax=False)
        smap = np.zeros(img.shape, np.int32)
        smap[ay, ax] = ref9.ref_fast9_max_barrier(img, ax, ay)
        assert (smap[ys, xs] == sc).all()
        for dy in (-1, 0, 1):
            for dx in (-1, 0, 1):
                if dx or dy:
                    assert (smap[ys + dy, xs + dx] < sc).all()
        # and it is maximal: every corner that is a strict 3x3 maximum is kept
        pad = np.pad(smap, 1)
        nb = np.max([pad[1 + dy:pad.shape[0] - 1 + dy, 1 + dx:pad.shape[1] - 1 + dx] for dy in (-1, 0, 1) for dx in (-1, 0, 1) if dx or dy], axis=0)
        keep = np.zeros(img.shape, bool)
        keep[ay, ax] = True
        keep &= smap > nb
        assert keep.sum() == len(xs)
