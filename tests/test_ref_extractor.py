"""The oracle's extractor against THE REFERENCE'S OWN src/ORBextractor.cc (compiled where it lies into oracle/_ref/libref_orbextractor.so
by oracle/Makefile, against the OpenCV stand-in of oracle/ref_shim/ whose primitives are the oracle's restatements, under a monotone
allocator -- see oracle/ref_shim/mini_cv.h and oracle/ref_orbextractor_capi.cpp).

This pins every line of reference-owned logic on the extractor rows of SURVEY 8a (pyramid flow, 30-px cell loop and threshold fallback,
DivideNode / DistributeOctTree incl. list order and the expand-biggest-first phase, IC_Angle, computeOrbDescriptor with the reference's
own pattern table, scale bookkeeping, the Frame overload with existing keys, ComputeKeyPointsDSOSingleLevel + ShiTomasiScore on the
reference's own libfast incl. the persistent mnGridSize) bit for bit.  cv::resize / cv::FAST / cv::GaussianBlur / cv::fastAtan2 /
cvRound are the same restatements on both sides and stay unpinned.

CPU tier; skipped where the library was never built (no reference checkout and no prebuilt oracle/_ref)."""
import numpy as np
import pytest

from oracle import oracle_py as O
from orb_ygz_slam_amd.scene import synth_frame

pytestmark = pytest.mark.skipif(O.ref_extractor_lib() is None, reason="oracle/_ref/libref_orbextractor.so not built (reference checkout absent)")

FIELDS = ("x", "y", "size", "angle", "response", "octave", "class_id")


def assert_same(rk, rd, ok, od, what):
    assert len(rk) == len(ok), (what, len(rk), len(ok))
    for f in FIELDS:
        assert (rk[f] == ok[f]).all(), (what, f, int((rk[f] != ok[f]).sum()))
    assert rd.shape == od.shape and (rd == od).all(), (what, "descriptors")


CONFIGS = [  # (w, h, nfeatures, scale_factor, nlevels, ini, min)
    (752, 480, 1000, 1.2, 8, 20, 7), (640, 480, 1000, 1.2, 8, 20, 7), (320, 240, 500, 1.2, 4, 20, 7), (401, 303, 700, 1.5, 5, 20, 7),
    (752, 480, 2000, 1.2, 8, 20, 7), (517, 389, 1500, 1.1, 10, 12, 5), (640, 360, 300, 2.0, 4, 30, 10), (203, 177, 250, 1.25, 3, 20, 7),
    (1280, 720, 3000, 1.2, 8, 20, 7), (752, 480, 1000, 1.2, 1, 20, 7),
]


@pytest.mark.parametrize("cfg", CONFIGS)
def test_operator_image_equals_reference(cfg):
    w, h, nf, sf, nl, ini, mn = cfg
    for seed in (3, 11):
        img = synth_frame(seed, w, h)
        rk, rd = O.ref_extract(img, nf, sf, nl, ini, mn)
        ok, od = O.Extractor(nf, sf, nl, ini, mn).extract(img)
        assert len(rk) > 50
        assert_same(rk, rd, ok, od, (cfg, seed))


def test_special_images_equal_reference():
    w, h = 480, 360
    rng = np.random.default_rng(5)
    yy, xx = np.mgrid[0:h, 0:w]
    imgs = {"flat": np.full((h, w), 90, np.uint8), "noise": rng.integers(0, 256, (h, w)).astype(np.uint8),
            "binary": (rng.integers(0, 2, (h, w)) * 255).astype(np.uint8), "sparse": ((rng.uniform(size=(h, w)) > 0.995) * 255).astype(np.uint8),
            "blocks": ((xx // 7 + yy // 5) % 2 * 200 + 20).astype(np.uint8),
            "bowls": np.clip((((xx % 16) - 8) ** 2 + ((yy % 16) - 8) ** 2) * (250.0 / 128.0), 0, 255).astype(np.uint8),
            "low_contrast": (synth_frame(2, w, h) // 16 + 100).astype(np.uint8)}   # cells that only fire at minThFAST
    for name, img in imgs.items():
        rk, rd = O.ref_extract(img, 800, 1.2, 6, 20, 7)
        ok, od = O.Extractor(800, 1.2, 6, 20, 7).extract(img)
        assert_same(rk, rd, ok, od, name)
    assert len(O.ref_extract(imgs["flat"], 800, 1.2, 6, 20, 7)[0]) == 0


def test_pyramid_equals_reference():
    img = synth_frame(4, 752, 480)
    for sf, nl in ((1.2, 8), (1.5, 5), (2.0, 4)):
        ref = O.ref_pyramid(img, sf, nl)
        ora = O.Extractor(1000, sf, nl, 20, 7).pyramid(img)
        for lvl in range(nl):
            assert ref[lvl].shape == ora[lvl].shape and (ref[lvl] == ora[lvl]).all(), (sf, nl, lvl)


def test_frame_overload_orbslam_with_existing_keys():
    """operator()(Frame*, ..., ORBSLAM_KEYPOINT): descriptors of the keys the frame already holds (at pt * invScale[octave] of their
    level, angle untouched) come first, the octree keypoints follow."""
    w, h, nf, sf, nl = 640, 480, 600, 1.2, 6
    a, b = synth_frame(6, w, h), synth_frame(6, w + 8, h + 8)[3:3 + h, 5:5 + w]
    oex = O.Extractor(nf, sf, nl, 20, 7)
    prev, _ = oex.extract(a)
    existing = prev[::7].copy()                        # keys tracked from the previous frame, all levels
    ref = O.RefFrameExtractor(nf, sf, nl, 20, 7)
    rk, rd = ref.extract(b, 0, existing)
    ref.close()
    nk, nd = oex.extract(b)
    ed = oex.describe_keys(b, existing, recompute_angle=False)
    ed = ed[1] if isinstance(ed, tuple) else ed
    assert len(rk) == len(existing) + len(nk)
    n = len(existing)
    for f in FIELDS:
        assert (rk[f][:n] == existing[f]).all() and (rk[f][n:] == nk[f]).all(), f
    assert (rd[:n] == ed).all() and (rd[n:] == nd).all()


def test_frame_overload_dso_sequence_with_state():
    """DSO_KEYPOINT over a short clip on one persistent extractor: FAST-10 per grid cell on the reference's libfast, occupancy of the
    existing keys, Shi-Tomasi selection, the mnGridSize retry loop whose result carries over to the next frame."""
    w, h, nf = 640, 480, 800
    base = synth_frame(8, w + 40, h + 40)
    ref = O.RefFrameExtractor(nf, 1.2, 8, 20, 7)
    oex = O.Extractor(nf, 1.2, 8, 20, 7)
    grid = -1
    existing = None
    for i in range(4):
        img = np.ascontiguousarray(base[5 * i:5 * i + h, 7 * i:7 * i + w])
        rk, rd = ref.extract(img, 2, existing)
        ok, od, grid = oex.extract_dso(img, existing, grid)
        assert len(rk) > 100
        assert_same(rk, rd, ok, od, ("dso frame", i))
        existing = ok[::5].copy()                      # a subset survives as the next frame's tracked keys
    ref.close()


@pytest.mark.parametrize("seed", range(12))
def test_fuzz_sizes_and_configs_equal_reference(seed):
    rng = np.random.default_rng(900 + seed)
    while True:   # the reference has undefined behaviour on degenerate levels (no 30-px cell fits: division by zero, :739-742; taller than
        # 1.5 x wide: nIni = 0, :541-544) where the oracle defines a result -- keep every level a landscape of at least 90 x 70 pixels
        w, h = int(rng.integers(120, 900)), int(rng.integers(100, 640))
        nl = int(rng.integers(1, 10))
        sf = float(rng.choice([1.1, 1.2, 1.2, 1.25, 1.5, 2.0]))
        top = sf ** (nl - 1)
        if w >= h and w / top >= 90 and h / top >= 70:
            break
    nf = int(rng.integers(50, 3000))
    ini = int(rng.choice([20, 20, 12, 30]))
    mn = int(rng.choice([7, 7, 5, 10]))
    img = synth_frame(1000 + seed, w, h)
    if seed % 4 == 3:
        img = np.clip(img.astype(np.int32) * 3 - 200, 0, 255).astype(np.uint8)     # hard contrast: dense corners, many octree ties
    rk, rd = O.ref_extract(img, nf, sf, nl, ini, mn)
    ok, od = O.Extractor(nf, sf, nl, ini, mn).extract(img)
    assert_same(rk, rd, ok, od, (w, h, nf, sf, nl, ini, mn))
