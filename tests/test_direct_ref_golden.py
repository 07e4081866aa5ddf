"""tests/golden/direct_ref.npz holds what THE REFERENCE'S OWN ORBmatcher::FindDirectProjection (GetWarpAffineMatrix, WarpAffine, GetBestSearchLevel, and
ygz::Align2D of src/Align.cc) returned for every keypoint of the two scenes of tests/direct_ref_cases.py (tools/make_golden_direct_ref.py, run where the
checkout is): refined pixel (bit pattern), search level, success flag, warped 10 x 10 patch -- 2019 candidates, some of which leave the image.  The oracle
(CPU tier) and the device (GPU tier) must reproduce those bytes wherever they run."""
import os

import numpy as np
import pytest

from tests import direct_ref_cases as D

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "direct_ref.npz")


def _check(make_extractor, run):
    g = np.load(GOLD)
    aligned = 0
    for j in range(len(D.SCENES)):
        ex = make_extractor()
        A, B, cur7, slot, ref7, ka, world, px0 = D.scene(j, ex)
        px, sl, ok, pt = run(ex, A, B, cur7, slot, ref7, ka, world, px0)
        assert np.array_equal(np.asarray(sl, np.int32), g["level%d" % j]) and np.array_equal(np.asarray(ok, np.uint8), g["ok%d" % j]), j
        assert np.array_equal(np.asarray(pt, np.uint8), g["patch%d" % j]), j
        assert np.array_equal(np.asarray(px, np.float32).view(np.uint32), g["px%d" % j].view(np.uint32)), j
        aligned += int(np.asarray(ok).sum())
    assert aligned > 1500


def test_oracle_reproduces_the_reference_direct_projection_golden(oracle):
    _check(lambda: oracle.Extractor(1000, 1.2, 8, 20, 7),
           lambda ex, A, B, cur7, slot, ref7, ka, world, px0: ex.find_direct_projection_batch([A], B, cur7, D.CAM, slot, ref7, ka, world, px0))


@pytest.mark.gpu
def test_device_reproduces_the_reference_direct_projection_golden():
    from orb_ygz_slam_amd import Extractor, make_camera
    cam = make_camera(D.W, D.H)

    def run(ex, A, B, cur7, slot, ref7, ka, world, px0):
        ex.image_cache_reserve(2, D.W, D.H)
        ex.image_cache_put(0, A)
        ex.image_cache_put(1, B)
        return ex.find_direct_projection_batch(cam, 1, cur7, slot, ref7, ka, world, px0, want_patches=True)

    _check(lambda: Extractor(1000, 1.2, 8, 20, 7, max_width=D.W, max_height=D.H, max_batch=1), run)
