"""tests/golden/frame_ref.npz holds what THE REFERENCE'S OWN src/Frame.cc and src/MapPoint.cc returned (tools/make_golden_frame_ref.py, run where
the checkout is) on the scenarios of tests/frame_ref_cases.py: 300 GetFeaturesInArea windows (index lists in the reference's order), isInFrustum over
1000 MapPoints at two viewing-cosine limits (flag, projections, predicted level, cosine -- bit patterns), ComputeDistinctiveDescriptors of twelve
tracks (the winning descriptor).  The oracle (CPU tier) and the device (GPU tier) must reproduce those bytes wherever they run."""
import os

import numpy as np
import pytest

from tests import frame_ref_cases as C

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "frame_ref.npz")


def _check(extractor, fia, frustum, distinctive):
    g = np.load(GOLD)
    k, d, sf = C.frame(extractor)
    wins = C.windows()
    lists = fia(k, sf, wins)
    assert np.array_equal(np.array([len(x) for x in lists], np.int32), g["fia_cnt"])
    assert np.array_equal(np.concatenate(lists).astype(np.int32), g["fia_idx"])
    fin = C.frustum_inputs(k, sf)
    for limit in (0.5, 0.9):
        red = C.frustum_reduce(frustum(k, d, sf, fin, limit))
        for name, a in red.items():
            want = g["frustum%g_%s" % (limit, name)]
            assert a.dtype == want.dtype and np.array_equal(a.view(np.uint8), want.view(np.uint8)), (limit, name)
    off, desc = C.tracks()
    best = distinctive(off, desc)
    assert np.array_equal(np.stack([desc[off[p] + best[p]] for p in range(len(off) - 1)]), g["distinctive_desc"])


def test_oracle_reproduces_the_reference_frame_golden(oracle):
    _check(oracle.Extractor(1000, 1.2, 8, 20, 7),
           lambda k, sf, wins: [oracle.features_in_area(k, sf, C.W, C.H, *w) for w in wins],
           lambda k, d, sf, fin, limit: oracle.is_in_frustum(k, d, sf, C.W, C.H, C.CAM, *fin, limit),
           oracle.distinctive_descriptors)


@pytest.mark.gpu
def test_device_reproduces_the_reference_frame_golden():
    from orb_ygz_slam_amd import Extractor, make_camera
    ex = Extractor(1000, 1.2, 8, 20, 7, max_width=C.W, max_height=C.H, max_batch=1)
    cam = make_camera(C.W, C.H, mb=C.CAM["mb"], mbf=C.CAM["mbf"])

    def fia(k, sf, wins):
        xyr = np.array([w[:3] for w in wins], np.float32)
        lv = np.array([w[3:] for w in wins], np.int32)
        return ex.features_in_area(cam, k, xyr, levels=lv)[0]

    _check(ex, fia, lambda k, d, sf, fin, limit: ex.is_in_frustum_batch(cam, *fin, limit), ex.distinctive_descriptors_batch)
