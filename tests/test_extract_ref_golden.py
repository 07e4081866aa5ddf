"""tests/golden/extract_ref.npz holds digests of what THE REFERENCE'S OWN src/ORBextractor.cc returned (tools/make_golden_extract_ref.py, run
where the checkout is: the reference compiled where it lies into oracle/_ref/libref_orbextractor.so over the OpenCV stand-in) for the 27 images /
configurations of tests/extract_ref_cases.py: every keypoint field and every descriptor byte.  The oracle (CPU tier) and the device (GPU tier) must
reproduce them wherever they run."""
import os

import numpy as np
import pytest

from tests import extract_ref_cases as C

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "extract_ref.npz")


def _check(extract):
    g = np.load(GOLD)
    cs = C.cases()
    assert [c[0] for c in cs] == [str(x) for x in g["names"]]
    total = 0
    for (name, img, cfg), cnt, dig in zip(cs, g["counts"], g["digests"]):
        k, d = extract(img, cfg)
        assert len(k) == int(cnt), (name, len(k), int(cnt))
        assert C.digest(k, d) == str(dig), name
        total += len(k)
    assert total > 20000


def test_oracle_reproduces_the_reference_extractor_golden(oracle):
    _check(lambda img, cfg: oracle.Extractor(*cfg).extract(img))


@pytest.mark.gpu
def test_device_reproduces_the_reference_extractor_golden():
    from orb_ygz_slam_amd import Extractor
    cache = {}

    def extract(img, cfg):
        key = (img.shape, cfg)
        if key not in cache:
            cache.clear()       # one context at a time
            cache[key] = Extractor(*cfg, max_width=img.shape[1], max_height=img.shape[0], max_batch=1)
        return cache[key].extract(img)

    _check(extract)
