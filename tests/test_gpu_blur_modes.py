"""GPU tier: k_describe / k_describe_list in the three ygzf_cv_mode settings against the oracle in the same mode -- on ordinary synthetic
frames (where a handful of pixels per frame are exact ties) and on the tie images of tests/blur_cases.py, where a seventh of all columns
ties and every descriptor depends on the rounding rule.  The oracle's modes are pinned on the CPU tier (tests/test_oracle_blur_modes.py)."""
import numpy as np
import pytest

from orb_ygz_slam_amd.synth import synth_frame
from tests.blur_cases import tie_image

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("mode", [0, 1, 2])
def test_extract_matches_oracle_in_each_mode(oracle, mode):
    from orb_ygz_slam_amd import Extractor
    for (w, h), seed in (((752, 480), 21), ((641, 479), 22), ((322, 243), 23)):   # w % 4 = 0, 1, 2
        img = synth_frame(seed, w, h)
        k, d = Extractor(1000, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=1, cv_mode=mode).extract(img)
        with oracle.cv_mode(mode):
            ok, od = oracle.Extractor(1000, 1.2, 8, 20, 7).extract(img)
        assert len(k) == len(ok) and (k == ok).all()
        assert (d == od).all(), "mode %d: %d descriptor rows differ" % (mode, int((d != od).any(axis=1).sum()))


def _describe_both(oracle, img, keys_xy, angles, mode):
    from orb_ygz_slam_amd import Extractor
    from orb_ygz_slam_amd.capi import KP_DTYPE
    h, w = img.shape
    ex = Extractor(500, 1.2, 1, 20, 7, max_width=w, max_height=h, max_batch=1, cv_mode=mode)
    ex.extract_batch_host(img[None])
    keys = np.zeros(len(keys_xy) * len(angles), KP_DTYPE)
    i = 0
    for (x, y) in keys_xy:
        for a in angles:
            keys[i] = (x, y, 31.0, a, 0.0, 0, -1)
            i += 1
    _, d = ex.describe_keys(keys, frame=0, recompute_angle=False)
    oex = oracle.Extractor(500, 1.2, 1, 20, 7)
    with oracle.cv_mode(mode):
        b = oracle.blur(img)
    od = np.stack([oex.descriptor(b, float(k["x"]), float(k["y"]), float(k["angle"])) for k in keys])
    return d, od


@pytest.mark.parametrize("w", [160, 161, 162, 163])
def test_tie_image_descriptors(oracle, w):
    """Interior keys (aligned window path) and keys 19 px from the borders (byte path with REFLECT_101) on an image full of exact ties."""
    h = 120
    img = tie_image(7 + w, w, h)
    xy = [(x, y) for x in (19, 24, 40, 61, 83, w - 41, w - 23, w - 19) for y in (19, 33, 60, h - 19)]
    angles = [0.0, 12.5, 33.0, 90.0, 135.0, 181.0, 217.5, 300.25, 359.5]
    got = []
    for mode in (0, 1, 2):
        d, od = _describe_both(oracle, img, xy, angles, mode)
        assert (d == od).all(), "mode %d: %d of %d rows differ" % (mode, int((d != od).any(axis=1).sum()), len(d))
        got.append(d)
    assert (got[0] != got[1]).any() and (got[1] != got[2]).any()      # the image really separates the modes


@pytest.mark.parametrize("w", [69, 70, 71, 72])
def test_tail_rule(oracle, w):
    """Only the LAST column ties: it is sampled by keys 19 px from the right border at some angles; SSE2 mode must equal integer mode unless
    that column lies inside the vector body (w % 4 == 0)."""
    h = 64
    rng = np.random.default_rng(w)
    img = tie_image(11, w, h, tail_tie=True)
    img[:, : w - 4] = rng.integers(100, 156, (1, w - 4), dtype=np.uint8)          # destroy the body ties, keep the tail one
    xy = [(w - 19, y) for y in (19, 30, h - 19)]
    angles = [float(a) for a in np.arange(0, 360, 2.5)]
    d0, od0 = _describe_both(oracle, img, xy, angles, 0)
    d1, od1 = _describe_both(oracle, img, xy, angles, 1)
    assert (d0 == od0).all() and (d1 == od1).all()
    if w % 4:
        assert (d0 == d1).all()
