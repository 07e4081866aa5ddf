"""GPU parity of Frame::ComputeStereoMatches (src/Frame.cc:509-682) against the oracle: mvuRight / mvDepth bit-identical (all float
expressions are evaluated in source order on both sides), in the host-array form and in the batch-resident form."""
import numpy as np
import pytest

from orb_ygz_slam_amd.scene import stereo_scene

pytestmark = pytest.mark.gpu
MB, MBF = 0.11, 47.9   # EuRoC stereo: baseline [m], fx * baseline


def _same(a, b):
    return np.array_equal(a.view(np.uint32), b.view(np.uint32))


@pytest.mark.parametrize("w,h,nfeat,seed", [(752, 480, 1200, 3), (640, 480, 1000, 4), (1280, 720, 2500, 5)])
def test_compute_stereo_matches(oracle, w, h, nfeat, seed):
    from orb_ygz_slam_amd import Extractor
    left, right, bh, ds = stereo_scene(seed, w, h)
    ex = Extractor(nfeat, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=2)
    oex = oracle.Extractor(nfeat, 1.2, 8, 20, 7)
    kl, dl = ex.extract(left)
    kr, dr = ex.extract(right)
    ur, dp = ex.compute_stereo_matches(left, right, kl, dl, kr, dr, MB, MBF)
    our, odp = oex.compute_stereo_matches(left, right, kl, dl, kr, dr, MB, MBF)
    assert _same(ur, our) and _same(dp, odp)
    ok = our >= 0
    assert ok.sum() > 0.25 * len(kl)
    # and the matches recover the rendered disparities
    band = np.minimum((kl["y"][ok] // bh).astype(int), len(ds) - 1)
    err = np.abs((kl["x"][ok] - our[ok]) - np.array(ds, np.float32)[band])
    assert np.median(err) < 0.5


def test_compute_stereo_matches_uhd_config5(oracle):
    """BASELINE config 5 shape: 3840x2160 stereo pair, 12 pyramid levels, 8000 features per eye -- extraction of both eyes and
    ComputeStereoMatches (batch-resident form, as bench.py --stereo runs it) bit-identical to the oracle."""
    from orb_ygz_slam_amd import Extractor
    w, h = 3840, 2160
    left, right, bh, ds = stereo_scene(31, w, h)
    ex = Extractor(8000, 1.2, 12, 20, 7, max_width=w, max_height=h, max_batch=2)
    oex = oracle.Extractor(8000, 1.2, 12, 20, 7)
    ex.extract_batch_host(np.stack([left, right]))
    ex.stereo_batch(MB, MBF)
    kl, dl = ex.batch_fetch(0)
    kr, dr = ex.batch_fetch(1)
    okl, odl = oex.extract(left)
    okr, odr = oex.extract(right)
    assert len(kl) == len(okl) > 7000 and (kl == okl).all() and (dl == odl).all()
    assert len(kr) == len(okr) and (kr == okr).all() and (dr == odr).all()
    ur, dp = ex.stereo_fetch(0)
    our, odp = oex.compute_stereo_matches(left, right, kl, dl, kr, dr, MB, MBF)
    assert _same(ur[:len(kl)], our) and _same(dp[:len(kl)], odp)
    assert (our >= 0).sum() > 1000
    ur2, dp2 = ex.compute_stereo_matches(left, right, kl, dl, kr, dr, MB, MBF)     # host-array form
    assert _same(ur2, our) and _same(dp2, odp)


def test_stereo_batch_resident(oracle):
    from orb_ygz_slam_amd import Extractor
    w, h = 752, 480
    pairs = [stereo_scene(s, w, h) for s in (10, 11, 12)]
    imgs = np.stack([im for p in pairs for im in (p[0], p[1])])
    ex = Extractor(1000, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=len(imgs))
    oex = oracle.Extractor(1000, 1.2, 8, 20, 7)
    ex.extract_batch_host(imgs)
    ex.stereo_batch(MB, MBF)
    for p in range(len(pairs)):
        kl, dl = ex.batch_fetch(2 * p)
        kr, dr = ex.batch_fetch(2 * p + 1)
        ur, dp = ex.stereo_fetch(p)
        our, odp = oex.compute_stereo_matches(imgs[2 * p], imgs[2 * p + 1], kl, dl, kr, dr, MB, MBF)
        assert _same(ur[:len(kl)], our) and _same(dp[:len(kl)], odp)
        assert (our >= 0).sum() > 100


def test_stereo_degenerate(oracle):
    """No right keypoints / unrelated right image: everything stays -1 and the empty median cut is skipped."""
    from orb_ygz_slam_amd import Extractor
    from orb_ygz_slam_amd.synth import synth_frame
    from orb_ygz_slam_amd.capi import KP_DTYPE
    w, h = 640, 480
    left = synth_frame(20, w, h)
    ex = Extractor(800, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=2)
    oex = oracle.Extractor(800, 1.2, 8, 20, 7)
    kl, dl = ex.extract(left)
    flat = np.full((h, w), 100, np.uint8)
    ur, dp = ex.compute_stereo_matches(left, flat, kl, dl, np.zeros(0, KP_DTYPE), np.zeros((0, 32), np.uint8), MB, MBF)
    assert (ur == -1).all() and (dp == -1).all()
    other = synth_frame(21, w, h)
    kr, dr = ex.extract(other)
    ur, dp = ex.compute_stereo_matches(left, other, kl, dl, kr, dr, MB, MBF)
    our, odp = oex.compute_stereo_matches(left, other, kl, dl, kr, dr, MB, MBF)
    assert _same(ur, our) and _same(dp, odp)
