"""CPU tier: the oracle's Frame::ComputeStereoMatches (src/Frame.cc:509-682) on a rendered rectified pair with known disparities."""
import numpy as np

from orb_ygz_slam_amd.scene import stereo_scene


def test_oracle_stereo_recovers_disparities(oracle):
    w, h = 752, 480
    left, right, bh, ds = stereo_scene(3, w, h)
    ex = oracle.Extractor(1200, 1.2, 8, 20, 7)
    kl, dl = ex.extract(left)
    kr, dr = ex.extract(right)
    mb, mbf = 0.11, 47.9
    ur, dp = ex.compute_stereo_matches(left, right, kl, dl, kr, dr, mb, mbf)
    ok = ur >= 0
    assert ok.sum() > 300
    assert ((ur < 0) == (dp < 0)).all()
    disp = kl["x"][ok] - ur[ok]
    assert (disp > 0).all() and (disp < mbf / mb).all()
    assert np.allclose(dp[ok], np.float32(mbf) / disp, rtol=1e-6)
    band = np.minimum((kl["y"][ok] // bh).astype(int), len(ds) - 1)
    err = np.abs(disp - np.array(ds, np.float32)[band])
    assert np.median(err) < 0.6 and (err < 2.5).mean() > 0.85
    # no right keypoints -> nothing matched, and the (empty) median cut is skipped
    ur0, dp0 = ex.compute_stereo_matches(left, right, kl, dl, kr[:0], dr[:0], mb, mbf)
    assert (ur0 == -1).all() and (dp0 == -1).all()
