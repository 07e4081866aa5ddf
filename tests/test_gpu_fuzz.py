"""GPU fuzz parity: random configurations of the matcher, stereo, DSO and extractor paths against the oracle.  The number of seeds per
test is YGZF_FUZZ_SEEDS (default 6) so that the suite stays short; run with a larger value when hunting."""
import os

import numpy as np
import pytest

from orb_ygz_slam_amd.synth import synth_frame

pytestmark = pytest.mark.gpu
SEEDS = range(int(os.environ.get("YGZF_FUZZ_SEEDS", "6")))


def _cfg(rng, max_w=800, max_h=620, max_feat=2500):
    w, h = int(rng.integers(160, max_w)), int(rng.integers(140, max_h))
    nl = int(rng.integers(2, 10))
    sf = float(rng.choice([1.1, 1.2, 1.2, 1.25, 1.5, 2.0]))
    nf = int(rng.integers(100, max_feat))
    return w, h, nl, sf, nf


@pytest.mark.parametrize("seed", SEEDS)
def test_fuzz_special_images(oracle, seed):
    """Degenerate image content: flat, saturated, 1-px checkerboard, ramps, sparse impulses, hard binary edges."""
    from orb_ygz_slam_amd import Extractor
    rng = np.random.default_rng(300 + seed)
    w, h, nl, sf, nf = _cfg(rng)
    yy, xx = np.mgrid[0:h, 0:w]
    imgs = [np.zeros((h, w), np.uint8), np.full((h, w), 255, np.uint8), (((xx + yy) & 1) * 255).astype(np.uint8),
            ((xx * 255) // max(w - 1, 1)).astype(np.uint8), ((xx // 7 + yy // 5) % 2 * 200 + 20).astype(np.uint8),
            (rng.uniform(size=(h, w)) > 0.995).astype(np.uint8) * 255, (rng.integers(0, 2, (h, w)) * 255).astype(np.uint8),
            np.clip(synth_frame(seed, w, h).astype(np.int32) * 3 - 200, 0, 255).astype(np.uint8)]
    ex = Extractor(nf, sf, nl, 20, 7, max_width=w, max_height=h, max_batch=len(imgs))
    oex = oracle.Extractor(nf, sf, nl, 20, 7)
    ex.extract_batch_host(np.stack(imgs))
    for f, img in enumerate(imgs):
        k, d = ex.batch_fetch(f)
        ok, od = oex.extract(img)
        assert len(k) == len(ok) and (k == ok).all() and (d == od).all(), (w, h, nl, sf, nf, f)


@pytest.mark.parametrize("seed", SEEDS)
def test_fuzz_search_by_projection_last(oracle, seed):
    from orb_ygz_slam_amd import Extractor, make_camera, EUROC
    rng = np.random.default_rng(400 + seed)
    w, h, nl, sf, nf = _cfg(rng, max_feat=1800)
    sf = 1.2 if sf > 1.3 else sf                 # frame pairs need some overlap across levels to produce matches
    base = synth_frame(500 + seed, w + 16, h + 16)
    dx, dy = int(rng.integers(0, 12)), int(rng.integers(0, 12))
    a, b = base[8:8 + h, 8:8 + w], base[dy:dy + h, dx:dx + w]
    ex = Extractor(nf, sf, nl, 20, 7, max_width=w, max_height=h, max_batch=1)
    oex = oracle.Extractor(nf, sf, nl, 20, 7)
    sfs = oex.tables()["scale"]
    ka, da = ex.extract(a)
    kb, db = ex.extract(b)
    if len(ka) == 0 or len(kb) == 0:
        pytest.skip("no keypoints")
    cam_d = dict(EUROC, mb=0.11, mbf=50.0)
    cam = make_camera(w, h, mb=0.11, mbf=50.0)
    n = len(ka)
    depth = rng.uniform(1.0, 9.0, n).astype(np.float32)
    world = np.stack([(ka["x"] - np.float32(EUROC["cx"])) / np.float32(EUROC["fx"]) * depth,
                      (ka["y"] - np.float32(EUROC["cy"])) / np.float32(EUROC["fy"]) * depth, depth], -1).astype(np.float32)
    ang = np.float32(np.deg2rad(rng.uniform(-1, 1)))
    Rcw = np.array([[np.cos(ang), 0, np.sin(ang)], [0, 1, 0], [-np.sin(ang), 0, np.cos(ang)]], np.float32)
    tcw = rng.uniform(-0.05, 0.05, 3).astype(np.float32)
    tcw[2] = np.float32(rng.choice([0.0, 0.4, -0.4]))                       # still / forward / backward
    I, z = np.eye(3, dtype=np.float32), np.zeros(3, np.float32)
    valid = (rng.uniform(size=n) > 0.1).astype(np.uint8)
    outl = (rng.uniform(size=n) > 0.9).astype(np.uint8)
    obs = (rng.uniform(size=n) > 0.3).astype(np.uint8)
    mono = bool(rng.integers(0, 2))
    uright = None if mono else np.where(rng.uniform(size=len(kb)) > 0.4, kb["x"] - rng.uniform(0, 30, len(kb)), -1.0).astype(np.float32)
    owner0 = ((rng.uniform(size=len(kb)) > 0.9) * rng.integers(1, 3, len(kb))).astype(np.uint8)
    th = float(rng.choice([3.0, 7.0, 15.0, 30.0]))
    chk, ori = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
    e = oracle.search_by_projection_last(kb, db, sfs, w, h, cam_d, ka, world, da, Rcw, tcw, I, z, th, mono, chk, ori, mp_valid=valid, outlier=outl,
                                         mp_has_obs=obs, u_right=uright, cur_owner=owner0)
    g = ex.search_by_projection_last(cam, kb, db, ka, world, da, Rcw, tcw, I, z, th, mono, chk, ori, mp_valid=valid, outlier=outl, mp_has_obs=obs,
                                     u_right=uright, cur_owner=owner0, scale_factors=sfs)
    assert g[0] == e[0] and (g[1] == e[1]).all() and (g[2] == e[2]).all(), (w, h, nl, sf, nf, th, mono, chk, ori)


@pytest.mark.parametrize("seed", SEEDS)
def test_fuzz_stereo(oracle, seed):
    from orb_ygz_slam_amd import Extractor
    from orb_ygz_slam_amd.scene import stereo_scene
    rng = np.random.default_rng(600 + seed)
    w, h, nl, sf, nf = _cfg(rng)
    ds = tuple(int(x) for x in rng.integers(0, 60, int(rng.integers(1, 9))))
    left, right, _, _ = stereo_scene(700 + seed, w, h, disparities=ds, noise=int(rng.integers(0, 6)))
    ex = Extractor(nf, sf, nl, 20, 7, max_width=w, max_height=h, max_batch=2)
    oex = oracle.Extractor(nf, sf, nl, 20, 7)
    kl, dl = ex.extract(left)
    kr, dr = ex.extract(right)
    mb = float(rng.choice([0.11, 0.5, 0.05]))
    mbf = float(rng.choice([47.9, 20.0, 200.0]))
    ur, dp = ex.compute_stereo_matches(left, right, kl, dl, kr, dr, mb, mbf)
    our, odp = oex.compute_stereo_matches(left, right, kl, dl, kr, dr, mb, mbf)
    assert np.array_equal(ur.view(np.uint32), our.view(np.uint32)) and np.array_equal(dp.view(np.uint32), odp.view(np.uint32)), (w, h, nl, sf, nf, mb, mbf)


@pytest.mark.parametrize("seed", SEEDS)
def test_fuzz_dso(oracle, seed):
    from orb_ygz_slam_amd import Extractor
    rng = np.random.default_rng(800 + seed)
    w, h, nl, sf, nf = _cfg(rng, max_feat=3000)
    nf = max(nf, int(w * h / 3600) + 1)              # initial grid <= 60 px
    img = synth_frame(900 + seed, w, h) if seed % 3 else (rng.integers(0, 256, (h, w), dtype=np.uint8) // 2 + 60).astype(np.uint8)
    ex = Extractor(nf, sf, nl, 20, 7, max_width=w, max_height=h, max_batch=1)
    oex = oracle.Extractor(nf, sf, nl, 20, 7)
    k0, _ = ex.extract(img)
    inv = oex.tables()["inv_scale"]
    sizes = np.array([ex.level_size(w, h, l) for l in range(nl)])
    lx, ly = k0["x"] * inv[k0["octave"]], k0["y"] * inv[k0["octave"]]
    keep = (lx >= 16) & (ly >= 16) & (lx < sizes[k0["octave"], 0] - 16) & (ly < sizes[k0["octave"], 1] - 16)
    existing = k0[keep][::int(rng.integers(2, 6))].copy()
    existing["x"] += rng.uniform(-0.4, 0.4, len(existing)).astype(np.float32)
    g_gpu = g_cpu = int(rng.choice([-1, 12, 25, 40]))
    for it in range(2):
        kg, dg, g_gpu = ex.extract_dso(img, existing=existing if it else None, grid_size=g_gpu)
        ko, do, g_cpu = oex.extract_dso(img, existing=existing if it else None, grid_size=g_cpu)
        assert g_gpu == g_cpu and len(kg) == len(ko), (w, h, nl, sf, nf, it, g_gpu, g_cpu, len(kg), len(ko))
        assert (kg == ko).all() and (dg == do).all(), (w, h, nl, sf, nf, it)
