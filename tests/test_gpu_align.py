"""GPU parity test of SparseImgAlign::run (HIP persistent-workgroup Gauss-Newton) vs the oracle: SE3 output within 1e-5
(north_star tolerance; fp32, different summation order), same feature count, and both near the ground-truth motion."""
import numpy as np
import pytest

from orb_ygz_slam_amd.scene import two_view_scene, quat_to_R
from orb_ygz_slam_amd.synth import synth_frame

pytestmark = pytest.mark.gpu
TOL = 1e-5  # on the 7 SE3 parameters (unit quaternion + translation in metres)


def _case(oracle, ex, oex, seed, rotvec, trans, nfeat=600):
    from orb_ygz_slam_amd import make_camera, EUROC
    w, h = 752, 480
    imgA, imgB, (R, t), backproject = two_view_scene(seed, w, h, EUROC, rotvec=rotvec, trans=trans)
    k, _ = ex.extract(imgA)
    pyrA, pyrB = ex.compute_pyramid(imgA), ex.compute_pyramid(imgB)
    world = backproject(k["x"], k["y"])
    inv = oex.tables()["inv_scale"]
    ident = np.array([0, 0, 0, 1, 0, 0, 0], np.float32)
    o_ret, o_T, o_info, o_H = oracle.sparse_img_align(k, world, ident, pyrA, ident, pyrB, inv, EUROC, 7, 1)
    g_ret, g_T, g_info, g_H = ex.sia_run(make_camera(w, h), k, world, ident, pyrA, ident, pyrB, inv, 7, 1)
    return (o_ret, o_T, o_info, o_H), (g_ret, g_T, g_info, g_H), (R, t)


def test_sia_matches_oracle_and_ground_truth(oracle):
    from orb_ygz_slam_amd import Extractor
    ex = Extractor(600, 1.2, 8, 20, 7, max_width=752, max_height=480, max_batch=1)
    oex = oracle.Extractor(600, 1.2, 8, 20, 7)
    worst = 0.0
    for seed, rv, tr in ((3, (0.004, -0.006, 0.003), (0.03, -0.02, 0.015)), (4, (0.0, 0.0, 0.0), (0.0, 0.0, 0.0)),
                         (5, (-0.01, 0.008, -0.004), (-0.04, 0.03, -0.02)), (6, (0.002, 0.001, 0.012), (0.01, 0.01, 0.05))):
        (o_ret, o_T, o_info, o_H), (g_ret, g_T, g_info, g_H), (R, t) = _case(oracle, ex, oex, seed, rv, tr)
        assert g_ret == o_ret and g_ret > 100, (seed, g_ret, o_ret)
        d = float(np.abs(g_T - o_T).max())
        worst = max(worst, d)
        assert d <= TOL, (seed, d, g_T, o_T, g_info, o_info)
        assert np.abs(g_T[4:] - t).max() < 5e-3
        ang = np.degrees(np.arccos(np.clip((np.trace(quat_to_R(g_T[:4]).T @ R) - 1) / 2, -1, 1)))
        assert ang < 0.05
        assert np.allclose(g_H, o_H, rtol=2e-3, atol=1e-1 * np.abs(o_H).max() * 1e-3)
    print("max |dT| vs oracle:", worst)


def test_sia_edge_cases(oracle):
    from orb_ygz_slam_amd import Extractor, make_camera, EUROC
    ex = Extractor(600, 1.2, 8, 20, 7, max_width=752, max_height=480, max_batch=1)
    imgA, imgB, _, backproject = two_view_scene(7, 752, 480, EUROC)
    k, _ = ex.extract(imgA)
    pyrA, pyrB = ex.compute_pyramid(imgA), ex.compute_pyramid(imgB)
    world = backproject(k["x"], k["y"])
    inv = ex.tables()["inv_scale"]
    ident = np.array([0, 0, 0, 1, 0, 0, 0], np.float32)
    cam = make_camera(752, 480)
    ret, T, _, _ = ex.sia_run(cam, k[:0], world[:0], ident, pyrA, ident, pyrB, inv, 7, 1)
    assert ret == 0                                    # reference: "no features to track" -> 0
    ret, T, info, _ = ex.sia_run(cam, k, world, ident, pyrA, ident, pyrB, inv, 7, 1, outlier=np.ones(len(k), np.uint8))
    o = oracle.sparse_img_align(k, world, ident, pyrA, ident, pyrB, inv, EUROC, 7, 1, outlier=np.ones(len(k), np.uint8))
    assert ret == 0 and o[0] == 0                      # nothing visible -> singular system -> stop, 0 measurements
    assert np.abs(T - o[1]).max() <= TOL


def test_align_batch_prev_matches_host_api(oracle):
    """Device-resident batch form (ref = previous frame of the batch, unit-depth points) == ygzf_sia_run on the same data,
    including the pyramid carried over from the previous batch."""
    from orb_ygz_slam_amd import Extractor, make_camera, EUROC
    w, h = 752, 480
    imgA, imgB, _, _ = two_view_scene(11, w, h, EUROC, Z=1.0, rotvec=(0.002, -0.003, 0.001), trans=(0.004, -0.003, 0.002))
    imgC, _, _, _ = two_view_scene(12, w, h, EUROC)
    imgs = np.stack([imgA, imgB, imgC])
    ex = Extractor(600, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=3)
    cam = make_camera(w, h)
    ident = np.array([0, 0, 0, 1, 0, 0, 0], np.float32)
    inv = ex.tables()["inv_scale"]
    for rnd in range(2):
        ex.extract_batch_host(imgs)
        ex.align_batch_prev(cam, 7, 1, 10)
        res = [ex.align_fetch(f) for f in range(3)]
        kps = [ex.batch_fetch(f)[0] for f in range(3)]
        pyr = [[ex.batch_fetch_level(f, l) for l in range(8)] for f in range(3)]
        for f in range(3):
            if f == 0 and rnd == 0:
                assert res[0][0] == 0          # no predecessor yet
                continue
            rf = (f - 1) % 3                   # f == 0 in round 1: the carried last frame of the previous batch
            k = kps[rf]
            world = np.stack([(k["x"] - np.float32(EUROC["cx"])) / np.float32(EUROC["fx"]),
                              (k["y"] - np.float32(EUROC["cy"])) / np.float32(EUROC["fy"]), np.ones(len(k), np.float32)], -1)
            ret, T, info, _ = ex2_run(ex, cam, k, world, ident, pyr[rf], pyr[f], inv)
            assert res[f][0] == ret, (rnd, f, res[f][0], ret)
            assert np.abs(res[f][1] - T).max() <= 1e-6
    assert res[1][0] > 100 and np.abs(res[1][1][4:]).max() > 1e-3   # A -> B really moved


def test_align_batch_prev_after_extract_ahead(oracle):
    """ygzf_set_extract_ahead: ygzf_compute_pyramid queues the extraction behind the pyramid, and "counts as an extraction for
    ygzf_align_batch_prev" (ygzf.h) -- the previous frame's pyramid has to leave the context's pyramid buffer BEFORE the new frame's pyramid is
    written over it (round 3 copied it afterwards: pair 0 aligned the new frame against itself).  Frame by frame, as Tracking would call it."""
    from orb_ygz_slam_amd import Extractor, make_camera, EUROC
    w, h = 752, 480
    imgA, imgB, _, _ = two_view_scene(11, w, h, EUROC, Z=1.0, rotvec=(0.002, -0.003, 0.001), trans=(0.004, -0.003, 0.002))
    imgC, _, _, _ = two_view_scene(11, w, h, EUROC, Z=1.0, rotvec=(-0.001, 0.002, 0.002), trans=(-0.003, 0.002, 0.001))
    ex = Extractor(600, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=1)
    cam = make_camera(w, h)
    ident = np.array([0, 0, 0, 1, 0, 0, 0], np.float32)
    inv = ex.tables()["inv_scale"]
    ex.set_extract_ahead(True)
    prev = None
    moved = 0
    for step, img in enumerate((imgA, imgB, imgC, imgA)):
        pyr = ex.compute_pyramid(img)
        k, d = ex.extract_resident(w, h)
        ex.align_batch_prev(cam, 7, 1, 10)
        ret, T = ex.align_fetch(0)[:2]
        if step == 0:
            assert ret == 0                     # no predecessor (and the carry is only switched on by this first call)
        else:                                   # (the carry was switched on by the call of step 0: frame A's pyramid is kept when B's is computed)
            pk, ppyr = prev
            world = np.stack([(pk["x"] - np.float32(EUROC["cx"])) / np.float32(EUROC["fx"]),
                              (pk["y"] - np.float32(EUROC["cy"])) / np.float32(EUROC["fy"]), np.ones(len(pk), np.float32)], -1)
            want_ret, want_T, _, _ = ex2_run(ex, cam, pk, world, ident, ppyr, pyr, inv)
            assert ret == want_ret and ret > 100, (step, ret, want_ret)
            assert np.abs(T - want_T).max() <= 1e-6, (step, T, want_T)
            moved += np.abs(T[4:]).max() > 1e-3
        prev = (k, [np.ascontiguousarray(p) for p in pyr])
    assert moved >= 2                           # a frame aligned against itself would give the identity


def test_normal_equations_bit_identical_to_the_device_order_oracle(oracle):
    """One linearisation per pyramid level (n_iter = 1, max_level = min_level): the kernel's H, chi2, measurement count and updated SE3 equal the oracle's
    device-order mode bit for bit -- the kernel's per-feature moment formulation, its fused multiply-adds and its reduction tree are restated there
    (oracle/oracle_align.cpp) on top of the reference's algorithm.  And so does a full coarse-to-fine run."""
    from orb_ygz_slam_amd import Extractor, make_camera, EUROC
    w, h = 752, 480
    imgA, imgB, _, bp = two_view_scene(11, w, h, EUROC, Z=3.0, rotvec=(0.002, -0.003, 0.001), trans=(0.01, -0.003, 0.002))
    ex = Extractor(1000, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=1)
    k, _ = ex.extract(imgA)
    pa, pb = ex.compute_pyramid(imgA), ex.compute_pyramid(imgB)
    world = bp(k["x"], k["y"])
    inv = ex.tables()["inv_scale"]
    cam = make_camera(w, h)
    ident = np.array([0, 0, 0, 1, 0, 0, 0], np.float32)
    bits = lambda a: np.ascontiguousarray(np.asarray(a, np.float32)).reshape(-1).view(np.uint32)
    for (mx, mn, it) in [(l, l, 1) for l in range(8)] + [(7, 1, 10), (4, 0, 3)]:
        g = ex.sia_run(cam, k, world, ident, pa, ident, pb, inv, mx, mn, it)
        o = oracle.sparse_img_align(k, world, ident, pa, ident, pb, inv, EUROC, mx, mn, it, device_order=True)
        assert g[0] == o[0] > 500
        assert np.array_equal(bits(g[3]), bits(o[3])), ("H", mx, mn, it)
        assert np.array_equal(bits(g[2]), bits(o[2])), ("iterations / chi2", mx, mn, it, g[2], o[2])
        assert np.array_equal(bits(g[1]), bits(o[1])), ("SE3", mx, mn, it, g[1], o[1])
        r = oracle.sparse_img_align(k, world, ident, pa, ident, pb, inv, EUROC, mx, mn, it)      # the reference's own summation order: north_star's tolerance
        assert np.abs(g[1] - r[1]).max() <= TOL


def test_align_large_batch_equals_small_batches():
    """Launches of 128 pairs and more keep their workgroups to 74 KB of LDS (two per CU: the coarse levels are then gathered from L2 instead
    of a staged copy): the same bytes read, so the same poses bit for bit as launches of a few pairs."""
    from orb_ygz_slam_amd import Extractor, make_camera, EUROC
    w, h, n = 320, 240, 132
    base = synth_frame(55, w + 40, h + 40)
    imgs = np.stack([base[(3 * i) % 17:(3 * i) % 17 + h, (5 * i) % 23:(5 * i) % 23 + w] for i in range(n)])
    cam = make_camera(w, h, fx=200.0, fy=200.0, cx=160.0, cy=120.0)
    big = Extractor(300, 1.2, 6, 20, 7, max_width=w, max_height=h, max_batch=n)
    big.extract_batch_host(imgs)
    big.align_batch_prev(cam, 5, 1, 10)
    got = [big.align_fetch(f) for f in range(n)]
    small = Extractor(300, 1.2, 6, 20, 7, max_width=w, max_height=h, max_batch=12)
    want = []
    for s0 in range(0, n, 12):
        small.extract_batch_host(imgs[s0:s0 + 12])
        small.align_batch_prev(cam, 5, 1, 10)
        want += [small.align_fetch(f) for f in range(12)]
    assert got[0][0] == 0 and sum(r[0] > 50 for r in got) > n // 2
    for f in range(1, n):
        assert got[f][0] == want[f][0] and np.array_equal(got[f][1], want[f][1]), f


def ex2_run(ex, cam, k, world, ident, pyr_ref, pyr_cur, inv):
    from orb_ygz_slam_amd import Extractor
    global _EX2
    try:
        _EX2
    except NameError:
        _EX2 = Extractor(600, 1.2, 8, 20, 7, max_width=64, max_height=64, max_batch=1)
    return _EX2.sia_run(cam, k, world, ident, pyr_ref, ident, pyr_cur, inv, 7, 1)


@pytest.mark.parametrize("nfeat", [2400, 6200])
def test_sia_large_feature_counts(oracle, nfeat):
    """The kernel's LDS plan changes with the feature count: above ~1750 features the point-only Jacobian terms are rebuilt per iteration
    instead of read from LDS, and near the 6400-feature ceiling no pyramid level of the current frame is staged in LDS any more.  Same
    1e-5 bar against the oracle on both plans (features beyond what the extractor finds are jittered copies: the aligner only needs
    pixel positions and 3D points)."""
    from orb_ygz_slam_amd import Extractor, make_camera, EUROC
    w, h = 752, 480
    imgA, imgB, (R, t), backproject = two_view_scene(21, w, h, EUROC, rotvec=(0.003, -0.004, 0.002), trans=(0.02, -0.015, 0.01))
    ex = Extractor(3000, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=1)
    oex = oracle.Extractor(3000, 1.2, 8, 20, 7)
    k0, _ = ex.extract(imgA)
    rng = np.random.default_rng(nfeat)
    reps = -(-nfeat // len(k0))
    k = np.concatenate([k0] * reps)[:nfeat].copy()
    k["x"] = np.clip(k["x"] + rng.uniform(-2, 2, nfeat).astype(np.float32) * (np.arange(nfeat) >= len(k0)), 20, w - 21)
    k["y"] = np.clip(k["y"] + rng.uniform(-2, 2, nfeat).astype(np.float32) * (np.arange(nfeat) >= len(k0)), 20, h - 21)
    world = backproject(k["x"], k["y"])
    pyrA, pyrB = ex.compute_pyramid(imgA), ex.compute_pyramid(imgB)
    inv = oex.tables()["inv_scale"]
    ident = np.array([0, 0, 0, 1, 0, 0, 0], np.float32)
    o_ret, o_T, o_info, o_H = oracle.sparse_img_align(k, world, ident, pyrA, ident, pyrB, inv, EUROC, 7, 1)
    g_ret, g_T, g_info, g_H = ex.sia_run(make_camera(w, h), k, world, ident, pyrA, ident, pyrB, inv, 7, 1)
    assert g_ret == o_ret and g_ret > nfeat // 2, (g_ret, o_ret)
    assert float(np.abs(g_T - o_T).max()) <= TOL, (g_T, o_T, g_info, o_info)
    assert np.abs(g_T[4:] - t).max() < 5e-3


def test_sia_run_on_cached_slots_equals_the_host_pyramid_form(oracle):
    """ygzf_sia_run_cached reads the pyramids of two image-cache slots (level 0 uploaded once, levels rebuilt on the device by the resize
    kernel) instead of two host pyramids: same bytes, so the same TCR bit for bit -- and the oracle's within 1e-5."""
    from orb_ygz_slam_amd import Extractor, make_camera, EUROC
    w, h = 752, 480
    imgA, imgB, (R, t), backproject = two_view_scene(14, w, h, EUROC, rotvec=(0.003, 0.005, -0.002), trans=(-0.02, 0.01, 0.02))
    ex = Extractor(800, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=1)
    oex = oracle.Extractor(800, 1.2, 8, 20, 7)
    cam = make_camera(w, h)
    k, _ = ex.extract(imgA)
    world = backproject(k["x"], k["y"])
    pyrA, pyrB = ex.compute_pyramid(imgA), ex.compute_pyramid(imgB)
    inv = oex.tables()["inv_scale"]
    ident = np.array([0, 0, 0, 1, 0, 0, 0], np.float32)
    h_ret, h_T, h_info, h_H = ex.sia_run(cam, k, world, ident, pyrA, ident, pyrB, inv, 7, 1)
    ex.image_cache_reserve(3, w, h)
    ex.image_cache_put(2, imgA)
    ex.image_cache_put(0, imgB)
    c_ret, c_T, c_info, c_H = ex.sia_run_cached(cam, 2, 0, k, world, ident, ident, inv, 7, 1)
    assert c_ret == h_ret and np.array_equal(c_T, h_T) and np.array_equal(c_info, h_info) and np.array_equal(c_H, h_H)
    o_ret, o_T, _, _ = oracle.sparse_img_align(k, world, ident, pyrA, ident, pyrB, inv, EUROC, 7, 1)
    assert c_ret == o_ret and float(np.abs(c_T - o_T).max()) <= TOL
    # level 0 in the range (the slot's own image), and an empty slot is refused
    c0 = ex.sia_run_cached(cam, 2, 0, k, world, ident, ident, inv, 2, 0)
    h0 = ex.sia_run(cam, k, world, ident, pyrA, ident, pyrB, inv, 2, 0)
    assert c0[0] == h0[0] and np.array_equal(c0[1], h0[1])
    from orb_ygz_slam_amd import YgzfError
    with pytest.raises(YgzfError):
        ex.sia_run_cached(cam, 1, 0, k, world, ident, ident, inv, 7, 1)


def test_cache_slot_filled_from_the_extractor_that_holds_the_image(oracle):
    """ygzf_image_cache_put_resident: a slot filled device to device from the context whose ComputePyramid / extract saw the image last holds
    the same bytes as an uploaded one (SparseImgAlign on it returns the same TCR bit for bit, odd width included); refused when the source
    holds nothing, another size, or is the cache's own context."""
    from orb_ygz_slam_amd import Extractor, make_camera, EUROC, YgzfError
    for (w, h) in ((752, 480), (641, 479)):
        imgA, imgB, (R, t), backproject = two_view_scene(15, w, h, EUROC, rotvec=(0.002, -0.004, 0.001), trans=(0.02, 0.01, -0.01))
        fe = Extractor(800, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=1)      # the Frame's extractor
        cache = Extractor(1000, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=1)  # the image cache's context
        cam = make_camera(w, h)
        k, _ = fe.extract(imgA)
        world = backproject(k["x"], k["y"])
        inv = fe.tables()["inv_scale"]
        ident = np.array([0, 0, 0, 1, 0, 0, 0], np.float32)
        cache.image_cache_reserve(4, w, h)
        assert not cache.has_resident_image(w, h)
        with pytest.raises(YgzfError):
            cache.image_cache_put_resident(0, cache)
        fresh = Extractor(800, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=1)
        with pytest.raises(YgzfError):
            cache.image_cache_put_resident(0, fresh)             # holds nothing yet
        cache.image_cache_put(0, imgA)
        cache.image_cache_put(1, imgB)
        want = cache.sia_run_cached(cam, 0, 1, k, world, ident, ident, inv, 7, 1)
        fe.extract(imgA)                                         # an extraction leaves image + pyramid in the context
        assert fe.has_resident_image(w, h) and not fe.has_resident_image(w + 1, h)
        cache.image_cache_put_resident(2, fe)
        fe.set_extract_ahead(True)
        fe.compute_pyramid(imgB)                                 # so does ComputePyramid (with the extraction queued behind it)
        cache.image_cache_put_resident(3, fe)
        got = cache.sia_run_cached(cam, 2, 3, k, world, ident, ident, inv, 7, 1)
        assert got[0] == want[0] and all(np.array_equal(a, b) for a, b in zip(got[1:], want[1:]))
        kb, db = fe.extract_resident(w, h)                       # the queued extraction is still collected afterwards
        kb2, db2 = fe.extract(imgB)
        assert np.array_equal(kb, kb2) and np.array_equal(db, db2)
        small = Extractor(800, 1.2, 8, 20, 7, max_width=320, max_height=240, max_batch=1)
        small.compute_pyramid(imgA[:240, :320].copy())
        with pytest.raises(YgzfError):
            cache.image_cache_put_resident(0, small)             # another size


def test_in_kernel_reference_patches_agree():
    """Launches of up to 32 pairs build the reference patches of all levels in a kernel of their own (k_sia_precompute); larger ones -- and
    YGZF_FORCE=sia_precompute=0 -- keep the per-level phase inside k_sia_run.  Both share one body (sia_ref_patch): the aligner tests, the bit-identity
    against the device-order oracle included, must hold with the in-kernel form too."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    from orb_ygz_slam_amd.capi import force_env
    env = dict(os.environ, YGZF_FORCE=force_env(sia_precompute=0))
    out = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", "-p", "no:cacheprovider", os.path.join(root, "tests", "test_gpu_align.py"),
                          os.path.join(root, "tests", "test_gpu_fuzz.py"), "-k", "(align or sia) and not in_kernel_reference_patches"],
                         cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
    assert " passed" in out.stdout

