"""CPU tests of the oracle's matcher and aligner restatements: definition-level properties (no reference vectors
exist for this path)."""
import numpy as np
import pytest
from scipy.linalg import expm

from orb_ygz_slam_amd.capi import EUROC, KP_DTYPE
from orb_ygz_slam_amd.scene import two_view_scene, rotvec_to_quat, quat_to_R
from orb_ygz_slam_amd.synth import synth_frame


def test_features_in_area_vs_bruteforce(oracle):
    ex = oracle.Extractor(1000, 1.2, 8, 20, 7)
    k, d = ex.extract(synth_frame(0, 752, 480))
    sf = ex.tables()["scale"]
    rng = np.random.default_rng(0)
    gw, gh = np.float32(64) / np.float32(752), np.float32(48) / np.float32(480)
    cellx = np.round((k["x"] * gw).astype(np.float32)).astype(int)
    celly = np.round((k["y"] * gh).astype(np.float32)).astype(int)
    in_grid = (cellx >= 0) & (cellx < 64) & (celly >= 0) & (celly < 48)       # PosInGrid uses round(): last half cell dropped
    for _ in range(200):
        x, y, r = rng.uniform(-20, 780), rng.uniform(-20, 500), rng.uniform(3, 60)
        lo, hi = ((-1, -1), (2, 4), (0, 3), (3, -1))[rng.integers(0, 4)]
        got = set(oracle.features_in_area(k, sf, 752, 480, x, y, r, lo, hi).tolist())
        x32, y32, r32 = np.float32(x), np.float32(y), np.float32(r)
        ok = (np.abs(k["x"] - x32) < r32) & (np.abs(k["y"] - y32) < r32) & in_grid
        if lo > 0 or hi >= 0:
            ok &= k["octave"] >= lo
            if hi >= 0:
                ok &= k["octave"] <= hi
        # the cell window [floor, ceil] always covers the |d| < r box, so brute force == grid lookup
        assert got == set(np.nonzero(ok)[0].tolist())


def test_search_by_projection_identity(oracle):
    ex = oracle.Extractor(1000, 1.2, 8, 20, 7)
    k, d = ex.extract(synth_frame(1, 752, 480))
    world = np.stack([(k["x"] - np.float32(EUROC["cx"])) / np.float32(EUROC["fx"]),
                      (k["y"] - np.float32(EUROC["cy"])) / np.float32(EUROC["fy"]), np.ones(len(k), np.float32)], -1)
    I, z = np.eye(3, dtype=np.float32), np.zeros(3, np.float32)
    n, m, o = oracle.search_by_projection_last(k, d, ex.tables()["scale"], 752, 480, EUROC, k, world, d, I, z, I, z, 15.0,
                                               check_ori=False)
    # same frame, identity pose: every keypoint finds itself at distance 0 (first minimum in visiting order) unless an
    # identical descriptor sits earlier in its window
    assert n >= 0.98 * len(k)
    hit = m >= 0
    assert (m[hit] == np.nonzero(hit)[0]).mean() > 0.98
    assert (o[hit] == 2).all()
    # with the rotation check all matches fall in bin 0 and survive
    n2, m2, _ = oracle.search_by_projection_last(k, d, ex.tables()["scale"], 752, 480, EUROC, k, world, d, I, z, I, z, 15.0)
    assert n2 == n and (m2 == m).all()
    # outliers / invalid map points are skipped
    valid = np.ones(len(k), np.uint8)
    valid[::2] = 0
    n3, m3, _ = oracle.search_by_projection_last(k, d, ex.tables()["scale"], 752, 480, EUROC, k, world, d, I, z, I, z, 15.0,
                                                 mp_valid=valid)
    assert set(m3[m3 >= 0].tolist()) <= set(np.nonzero(valid)[0].tolist())


def test_se3_exp_matches_matrix_exponential(oracle):
    rng = np.random.default_rng(0)
    for _ in range(50):
        a = rng.normal(0, 0.3, 6)
        if _ % 10 == 0:
            a[3:] *= 1e-7          # Taylor branch below SophusConstants<float>::epsilon
        q7 = oracle.se3_exp(a.astype(np.float32))
        M = np.zeros((4, 4))
        wx, wy, wz = a[3:]
        M[:3, :3] = [[0, -wz, wy], [wz, 0, -wx], [-wy, wx, 0]]
        M[:3, 3] = a[:3]
        E = expm(M)
        assert np.abs(quat_to_R(q7[:4]) - E[:3, :3]).max() < 2e-6
        assert np.abs(q7[4:] - E[:3, 3]).max() < 2e-6
        inv = oracle.se3_inverse(q7)
        ident = oracle.se3_mul(q7, inv)
        assert np.abs(ident - np.array([0, 0, 0, 1, 0, 0, 0])).max() < 1e-6


def _align_inputs(oracle, seed=3, nfeat=600):
    w, h = 752, 480
    imgA, imgB, (R, t), backproject = two_view_scene(seed, w, h, EUROC)
    ex = oracle.Extractor(nfeat, 1.2, 8, 20, 7)
    k, _ = ex.extract(imgA)
    pyrA = ex.pyramid(imgA)
    pyrB = ex.pyramid(imgB)
    world = backproject(k["x"], k["y"])
    return dict(w=w, h=w, k=k, world=world, pyrA=pyrA, pyrB=pyrB, inv=ex.tables()["inv_scale"], R=R, t=t)


def test_sparse_img_align_recovers_motion(oracle):
    s = _align_inputs(oracle)
    ident = np.array([0, 0, 0, 1, 0, 0, 0], np.float32)
    ret, T, info, H = oracle.sparse_img_align(s["k"], s["world"], ident, s["pyrA"], ident, s["pyrB"], s["inv"], EUROC, 7, 1)
    assert ret > 100
    R_est = quat_to_R(T[:4])
    ang = np.degrees(np.arccos(np.clip((np.trace(R_est.T @ s["R"]) - 1) / 2, -1, 1)))
    assert ang < 0.05, ang
    assert np.abs(T[4:] - s["t"]).max() < 5e-3, (T[4:], s["t"])
    assert np.allclose(H, H.T, rtol=1e-3, atol=1e-2) and (np.linalg.eigvalsh(H.astype(np.float64)) > 0).all()
    # no features -> 0 (reference src/SparseImageAlign.cc:24-27)
    r0, T0, _, _ = oracle.sparse_img_align(s["k"][:0], s["world"][:0], ident, s["pyrA"], ident, s["pyrB"], s["inv"], EUROC, 7, 1)
    assert r0 == 0
    # all outliers -> nothing visible -> 0 measurements
    r1, _, _, _ = oracle.sparse_img_align(s["k"], s["world"], ident, s["pyrA"], ident, s["pyrB"], s["inv"], EUROC, 7, 1,
                                          outlier=np.ones(len(s["k"]), np.uint8))
    assert r1 == 0


def test_device_order_mode_is_the_same_algorithm(oracle):
    """The oracle's device-order mode (the summation the HIP kernel is compared with bit for bit) differs from the reference-order mode by rounding
    only: same measurement count, same iteration count, SE3 within a few 1e-7 on a well-conditioned scene; and it is deterministic."""
    from orb_ygz_slam_amd.capi import EUROC
    from orb_ygz_slam_amd.scene import two_view_scene
    w, h = 752, 480
    A, B, _, bp = two_view_scene(11, w, h, EUROC, Z=3.0, rotvec=(0.002, -0.003, 0.001), trans=(0.01, -0.003, 0.002))
    oex = oracle.Extractor(600, 1.2, 8, 20, 7)
    k, _ = oex.extract(A)
    world = bp(k["x"], k["y"])
    ident = np.array([0, 0, 0, 1, 0, 0, 0], np.float32)
    inv = oex.tables()["inv_scale"]
    pa, pb = oex.pyramid(A), oex.pyramid(B)
    r = oracle.sparse_img_align(k, world, ident, pa, ident, pb, inv, EUROC, 7, 1)
    d = oracle.sparse_img_align(k, world, ident, pa, ident, pb, inv, EUROC, 7, 1, device_order=True)
    d2 = oracle.sparse_img_align(k, world, ident, pa, ident, pb, inv, EUROC, 7, 1, device_order=True)
    assert r[0] == d[0] > 300 and abs(r[2][0] - d[2][0]) <= 1
    assert np.abs(r[1] - d[1]).max() < 2e-6 and not np.array_equal(r[1], d[1])
    assert np.array_equal(d[1], d2[1]) and np.array_equal(d[3], d2[3])
    assert np.allclose(np.asarray(r[3]), np.asarray(d[3]), rtol=1e-4)


def test_search_for_triangulation_against_reference_golden():
    """tests/golden/triangulation_ref.npz holds what the REFERENCE'S OWN src/ORBmatcher.cc returned for the cases of tests/tri_cases.py
    (tools/make_golden_triangulation.py, run where the reference checkout is): the oracle must reproduce it anywhere -- the file travels, the
    checkout does not."""
    import hashlib
    import os
    from oracle import oracle_py as O
    from orb_ygz_slam_amd.scene import synth_frame
    from tests.tri_cases import cases
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "triangulation_ref.npz"))
    w, h = 752, 480
    base = synth_frame(50, w + 16, h + 16)
    a, b = base[8:8 + h, 8:8 + w], base[10:10 + h, 5:5 + w]
    oex = O.Extractor(1000, 1.2, 8, 20, 7)
    ka, da = oex.extract(a)
    kb, db = oex.extract(b)
    hsh = hashlib.sha256()
    for arr in (ka, da, kb, db):
        hsh.update(np.ascontiguousarray(arr).tobytes())
    assert hsh.hexdigest() == str(g["inputs_sha256"]), "the inputs of the golden cases drifted (extractor output changed?)"
    sf = oex.tables()["scale"]
    labels = [str(x) for x in g["labels"]]
    got = cases(ka, da, kb, db)
    assert [l for l, _ in got] == labels
    for label, kw in got:
        n, m = O.search_for_triangulation(scale_factors2=sf, level_sigma2_2=(sf * sf).astype(np.float32), **kw)
        assert n == int(g["n_" + label]) and (np.where(m == -2, -1, m) == g["m_" + label]).all(), label

