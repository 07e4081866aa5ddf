"""The oracle's matcher against THE REFERENCE'S OWN src/ORBmatcher.cc (compiled where it lies into oracle/_ref/libref_orbmatcher.so by
oracle/Makefile, against oracle/ref_shim/: OpenCV stand-in + plain-data Frame / KeyFrame / MapPoint + 3x3 arithmetic stand-ins for
Eigen / Sophus -- see oracle/ref_orbmatcher_capi.cpp).  Both sides get identical flat inputs through the same Python wrappers.

Pinned by this: SearchByProjection(F, MapPoints), SearchByProjection(Cur, Last), SearchByProjection(Cur, KF, found), SearchForInitialization,
SearchByBoW(KF, F), DescriptorDistance, ComputeThreeMaxima, RadiusByViewingCos -- rows a-10 ... a-14 of SURVEY 8a.  Not pinned: the Frame
grid (GetFeaturesInArea), MapPoint::PredictScale and the 3x3 float products, which are the oracle's restatements on both sides.

CPU tier; skipped where the library was never built."""
import numpy as np
import pytest

from oracle import oracle_py as O
from orb_ygz_slam_amd.scene import synth_frame

pytestmark = pytest.mark.skipif(O.ref_matcher_lib() is None, reason="oracle/_ref/libref_orbmatcher.so not built (reference checkout absent)")

CAM = dict(fx=458.654, fy=457.296, cx=367.215, cy=248.375)
W, H = 752, 480


def canon(m):
    return np.where(m == -2, -1, m)      # matched-then-culled slots: NULL either way


def unit_world(keys):
    return np.stack([(keys["x"] - np.float32(CAM["cx"])) / np.float32(CAM["fx"]), (keys["y"] - np.float32(CAM["cy"])) / np.float32(CAM["fy"]),
                     np.ones(len(keys), np.float32)], -1).astype(np.float32)


@pytest.fixture(scope="module")
def pair():
    base = synth_frame(50, W + 16, H + 16)
    a, b = base[8:8 + H, 8:8 + W], base[10:10 + H, 5:5 + W]
    oex = O.Extractor(1000, 1.2, 8, 20, 7)
    ka, da = oex.extract(a)
    kb, db = oex.extract(b)
    return ka, da, kb, db, oex.tables()["scale"]


def test_descriptor_distance(pair):
    ka, da, kb, db, sf = pair
    import ctypes
    L = O.ref_matcher_lib()
    L.yo_descriptor_distance.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    for i in range(0, min(len(da), len(db)), 7):
        assert L.yo_descriptor_distance(da[i].ctypes.data, db[i].ctypes.data) == int(np.unpackbits(da[i] ^ db[i]).sum())


def test_search_by_projection_last_equals_reference(pair):
    ka, da, kb, db, sf = pair
    rng = np.random.default_rng(3)
    n = len(ka)
    depth = rng.uniform(2.0, 8.0, n).astype(np.float32)
    world = unit_world(ka) * depth[:, None]
    ang = np.float32(np.deg2rad(0.5))
    Rcw = np.array([[np.cos(ang), 0, np.sin(ang)], [0, 1, 0], [-np.sin(ang), 0, np.cos(ang)]], np.float32)
    tcw = np.array([0.02, -0.01, 0.03], np.float32)
    angl = np.float32(np.deg2rad(-0.3))
    Rlw = np.array([[1, 0, 0], [0, np.cos(angl), -np.sin(angl)], [0, np.sin(angl), np.cos(angl)]], np.float32)
    tlw = np.array([-0.01, 0.02, 0.0], np.float32)
    I, z = np.eye(3, dtype=np.float32), np.zeros(3, np.float32)
    valid = (rng.uniform(size=n) > 0.1).astype(np.uint8)
    outl = (rng.uniform(size=n) > 0.9).astype(np.uint8)
    obs = (rng.uniform(size=n) > 0.3).astype(np.uint8)
    uright = np.where(rng.uniform(size=len(kb)) > 0.5, kb["x"] - 5.0, -1.0).astype(np.float32)
    owner0 = ((rng.uniform(size=len(kb)) > 0.9) * rng.integers(1, 3, len(kb))).astype(np.uint8)
    cases = [dict(th=15.0, mono=True, check_level=True, check_ori=True), dict(th=7.0, mono=False, check_level=True, check_ori=True, u_right=uright, mb=0.11, mbf=50.0),
             dict(th=30.0, mono=True, check_level=False, check_ori=False), dict(th=15.0, mono=False, check_level=True, check_ori=True, tz=0.5, mb=0.11, mbf=50.0),
             dict(th=15.0, mono=False, check_level=True, check_ori=True, tz=-0.5, mb=0.11, mbf=50.0),
             dict(th=15.0, mono=True, check_level=True, check_ori=True, last=(Rlw, tlw))]
    total = 0
    for cs in cases:
        cam = dict(CAM, mb=cs.get("mb", 0.0), mbf=cs.get("mbf", 0.0))
        t = tcw.copy()
        t[2] += cs.get("tz", 0.0)
        Rl, tl = cs.get("last", (I, z))
        kw = dict(mp_valid=valid, outlier=outl, mp_has_obs=obs, u_right=cs.get("u_right"), cur_owner=owner0)
        args = (kb, db, sf, W, H, cam, ka, world, da, Rcw, t, Rl, tl, cs["th"], cs["mono"], cs["check_level"], cs["check_ori"])
        e_n, e_m, e_o = O.search_by_projection_last(*args, **kw)
        with O.reference_matcher():
            r_n, r_m, r_o = O.search_by_projection_last(*args, **kw)
        assert r_n == e_n, (cs, r_n, e_n)
        assert (r_m == canon(e_m)).all() and (r_o == e_o).all(), cs
        total += e_n
    assert total > 300


def test_search_by_projection_mappoints_equals_reference(pair):
    ka, da, kb, db, sf = pair
    rng = np.random.default_rng(7)
    M = len(ka)
    px = (ka["x"] - 3.0 + rng.normal(0, 1.0, M)).astype(np.float32)
    py = (ka["y"] + 2.0 + rng.normal(0, 1.0, M)).astype(np.float32)
    vc = rng.uniform(0.99, 1.0, M).astype(np.float32)
    vc[::5] = rng.uniform(0.9, 0.998, len(vc[::5]))           # both branches of RadiusByViewingCos
    lvl = np.clip(ka["octave"] + rng.integers(-1, 2, M), 0, 7).astype(np.int32)
    tiv = (rng.uniform(size=M) > 0.15).astype(np.uint8)
    bad = (rng.uniform(size=M) > 0.95).astype(np.uint8)
    obs = (rng.uniform(size=M) > 0.2).astype(np.uint8)
    pxr = (px - 4.0).astype(np.float32)
    uright = np.where(rng.uniform(size=len(kb)) > 0.5, kb["x"] - 4.0, -1.0).astype(np.float32)
    owner0 = ((rng.uniform(size=len(kb)) > 0.9) * rng.integers(1, 3, len(kb))).astype(np.uint8)
    for cs in (dict(th=1.0, check_level=False, nnratio=0.8), dict(th=3.0, check_level=True, nnratio=0.8),
               dict(th=5.0, check_level=False, nnratio=0.6, stereo=True), dict(th=8.0, check_level=True, nnratio=0.9, stereo=True)):
        kw = dict(is_bad=bad, mp_has_obs=obs, owner=owner0)
        if cs.get("stereo"):
            kw.update(proj_xr=pxr, u_right=uright)
        args = (kb, db, sf, W, H, CAM, tiv, px, py, vc, lvl, da, cs["th"], cs["check_level"], cs["nnratio"])
        e_n, e_m, e_o = O.search_by_projection_mappoints(*args, **kw)
        with O.reference_matcher():
            r_n, r_m, r_o = O.search_by_projection_mappoints(*args, **kw)
        assert r_n == e_n and (r_m == canon(e_m)).all() and (r_o == e_o).all(), cs
        assert e_n > 50


def test_search_by_projection_keyframe_equals_reference(pair):
    ka, da, kb, db, sf = pair
    rng = np.random.default_rng(11)
    n = len(ka)
    depth = rng.uniform(2.0, 8.0, n).astype(np.float32)
    world = unit_world(ka) * depth[:, None]
    ang = np.float32(np.deg2rad(0.4))
    Rcw = np.array([[np.cos(ang), 0, np.sin(ang)], [0, 1, 0], [-np.sin(ang), 0, np.cos(ang)]], np.float32)
    tcw = np.array([0.03, 0.01, -0.02], np.float32)
    dist = np.linalg.norm(world, axis=1).astype(np.float32)
    mf_max = (dist * sf[ka["octave"]]).astype(np.float32)
    mf_min = (mf_max / sf[7]).astype(np.float32)
    max_inv, min_inv = (np.float32(1.2) * mf_max).astype(np.float32), (np.float32(0.8) * mf_min).astype(np.float32)
    usable = (rng.uniform(size=n) > 0.15).astype(np.uint8)
    owner0 = (rng.uniform(size=len(kb)) > 0.9).astype(np.uint8)
    log_sf = np.log(np.float32(1.2))
    # (ORBdist >= 256 is left out: with every candidate slot taken the reference writes mvpMapPoints[-1], src/ORBmatcher.cc:1430-1432)
    for th, orb_dist, ori in ((10.0, 100, True), (3.0, 64, True), (10.0, 100, False), (25.0, 200, True)):
        args = (kb, db, sf, W, H, CAM, usable, world, max_inv, min_inv, mf_max, ka["angle"], da, Rcw, tcw, log_sf, th, orb_dist, ori)
        e_n, e_m, e_o, _ = O.search_by_projection_kf(*args, owner=owner0)
        with O.reference_matcher():
            r_n, r_m, r_o, _ = O.search_by_projection_kf(*args, owner=owner0)
        assert r_n == e_n and (r_m == canon(e_m)).all(), (th, orb_dist, ori, r_n, e_n)
        assert ((r_o != 0) == (e_o != 0)).all()
        if th >= 10:
            assert e_n > 50


def test_search_for_initialization_equals_reference():
    base = synth_frame(70, W + 32, H + 32)
    a, b = base[16:16 + H, 16:16 + W], base[20:20 + H, 9:9 + W]
    oex = O.Extractor(2000, 1.2, 8, 20, 7)
    sf = oex.tables()["scale"]
    ka, da = oex.extract(a)
    kb, db = oex.extract(b)
    prev = np.stack([ka["x"], ka["y"]], -1).astype(np.float32)
    state = prev
    for window, ratio, ori in ((100, 0.9, True), (30, 0.9, True), (100, 0.6, False), (100, 0.05, True), (100, 0.9, True)):
        args = (ka, da, kb, db, sf, W, H, CAM, state, window, ratio, ori)
        e_n, e_m, e_p = O.search_for_initialization(*args)
        with O.reference_matcher():
            r_n, r_m, r_p = O.search_for_initialization(*args)
        assert r_n == e_n and (r_m == e_m).all() and (r_p == e_p).all(), (window, ratio, ori)
        state = e_p                                    # the next call starts from the updated vbPrevMatched
    assert e_n > 50


def _fake_feature_vector(desc, bits):
    node = (desc[:, 0].astype(np.int32) >> (8 - bits)) if bits <= 8 else ((desc[:, 0].astype(np.int32) << (bits - 8)) | (desc[:, 1] >> (16 - bits)))
    return {int(n): np.nonzero(node == n)[0].astype(np.int32) for n in np.unique(node)}


def _join(fv_kf, fv_f):
    nodes = sorted(set(fv_kf) & set(fv_f))
    ko, fo, ki, fi = [0], [0], [], []
    for n in nodes:
        ki.extend(fv_kf[n]); fi.extend(fv_f[n])
        ko.append(len(ki)); fo.append(len(fi))
    return np.array(ko, np.int32), np.array(ki, np.int32), np.array(fo, np.int32), np.array(fi, np.int32)


def test_search_by_bow_equals_reference(pair):
    ka, da, kb, db, sf = pair
    rng = np.random.default_rng(5)
    valid = (rng.uniform(size=len(ka)) > 0.1).astype(np.uint8)
    for bits, ratio, ori in ((4, 0.7, False), (6, 0.75, True), (1, 0.9, True), (10, 0.7, True)):
        ko, ki, fo, fi = _join(_fake_feature_vector(da, bits), _fake_feature_vector(db, bits))
        args = (ko, ki, fo, fi, valid, ka, da, kb, db, ratio, ori)
        e_n, e_m = O.search_by_bow(*args)
        with O.reference_matcher():
            r_n, r_m = O.search_by_bow(*args)
        assert r_n == e_n and (r_m == canon(e_m)).all(), (bits, ratio, ori, r_n, e_n)
        if bits <= 6:
            assert e_n > 20


def test_search_for_triangulation_equals_reference(pair):
    """SearchForTriangulation + CheckDistEpipolarLine (src/ORBmatcher.cc:596-741, :136-153): the reference's surviving pairs are the
    oracle's match12 >= 0, for epipoles outside and inside the image, mono / stereo keypoints, bOnlyStereo and the rotation check."""
    from tests.tri_cases import cases
    ka, da, kb, db, sf = pair
    sigma2 = (sf * sf).astype(np.float32)
    seen = {}
    for label, kw in cases(ka, da, kb, db):
        e_n, e_m = O.search_for_triangulation(scale_factors2=sf, level_sigma2_2=sigma2, **kw)
        with O.reference_matcher():
            r_n, r_m = O.search_for_triangulation(scale_factors2=sf, level_sigma2_2=sigma2, **kw)
        assert r_n == e_n and (r_m == canon(e_m)).all(), (label, r_n, e_n)
        assert e_n == int((e_m >= 0).sum())
        seen[label] = (e_n, int((e_m == -2).sum()))
    assert seen["lateral"][0] > 100 and seen["lateral"][1] > 0          # many pairs on their epipolar lines; the rotation check culls some
    assert seen["forward"][0] < seen["lateral"][0]                         # radial epipolar lines: fewer shifted pairs lie on them
    assert seen["epipole-stereo"][0] > seen["epipole-mono"][0]             # stereo keypoints skip the epipole-distance test (:668-673)
    assert 0 < seen["only-stereo"][0] < seen["lateral"][0]
    assert seen["zero-F"] == (0, 0)


@pytest.mark.parametrize("seed", range(6))
def test_fuzz_triangulation_equals_reference(seed):
    """Random scenes, node granularities, relative poses and flags through SearchForTriangulation."""
    from tests.tri_cases import fake_feature_vector, join, geometry
    rng = np.random.default_rng(4100 + seed)
    w, h = int(rng.integers(320, 800)), int(rng.integers(240, 520))
    base = synth_frame(4200 + seed, w + 16, h + 16)
    a, b = base[8:8 + h, 8:8 + w], base[10:10 + h, 5:5 + w]
    oex = O.Extractor(int(rng.choice([300, 900])), 1.2, 8, 20, 7)
    sf = oex.tables()["scale"]
    ka, da = oex.extract(a)
    kb, db = oex.extract(b)
    if len(ka) < 20 or len(kb) < 20:
        pytest.skip("too few keypoints")
    for _ in range(4):
        bits = int(rng.integers(1, 9))
        o1, i1, o2, i2 = join(fake_feature_vector(da, bits), fake_feature_vector(db, bits))
        ep = None if rng.uniform() < 0.5 else (rng.uniform(0, w), rng.uniform(0, h))
        F12, Cw1, R2w, t2w, cam2 = geometry(float(rng.choice([0.002, 0.05, 0.8])), float(rng.uniform(-0.5, 0.5)), epipole=ep)
        Cw1 = rng.uniform(-0.01, 0.01, 3).astype(np.float32)   # not consistent with F12: only the epipole moves, the matcher does not care
        stereo = rng.uniform() < 0.5
        kf1 = dict(keys=ka, desc=da, has_mp=(rng.uniform(size=len(ka)) < rng.uniform(0, 0.6)).astype(np.uint8),
                   u_right=np.where(rng.uniform(size=len(ka)) < 0.5, ka["x"] - 3, -1).astype(np.float32) if stereo else None)
        kf2 = dict(keys=kb, desc=db, has_mp=(rng.uniform(size=len(kb)) < rng.uniform(0, 0.6)).astype(np.uint8),
                   u_right=np.where(rng.uniform(size=len(kb)) < 0.5, kb["x"] - 3, -1).astype(np.float32) if stereo else None)
        kw = dict(off1=o1, idx1=i1, off2=o2, idx2=i2, kf1=kf1, kf2=kf2, scale_factors2=sf, level_sigma2_2=(sf * sf).astype(np.float32), F12=F12,
                  Cw1=Cw1, R2w=R2w, t2w=t2w, cam2=cam2, only_stereo=bool(stereo and rng.uniform() < 0.3), check_ori=bool(rng.uniform() < 0.7))
        e_n, e_m = O.search_for_triangulation(**kw)
        with O.reference_matcher():
            r_n, r_m = O.search_for_triangulation(**kw)
        assert r_n == e_n and (r_m == canon(e_m)).all(), (seed, bits, r_n, e_n)


@pytest.mark.parametrize("seed", range(8))
def test_fuzz_projection_searches_equal_reference(seed):
    """Random image pairs, poses, thresholds and flag combinations through the three projection searches."""
    rng = np.random.default_rng(1200 + seed)
    w, h = int(rng.integers(320, 800)), int(rng.integers(240, 520))
    nf = int(rng.choice([300, 800, 1500]))
    base = synth_frame(1300 + seed, w + 16, h + 16)
    dx, dy = int(rng.integers(0, 12)), int(rng.integers(0, 12))
    a, b = base[8:8 + h, 8:8 + w], base[dy:dy + h, dx:dx + w]
    oex = O.Extractor(nf, 1.2, 8, 20, 7)
    sf = oex.tables()["scale"]
    ka, da = oex.extract(a)
    kb, db = oex.extract(b)
    if len(ka) < 20 or len(kb) < 20:
        pytest.skip("too few keypoints")
    cam = dict(CAM, mb=0.11, mbf=40.0)
    n = len(ka)
    depth = rng.uniform(1.0, 9.0, n).astype(np.float32)
    world = np.stack([(ka["x"] - np.float32(CAM["cx"])) / np.float32(CAM["fx"]) * depth, (ka["y"] - np.float32(CAM["cy"])) / np.float32(CAM["fy"]) * depth,
                      depth], -1).astype(np.float32)
    ang = np.float32(np.deg2rad(rng.uniform(-1, 1)))
    Rcw = np.array([[np.cos(ang), 0, np.sin(ang)], [0, 1, 0], [-np.sin(ang), 0, np.cos(ang)]], np.float32)
    tcw = rng.uniform(-0.05, 0.05, 3).astype(np.float32)
    tcw[2] = np.float32(rng.choice([0.0, 0.4, -0.4]))
    I, z = np.eye(3, dtype=np.float32), np.zeros(3, np.float32)
    valid = (rng.uniform(size=n) > 0.1).astype(np.uint8)
    outl = (rng.uniform(size=n) > 0.9).astype(np.uint8)
    obs = (rng.uniform(size=n) > 0.3).astype(np.uint8)
    mono = bool(rng.integers(0, 2))
    uright = None if mono else np.where(rng.uniform(size=len(kb)) > 0.4, kb["x"] - rng.uniform(0, 30, len(kb)), -1.0).astype(np.float32)
    owner0 = ((rng.uniform(size=len(kb)) > 0.9) * rng.integers(1, 3, len(kb))).astype(np.uint8)
    th = float(rng.choice([3.0, 7.0, 15.0, 30.0]))
    chk, ori = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
    args = (kb, db, sf, w, h, cam, ka, world, da, Rcw, tcw, I, z, th, mono, chk, ori)
    kw = dict(mp_valid=valid, outlier=outl, mp_has_obs=obs, u_right=uright, cur_owner=owner0)
    e = O.search_by_projection_last(*args, **kw)
    with O.reference_matcher():
        r = O.search_by_projection_last(*args, **kw)
    assert r[0] == e[0] and (r[1] == canon(e[1])).all() and (r[2] == e[2]).all(), ("last", w, h, nf, th, mono, chk, ori)
    # F vs MapPoints
    M = n
    px = (ka["x"] + (8 - dx) + rng.normal(0, 1.5, M)).astype(np.float32)
    py = (ka["y"] + (8 - dy) + rng.normal(0, 1.5, M)).astype(np.float32)
    vc = rng.uniform(0.9, 1.0, M).astype(np.float32)
    lvl = np.clip(ka["octave"] + rng.integers(-1, 2, M), 0, 7).astype(np.int32)
    tiv = (rng.uniform(size=M) > 0.15).astype(np.uint8)
    bad = (rng.uniform(size=M) > 0.95).astype(np.uint8)
    nn = float(rng.choice([0.6, 0.8, 0.9]))
    args = (kb, db, sf, w, h, cam, tiv, px, py, vc, lvl, da, float(rng.choice([1.0, 3.0, 5.0])), chk, nn)
    kw = dict(is_bad=bad, mp_has_obs=obs, owner=owner0, proj_xr=(px - 6.0).astype(np.float32), u_right=uright)
    e = O.search_by_projection_mappoints(*args, **kw)
    with O.reference_matcher():
        r = O.search_by_projection_mappoints(*args, **kw)
    assert r[0] == e[0] and (r[1] == canon(e[1])).all() and (r[2] == e[2]).all(), ("mappoints", w, h, nf)
    # Cur vs KeyFrame
    dist = np.linalg.norm(world, axis=1).astype(np.float32)
    mf_max = (dist * sf[ka["octave"]]).astype(np.float32)
    mf_min = (mf_max / sf[7]).astype(np.float32)
    args = (kb, db, sf, w, h, cam, valid, world, (np.float32(1.2) * mf_max).astype(np.float32), (np.float32(0.8) * mf_min).astype(np.float32), mf_max,
            ka["angle"], da, Rcw, tcw, np.log(np.float32(1.2)), th, int(rng.choice([64, 100, 150])), ori)
    own1 = (owner0 != 0).astype(np.uint8)
    e = O.search_by_projection_kf(*args, owner=own1)
    with O.reference_matcher():
        r = O.search_by_projection_kf(*args, owner=own1)
    assert r[0] == e[0] and (r[1] == canon(e[1])).all() and ((r[2] != 0) == (e[2] != 0)).all(), ("kf", w, h, nf)


def _ref_find_direct_projection_batch(oex, ref_imgs, cur_img, cur_Tcw7, cam, ref_slot, ref_Tcw7, ref_kp, mp_world, px_curr):
    """ORBmatcher::FindDirectProjection run by the reference's own code (yr_find_direct_projection_batch); pyramids from the oracle."""
    import ctypes as C
    L = O.ref_matcher_lib()
    tb = oex.tables()
    nl = len(tb["scale"])
    pyr_refs = [oex.pyramid(np.ascontiguousarray(r, np.uint8)) for r in ref_imgs]
    pyr_cur = oex.pyramid(np.ascontiguousarray(cur_img, np.uint8))
    lw = np.array([p.shape[1] for p in pyr_cur], np.int32)
    lh = np.array([p.shape[0] for p in pyr_cur], np.int32)
    keep = [np.ascontiguousarray(l) for p in pyr_refs for l in p] + [np.ascontiguousarray(l) for l in pyr_cur]
    rp = (C.c_void_p * (len(pyr_refs) * nl))(*[a.ctypes.data for a in keep[:len(pyr_refs) * nl]])
    cp = (C.c_void_p * nl)(*[a.ctypes.data for a in keep[len(pyr_refs) * nl:]])
    sf = np.ascontiguousarray(tb["scale"], np.float32)
    ils = np.ascontiguousarray(tb["inv_sigma2"] if "inv_sigma2" in tb else 1.0 / (tb["scale"] * tb["scale"]), np.float32)
    ct = np.ascontiguousarray(cur_Tcw7, np.float32)
    rs = np.ascontiguousarray(ref_slot, np.int32)
    rt = np.ascontiguousarray(ref_Tcw7, np.float32)
    rk = np.ascontiguousarray(ref_kp, O.KP_DTYPE)
    mw = np.ascontiguousarray(mp_world, np.float32)
    px = np.array(px_curr, np.float32).reshape(-1, 2).copy()
    n = len(rs)
    sl = np.zeros(max(n, 1), np.int32)
    ok = np.zeros(max(n, 1), np.uint8)
    pt = np.zeros((max(n, 1), 100), np.uint8)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    L.yr_find_direct_projection_batch.restype = None
    L.yr_find_direct_projection_batch.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                                  C.c_float, C.c_float, C.c_float, C.c_float, C.c_int] + [C.c_void_p] * 8
    L.yr_find_direct_projection_batch(nl, p(sf), p(ils), p(lw), p(lh), len(pyr_refs), rp, cp, p(ct), cam["fx"], cam["fy"], cam["cx"], cam["cy"], n,
                                      p(rs), p(rt), p(rk), p(mw), p(px), p(sl), p(ok), p(pt))
    return px, sl[:n], ok[:n], pt[:n]


def test_find_direct_projection_equals_reference():
    """FindDirectProjection + GetWarpAffineMatrix + WarpAffine + GetBestSearchLevel + the reference's own Align2D (src/Align.cc) on a rendered
    two-view scene: refined pixel, search level, success flag and the warped 10x10 patch, bit for bit.  (Pose algebra and the 2x2 / 3x3
    inverses are the oracle's conventions on both sides.)"""
    from orb_ygz_slam_amd.scene import rotvec_to_quat, two_view_scene
    w, h = 752, 480
    for seed, rv, tr in ((9, (0.01, -0.02, 0.03), (0.1, -0.05, 0.2)), (4, (-0.03, 0.01, -0.02), (-0.2, 0.1, -0.3))):
        A, B, (R, t), bp = two_view_scene(seed, w, h, CAM, Z=4.0, rotvec=rv, trans=tr)
        oex = O.Extractor(1000, 1.2, 8, 20, 7)
        ka, _ = oex.extract(A)
        world = bp(ka["x"], ka["y"])
        q = rotvec_to_quat(rv)
        cur7 = np.array([q[0], q[1], q[2], q[3], *tr], np.float32)
        ref7 = np.tile(np.array([0, 0, 0, 1, 0, 0, 0], np.float32), (len(ka), 1))
        Xc = (R @ world.T.astype(np.float64)).T + t
        u = CAM["fx"] * Xc[:, 0] / Xc[:, 2] + CAM["cx"]
        v = CAM["fy"] * Xc[:, 1] / Xc[:, 2] + CAM["cy"]
        rng = np.random.default_rng(seed)
        px0 = np.stack([u, v], -1) + rng.uniform(-2.5, 2.5, (len(ka), 2))
        px0[::17] += 400.0                                  # candidates that leave the image: Align2D breaks out, success = false
        slot = np.zeros(len(ka), np.int32)
        e_px, e_sl, e_ok, e_pt = oex.find_direct_projection_batch([A], B, cur7, CAM, slot, ref7, ka, world, px0)
        r_px, r_sl, r_ok, r_pt = _ref_find_direct_projection_batch(oex, [A], B, cur7, CAM, slot, ref7, ka, world, px0)
        assert (r_sl == e_sl).all() and (r_ok == e_ok).all() and (r_pt == e_pt).all()
        assert (r_px.view(np.uint32) == e_px.view(np.uint32)).all()
        assert e_ok.sum() > 0.5 * len(ka) and (e_ok == 0).sum() > 10


def test_sparse_img_align_equals_reference():
    """SparseImgAlign(max_level, min_level, n_iter).run(ref, cur, TCR): the reference's own src/SparseImageAlign.cc + NLSSolver (Gauss-Newton
    driver, level loop, caches, visibility, stop / rollback rules) against the oracle -- resulting SE3 (bit pattern), return value,
    number of linearisations, final chi2 and Hessian.  max_level <= 5: the reference indexes `int iterations[6]` with the level; n_iter = 10:
    run() overwrites the constructor's n_iter with iterations[level] = 10 on every level (src/SparseImageAlign.cc:39-44), the oracle's n_iter
    parameter generalises that."""
    from orb_ygz_slam_amd.scene import two_view_scene
    w, h = 752, 480
    ident = np.array([0, 0, 0, 1, 0, 0, 0], np.float32)
    cases = [(21, (0.004, -0.006, 0.003), (0.02, -0.01, 0.03), 5, 1, 10), (22, (-0.01, 0.004, 0.0), (-0.03, 0.02, 0.01), 4, 0, 10),
             (23, (0.002, 0.002, -0.008), (0.0, 0.0, 0.05), 5, 2, 10), (24, (0.03, -0.02, 0.01), (0.3, -0.2, 0.4), 5, 1, 10)]   # last: too far -> rollbacks
    for seed, rv, tr, max_level, min_level, n_iter in cases:
        A, B, _, bp = two_view_scene(seed, w, h, CAM, Z=3.0, rotvec=rv, trans=tr)
        oex = O.Extractor(600, 1.2, 8, 20, 7)
        k, _ = oex.extract(A)
        world = bp(k["x"], k["y"])
        pa, pb = oex.pyramid(A), oex.pyramid(B)
        inv = oex.tables()["inv_scale"]
        rng = np.random.default_rng(seed)
        valid = (rng.uniform(size=len(k)) > 0.1).astype(np.uint8)
        outl = (rng.uniform(size=len(k)) > 0.95).astype(np.uint8)
        args = (k, world, ident, pa, ident, pb, inv, CAM, max_level, min_level, n_iter)
        e_ret, e_T, e_info, e_H = O.sparse_img_align(*args, mp_valid=valid, outlier=outl)
        with O.reference_matcher():
            r_ret, r_T, r_info, r_H = O.sparse_img_align(*args, mp_valid=valid, outlier=outl)
        assert r_ret == e_ret and r_info[0] == e_info[0], (seed, r_ret, e_ret, r_info, e_info)
        assert (r_T.view(np.uint32) == e_T.view(np.uint32)).all(), (seed, r_T, e_T)
        assert r_info[1] == e_info[1] and (r_H == e_H).all()
        assert e_ret > 100
    # no features: "SparseImgAlign: no features to track!" -> 0
    with O.reference_matcher():
        assert O.sparse_img_align(k[:0], world[:0], ident, pa, ident, pb, inv, CAM, 5, 1)[0] == 0
