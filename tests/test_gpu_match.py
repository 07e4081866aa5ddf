"""GPU parity tests of the matcher hot path (ORBmatcher::SearchByProjection(Cur, Last) + Frame grid) vs the oracle:
result-identical match assignment (which Last keypoint's MapPoint lands in which Cur slot), ownership state and count."""
import numpy as np
import pytest

from orb_ygz_slam_amd.synth import synth_frame

pytestmark = pytest.mark.gpu


def _unit_world(keys, cam):
    w = np.zeros((len(keys), 3), np.float32)
    w[:, 0] = (keys["x"] - np.float32(cam["cx"])) / np.float32(cam["fx"])
    w[:, 1] = (keys["y"] - np.float32(cam["cy"])) / np.float32(cam["fy"])
    w[:, 2] = 1.0
    return w


@pytest.mark.parametrize("split", [0, 1, 2, 3, 5, 8])
def test_match_batch_prev_equals_oracle(oracle, split, monkeypatch):
    """`split`: workgroups per pair of k_match_last (YGZF_FORCE=match_split=n, read when the context is created; 0 = the library's choice, which is 8
    for a launch of five pairs, 1 = one workgroup per pair as in large batches): the same matches whichever way the queries are dealt."""
    from orb_ygz_slam_amd import Extractor, make_camera, EUROC
    from orb_ygz_slam_amd.capi import force_env
    monkeypatch.setenv("YGZF_FORCE", force_env(match_split=split if split else None))
    w, h = 752, 480
    base = synth_frame(40, w + 16, h + 16)
    # consecutive frames = shifted crops of one scene (+ a different scene) so that matches exist
    imgs = np.stack([base[8:8 + h, 8:8 + w], base[9:9 + h, 10:10 + w], base[6:6 + h, 11:11 + w], synth_frame(41, w, h),
                     base[7:7 + h, 9:9 + w]])
    ex = Extractor(1000, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=len(imgs))
    oex = oracle.Extractor(1000, 1.2, 8, 20, 7)
    cam = make_camera(w, h)
    I, z = np.eye(3, dtype=np.float32), np.zeros(3, np.float32)
    prev = None
    for rnd in range(2):   # second round exercises the carry of the previous batch's last frame
        ex.extract_batch_host(imgs)
        ex.match_batch_prev(cam, 15.0, True, True, True)
        counts = ex.match_counts()
        for f in range(len(imgs)):
            k, d = ex.batch_fetch(f)
            m, o = ex.match_fetch(f)
            m, o = m[:len(k)], o[:len(k)]
            if prev is None:
                assert counts[f] == 0 and (m == -1).all()
            else:
                pk, pd = prev
                exp_n, exp_m, exp_o = oracle.search_by_projection_last(k, d, oex.tables()["scale"], w, h, EUROC, pk,
                                                                       _unit_world(pk, EUROC), pd, I, z, I, z, 15.0)
                assert counts[f] == exp_n, (rnd, f, counts[f], exp_n)
                assert (m == exp_m).all() and (o == exp_o).all()
                if f in (1, 2):
                    assert exp_n > 100
            prev = (k, d)


def test_search_by_projection_last_variants(oracle):
    from orb_ygz_slam_amd import Extractor, make_camera, EUROC
    w, h = 752, 480
    base = synth_frame(50, w + 16, h + 16)
    a, b = base[8:8 + h, 8:8 + w], base[10:10 + h, 5:5 + w]
    ex = Extractor(1000, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=1)
    oex = oracle.Extractor(1000, 1.2, 8, 20, 7)
    sf = oex.tables()["scale"]
    ka, da = ex.extract(a)
    kb, db = ex.extract(b)
    rng = np.random.default_rng(3)
    n = len(ka)
    depth = rng.uniform(2.0, 8.0, n).astype(np.float32)
    world = _unit_world(ka, EUROC) * depth[:, None]
    # small relative motion: rotation about y by 0.5 deg + translation
    ang = np.float32(np.deg2rad(0.5))
    Rcw = np.array([[np.cos(ang), 0, np.sin(ang)], [0, 1, 0], [-np.sin(ang), 0, np.cos(ang)]], np.float32)
    tcw = np.array([0.02, -0.01, 0.03], np.float32)
    I, z = np.eye(3, dtype=np.float32), np.zeros(3, np.float32)
    valid = (rng.uniform(size=n) > 0.1).astype(np.uint8)
    outl = (rng.uniform(size=n) > 0.9).astype(np.uint8)
    obs = (rng.uniform(size=n) > 0.3).astype(np.uint8)
    uright = np.where(rng.uniform(size=len(kb)) > 0.5, kb["x"] - 5.0, -1.0).astype(np.float32)
    owner0 = (rng.uniform(size=len(kb)) > 0.95).astype(np.uint8) * 2
    cases = [dict(th=15.0, mono=True, check_level=True, check_ori=True),
             dict(th=7.0, mono=False, check_level=True, check_ori=True, u_right=uright, mb=0.11, mbf=50.0),
             dict(th=30.0, mono=True, check_level=False, check_ori=False),
             dict(th=15.0, mono=False, check_level=True, check_ori=True, tz=0.5, mb=0.11, mbf=50.0),    # bForward
             dict(th=15.0, mono=False, check_level=True, check_ori=True, tz=-0.5, mb=0.11, mbf=50.0)]   # bBackward
    for cs in cases:
        cam_d = dict(EUROC, mb=cs.get("mb", 0.0), mbf=cs.get("mbf", 0.0))
        cam = make_camera(w, h, mb=cam_d["mb"], mbf=cam_d["mbf"])
        t = tcw.copy()
        t[2] += cs.get("tz", 0.0)
        kw = dict(mp_valid=valid, outlier=outl, mp_has_obs=obs, u_right=cs.get("u_right"), cur_owner=owner0)
        e_n, e_m, e_o = oracle.search_by_projection_last(kb, db, sf, w, h, cam_d, ka, world, da, Rcw, t, I, z, cs["th"], cs["mono"],
                                                         cs["check_level"], cs["check_ori"], **kw)
        g_n, g_m, g_o = ex.search_by_projection_last(cam, kb, db, ka, world, da, Rcw, t, I, z, cs["th"], cs["mono"], cs["check_level"],
                                                     cs["check_ori"], scale_factors=sf, **kw)
        assert g_n == e_n, (cs, g_n, e_n)
        assert (g_m == e_m).all() and (g_o == e_o).all(), cs
    # empty sides
    g_n, g_m, g_o = ex.search_by_projection_last(make_camera(w, h), kb, db, ka[:0], world[:0], da[:0], I, z, I, z, 15.0)
    assert g_n == 0 and (g_m == -1).all()


def test_search_by_projection_mappoints(oracle):
    """ORBmatcher::SearchByProjection(F, MapPoints, th, checkLevel) (Tracking::SearchLocalPoints) vs the oracle."""
    from orb_ygz_slam_amd import Extractor, make_camera, EUROC
    w, h = 752, 480
    base = synth_frame(60, w + 16, h + 16)
    a, b = base[8:8 + h, 8:8 + w], base[9:9 + h, 11:11 + w]
    ex = Extractor(1000, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=1)
    oex = oracle.Extractor(1000, 1.2, 8, 20, 7)
    sf = oex.tables()["scale"]
    ka, da = ex.extract(a)     # "local map points": keypoints of frame A, projected where they land in B (+ jitter)
    kb, db = ex.extract(b)
    rng = np.random.default_rng(7)
    M = len(ka)
    px = (ka["x"] - 3.0 + rng.normal(0, 1.0, M)).astype(np.float32)
    py = (ka["y"] - 1.0 + rng.normal(0, 1.0, M)).astype(np.float32)
    vc = rng.uniform(0.99, 1.0, M).astype(np.float32)
    lvl = np.clip(ka["octave"] + rng.integers(-1, 2, M), 0, 7).astype(np.int32)
    tiv = (rng.uniform(size=M) > 0.15).astype(np.uint8)
    bad = (rng.uniform(size=M) > 0.95).astype(np.uint8)
    obs = (rng.uniform(size=M) > 0.2).astype(np.uint8)
    pxr = (px - 4.0).astype(np.float32)
    uright = np.where(rng.uniform(size=len(kb)) > 0.5, kb["x"] - 4.0, -1.0).astype(np.float32)
    owner0 = (rng.uniform(size=len(kb)) > 0.93).astype(np.uint8) * 2
    for cs in (dict(th=1.0, check_level=False, nnratio=0.8), dict(th=3.0, check_level=True, nnratio=0.8),
               dict(th=5.0, check_level=False, nnratio=0.6, stereo=True), dict(th=8.0, check_level=True, nnratio=0.9, stereo=True)):
        kw = dict(is_bad=bad, mp_has_obs=obs, owner=owner0)
        if cs.get("stereo"):
            kw.update(proj_xr=pxr, u_right=uright)
        e_n, e_m, e_o = oracle.search_by_projection_mappoints(kb, db, sf, w, h, EUROC, tiv, px, py, vc, lvl, da, cs["th"], cs["check_level"],
                                                              cs["nnratio"], **kw)
        g_n, g_m, g_o = ex.search_by_projection_mappoints(make_camera(w, h), kb, db, tiv, px, py, vc, lvl, da, cs["th"], cs["check_level"],
                                                          cs["nnratio"], scale_factors=sf, **kw)
        assert g_n == e_n, (cs, g_n, e_n)
        assert (g_m == e_m).all() and (g_o == e_o).all(), cs
        assert e_n > 50


def test_search_by_projection_keyframe(oracle):
    """SearchByProjection(Cur, KF, found, th, ORBdist) (relocalisation refinement, src/ORBmatcher.cc:1352-1469): the oracle runs the whole
    function; the device gets the host prologue's (valid, u, v, level) and must reproduce assignment, ownership and count."""
    from orb_ygz_slam_amd import Extractor, make_camera, EUROC
    w, h = 752, 480
    base = synth_frame(60, w + 16, h + 16)
    a, b = base[8:8 + h, 8:8 + w], base[11:11 + h, 4:4 + w]
    ex = Extractor(1000, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=1)
    oex = oracle.Extractor(1000, 1.2, 8, 20, 7)
    sf = oex.tables()["scale"]
    ka, da = ex.extract(a)      # the KeyFrame
    kb, db = ex.extract(b)      # the current frame
    cam = make_camera(w, h)
    rng = np.random.default_rng(11)
    n = len(ka)
    depth = rng.uniform(2.0, 8.0, n).astype(np.float32)
    world = _unit_world(ka, EUROC) * depth[:, None]
    ang = np.float32(np.deg2rad(0.4))
    Rcw = np.array([[np.cos(ang), 0, np.sin(ang)], [0, 1, 0], [-np.sin(ang), 0, np.cos(ang)]], np.float32)
    tcw = np.array([0.03, 0.01, -0.02], np.float32)
    # scale-invariance range of each point as MapPoint::UpdateNormalAndDepth sets it from its reference observation
    dist = np.linalg.norm(world, axis=1).astype(np.float32)
    mf_max = (dist * sf[ka["octave"]]).astype(np.float32)
    mf_min = (mf_max / sf[7]).astype(np.float32)
    max_inv, min_inv = (np.float32(1.2) * mf_max).astype(np.float32), (np.float32(0.8) * mf_min).astype(np.float32)
    usable = (rng.uniform(size=n) > 0.15).astype(np.uint8)          # NULL / bad / already found
    owner0 = (rng.uniform(size=len(kb)) > 0.9).astype(np.uint8)       # slots the first relocalisation pass filled
    log_sf = np.log(np.float32(1.2))
    for th, orb_dist, ori in ((10.0, 100, True), (3.0, 64, True), (10.0, 100, False), (25.0, 256, True)):
        e_n, e_m, e_o, (valid, u, v, lvl) = oracle.search_by_projection_kf(kb, db, sf, w, h, EUROC, usable, world, max_inv, min_inv, mf_max,
                                                                          ka["angle"], da, Rcw, tcw, log_sf, th, orb_dist, ori, owner=owner0)
        g_n, g_m, g_o = ex.search_by_projection_kf(cam, kb, db, valid, u, v, lvl, ka["angle"], da, th, orb_dist, ori, owner=owner0,
                                                   scale_factors=sf)
        assert valid.sum() > 300
        assert g_n == e_n and (g_m == e_m).all()
        assert ((g_o != 0) == (e_o != 0)).all()
        if th >= 10:
            assert e_n > 50


def test_search_for_initialization(oracle):
    """SearchForInitialization (monocular initialisation, src/ORBmatcher.cc:375-478): matches12, count and the updated vbPrevMatched."""
    from orb_ygz_slam_amd import Extractor, make_camera, EUROC
    w, h = 752, 480
    base = synth_frame(70, w + 32, h + 32)
    a, b = base[16:16 + h, 16:16 + w], base[20:20 + h, 9:9 + w]
    ex = Extractor(2000, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=1)     # the Ini extractor: 2 x nFeatures
    oex = oracle.Extractor(2000, 1.2, 8, 20, 7)
    sf = oex.tables()["scale"]
    ka, da = ex.extract(a)
    kb, db = ex.extract(b)
    cam = make_camera(w, h)
    prev = np.stack([ka["x"], ka["y"]], -1).astype(np.float32)                      # vbPrevMatched starts as F1's own positions
    for window, ratio, ori in ((100, 0.9, True), (30, 0.9, True), (100, 0.6, False), (100, 0.05, True)):
        e_n, e_m, e_p = oracle.search_for_initialization(ka, da, kb, db, sf, w, h, EUROC, prev, window, ratio, ori)
        g_n, g_m, g_p = ex.search_for_initialization(cam, ka, da, kb, db, prev, window, ratio, ori, scale_factors=sf)
        assert g_n == e_n and (g_m == e_m).all() and (g_p == e_p).all()
        if ratio >= 0.6:
            assert e_n > 50
    # a second round starts from the updated positions (the reference calls it once per frame until initialisation succeeds)
    e_n, e_m, e_p = oracle.search_for_initialization(ka, da, kb, db, sf, w, h, EUROC, prev, 100, 0.9, True)
    e_n2, e_m2, e_p2 = oracle.search_for_initialization(ka, da, kb, db, sf, w, h, EUROC, e_p, 100, 0.9, True)
    g_n2, g_m2, g_p2 = ex.search_for_initialization(cam, ka, da, kb, db, e_p, 100, 0.9, True, scale_factors=sf)
    assert g_n2 == e_n2 and (g_m2 == e_m2).all() and (g_p2 == e_p2).all()


def _fake_feature_vector(desc, bits):
    """Stand-in for DBoW2's FeatureVector (the vocabulary blob is not shipped): node id = leading descriptor bits."""
    node = (desc[:, 0].astype(np.int32) >> (8 - bits)) if bits <= 8 else ((desc[:, 0].astype(np.int32) << (bits - 8)) | (desc[:, 1] >> (16 - bits)))
    return {int(n): np.nonzero(node == n)[0].astype(np.int32) for n in np.unique(node)}


def _join(fv_kf, fv_f):
    nodes = sorted(set(fv_kf) & set(fv_f))               # the merge-join of :169-247
    ko, fo, ki, fi = [0], [0], [], []
    for n in nodes:
        ki.extend(fv_kf[n]); fi.extend(fv_f[n])
        ko.append(len(ki)); fo.append(len(fi))
    return np.array(ko, np.int32), np.array(ki, np.int32), np.array(fo, np.int32), np.array(fi, np.int32)


def test_search_by_bow(oracle):
    """SearchByBoW(KF, F) per-node brute force (src/ORBmatcher.cc:155-263) on a joined node list."""
    from orb_ygz_slam_amd import Extractor
    w, h = 752, 480
    base = synth_frame(80, w + 16, h + 16)
    a, b = base[8:8 + h, 8:8 + w], base[9:9 + h, 10:10 + w]
    ex = Extractor(1000, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=1)
    ka, da = ex.extract(a)
    kb, db = ex.extract(b)
    rng = np.random.default_rng(5)
    valid = (rng.uniform(size=len(ka)) > 0.1).astype(np.uint8)
    for bits, ratio, ori in ((4, 0.7, False), (6, 0.75, True), (1, 0.9, True), (10, 0.7, True)):
        ko, ki, fo, fi = _join(_fake_feature_vector(da, bits), _fake_feature_vector(db, bits))
        e_n, e_m = oracle.search_by_bow(ko, ki, fo, fi, valid, ka, da, kb, db, ratio, ori)
        g_n, g_m = ex.search_by_bow(ko, ki, fo, fi, valid, ka, da, kb, db, ratio, ori)
        assert g_n == e_n and (g_m == e_m).all()
        if bits <= 6:
            assert e_n > 20


def test_search_for_triangulation(oracle):
    """SearchForTriangulation + CheckDistEpipolarLine (src/ORBmatcher.cc:596-741, :136-153) against the oracle: the cases of the CPU pin
    (tests/test_ref_matcher.py holds the oracle to the reference's own source on exactly these) + the context's tables instead of KF2's."""
    from orb_ygz_slam_amd import Extractor
    from tests.tri_cases import cases
    w, h = 752, 480
    base = synth_frame(50, w + 16, h + 16)
    a, b = base[8:8 + h, 8:8 + w], base[10:10 + h, 5:5 + w]
    ex = Extractor(1000, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=1)
    ka, da = ex.extract(a)
    kb, db = ex.extract(b)
    sf = oracle.Extractor(1000, 1.2, 8, 20, 7).tables()["scale"]
    sg = (sf * sf).astype(np.float32)
    total = 0
    for label, kw in cases(ka, da, kb, db):
        e_n, e_m = oracle.search_for_triangulation(scale_factors2=sf, level_sigma2_2=sg, **kw)
        g_n, g_m = ex.search_for_triangulation(scale_factors2=sf, level_sigma2_2=sg, **kw)
        assert g_n == e_n and (g_m == e_m).all(), (label, g_n, e_n)
        g_n2, g_m2 = ex.search_for_triangulation(scale_factors2=None, level_sigma2_2=None, **kw)   # the extractor's own tables
        assert g_n2 == e_n and (g_m2 == e_m).all(), label
        total += e_n
    assert total > 1500
    # other tables than the extractor's: a coarser sigma2 admits more pairs
    label, kw = cases(ka, da, kb, db)[0]
    sg4 = (4 * sg).astype(np.float32)
    e_n, e_m = oracle.search_for_triangulation(scale_factors2=sf, level_sigma2_2=sg4, **kw)
    g_n, g_m = ex.search_for_triangulation(scale_factors2=sf, level_sigma2_2=sg4, **kw)
    assert g_n == e_n and (g_m == e_m).all()
    # degenerate inputs: no common node, an empty KeyFrame
    z = np.zeros(1, np.int32)
    kw0 = dict(kw, off1=z, idx1=z[:0], off2=z, idx2=z[:0])
    assert ex.search_for_triangulation(scale_factors2=sf, level_sigma2_2=sg, **kw0)[0] == 0
    empty = dict(keys=ka[:0], desc=da[:0], has_mp=np.zeros(0, np.uint8), u_right=None)
    kw1 = dict(kw, kf1=empty, off1=np.zeros_like(kw["off1"]), idx1=z[:0])
    n, m = ex.search_for_triangulation(scale_factors2=sf, level_sigma2_2=sg, **kw1)
    assert n == 0 and len(m) == 0


def test_search_for_triangulation_against_reference_golden():
    """The device against tests/golden/triangulation_ref.npz: the surviving pairs the reference's own src/ORBmatcher.cc produced for these cases
    (tools/make_golden_triangulation.py); culled slots read -1 there (the reference returns pairs only)."""
    import hashlib
    import os
    from orb_ygz_slam_amd import Extractor
    from tests.tri_cases import cases
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "triangulation_ref.npz"))
    w, h = 752, 480
    base = synth_frame(50, w + 16, h + 16)
    a, b = base[8:8 + h, 8:8 + w], base[10:10 + h, 5:5 + w]
    ex = Extractor(1000, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=1)
    ka, da = ex.extract(a)
    kb, db = ex.extract(b)
    hsh = hashlib.sha256()
    for arr in (ka, da, kb, db):
        hsh.update(np.ascontiguousarray(arr).tobytes())
    assert hsh.hexdigest() == str(g["inputs_sha256"]), "device keypoints / descriptors differ from the golden cases' inputs"
    for label, kw in cases(ka, da, kb, db):
        n, m = ex.search_for_triangulation(scale_factors2=None, level_sigma2_2=None, **kw)
        assert n == int(g["n_" + label]) and (np.where(m == -2, -1, m) == g["m_" + label]).all(), label


def test_search_for_triangulation_rejects_bad_input():
    from orb_ygz_slam_amd import Extractor
    from orb_ygz_slam_amd.capi import YgzfError
    from tests.tri_cases import cases
    w, h = 640, 480
    a = synth_frame(51, w, h)
    ex = Extractor(500, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=1)
    ka, da = ex.extract(a)
    label, kw = cases(ka, da, ka, da)[0]
    bad = dict(kw, idx2=kw["idx2"] + len(ka))
    with pytest.raises(YgzfError):
        ex.search_for_triangulation(scale_factors2=None, level_sigma2_2=None, **bad)
    k2 = ka.copy()
    k2["octave"][0] = 9
    with pytest.raises(YgzfError):
        ex.search_for_triangulation(scale_factors2=None, level_sigma2_2=None, **dict(kw, kf2=dict(kw["kf2"], keys=k2)))
    off = kw["off1"].copy()
    off[1], off[2] = off[2], off[1]
    if off[1] != off[2]:
        with pytest.raises(YgzfError):
            ex.search_for_triangulation(scale_factors2=None, level_sigma2_2=None, **dict(kw, off1=off))


@pytest.mark.parametrize("plan", ["match_serial=1", "match_serial=2", "match_split=1", "match_lanes=fixed"])
def test_in_order_plans_agree(plan):
    """The matcher resolves the reference's in-order semantics as a block-wide fixpoint (match_kernels.hip); the one-wave in-order pass it
    replaced stays as the fall-back (list extensions used up, no convergence).  YGZF_FORCE=match_serial=1 runs that pass alone, =2 the fixpoint
    with no extension slots, i.e. with the hand-over at the first exhausted list; match_split=1 keeps a pair on ONE workgroup (the form
    of 256-pair launches, which the single pairs of the tests otherwise never take), match_lanes=fixed gives every query eight lanes.
    The matcher tests and the projected-search fuzzers must hold against the oracle under each."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    from orb_ygz_slam_amd.capi import force_env
    env = dict(os.environ, YGZF_FORCE=force_env(**dict([plan.split("=")])))
    out = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", "-p", "no:cacheprovider", os.path.join(root, "tests", "test_gpu_match.py"),
                          os.path.join(root, "tests", "test_gpu_fuzz.py"), "-k",
                          "(search_by_projection or search_by_bow or search_for_initialization or test_fuzz_projected_searches or test_fuzz_search_by_projection_last) and not in_order_plans"],
                         cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
    assert " passed" in out.stdout


def test_matcher_is_deterministic_over_repeats(oracle):
    """Regression for a missing barrier between the grid's per-cell sort and the candidate scans (found by the 9000-case fuzz sweep:
    about 1 run in 50 of two particular cases lost a match): the same search repeated many times must equal the oracle every time."""
    from orb_ygz_slam_amd import Extractor, make_camera, EUROC
    for seed in (379, 409):
        rng = np.random.default_rng(1000 + seed)
        w, h = int(rng.integers(200, 1100)), int(rng.integers(160, 800))
        ex = Extractor(300, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=1)
        oex = oracle.Extractor(300, 1.2, 8, 20, 7)
        sf = oex.tables()["scale"]
        base = synth_frame(1100 + seed, w + 16, h + 16)
        a, b = base[8:8 + h, 8:8 + w], base[10:10 + h, 5:5 + w]
        ka, da = ex.extract(a)
        kb, db = ex.extract(b)
        cam = make_camera(w, h)
        prev = np.stack([ka["x"], ka["y"]], -1).astype(np.float32)
        e = oracle.search_for_initialization(ka, da, kb, db, sf, w, h, EUROC, prev, 10, 0.9, True)
        I, z = np.eye(3, dtype=np.float32), np.zeros(3, np.float32)
        e0 = oracle.search_by_projection_last(kb, db, sf, w, h, EUROC, ka, _unit_world(ka, EUROC), da, I, z, I, z, 15.0)
        for _ in range(150):
            g = ex.search_for_initialization(cam, ka, da, kb, db, prev, 10, 0.9, True, scale_factors=sf)
            g0 = ex.search_by_projection_last(cam, kb, db, ka, _unit_world(ka, EUROC), da, I, z, I, z, 15.0, scale_factors=sf)
            assert g[0] == e[0] and (g[1] == e[1]).all()
            assert g0[0] == e0[0] and (g0[1] == e0[1]).all()


def test_large_transfers_bypass_the_page_locked_staging():
    """One-frame entry points pack their host arrays into a page-locked staging area that must not grow without bound: beyond kPackedMax the
    arrays cross one by one from / to the caller's memory (PackedTransfer::direct, csrc/ygzf_ctx.h).  With YGZF_FORCE=packed_max=4096 every matcher /
    BoW / triangulation / aligner call of an ordinary frame takes that path: the suites must hold unchanged."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    from orb_ygz_slam_amd.capi import force_env
    env = dict(os.environ, YGZF_FORCE=force_env(packed_max=4096))
    out = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", "-p", "no:cacheprovider", os.path.join(root, "tests", "test_gpu_match.py"),
                          os.path.join(root, "tests", "test_gpu_frustum.py"), os.path.join(root, "tests", "test_gpu_grid.py"), os.path.join(root, "tests", "test_gpu_align.py"),
                          "-k", "not large_transfers and not in_kernel_reference_patches and not deterministic"],
                         cwd=root, env=env, capture_output=True, text=True, timeout=1200)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
    assert " passed" in out.stdout
