"""Scenarios of the four Tracking-side searches (SearchByProjection(Cur, Last), SearchByProjection(F, MapPoints), SearchForInitialization,
SearchByBoW) shared by tools/make_golden_matcher_ref.py -- which runs them through THE REFERENCE'S OWN src/ORBmatcher.cc
(oracle/_ref/libref_orbmatcher.so) and commits what it returned as tests/golden/matcher_ref.npz -- and by the tests that hold the oracle
(CPU tier) and the device (GPU tier) to those bytes.  Arguments follow oracle_py's signatures; results are reduced to what the reference can
report: a slot matched and then culled by the rotation check reads -1, as a slot never touched (canon)."""
import numpy as np

from orb_ygz_slam_amd.scene import synth_frame

W, H = 752, 480
CAM = dict(fx=458.654, fy=457.296, cx=367.215, cy=248.375)


def canon(m):
    return np.where(m == -2, -1, m)


def inputs(extractor):
    """extractor: an object with extract(img) and tables() (the oracle's or the device's: bit-exact with each other)."""
    base = synth_frame(50, W + 16, H + 16)
    a, b = base[8:8 + H, 8:8 + W], base[10:10 + H, 5:5 + W]
    ka, da = extractor.extract(a)
    kb, db = extractor.extract(b)
    return ka, da, kb, db, extractor.tables()["scale"]


def cases(ka, da, kb, db, sf):
    """-> list of (name, function name, args, kwargs)."""
    f32 = np.float32
    out = []
    rng = np.random.default_rng(3)
    n = len(ka)
    depth = rng.uniform(2.0, 8.0, n).astype(f32)
    world = (np.stack([(ka["x"] - f32(CAM["cx"])) / f32(CAM["fx"]), (ka["y"] - f32(CAM["cy"])) / f32(CAM["fy"]), np.ones(n, f32)], -1).astype(f32) * depth[:, None]).astype(f32)
    ang = f32(np.deg2rad(0.5))
    Rcw = np.array([[np.cos(ang), 0, np.sin(ang)], [0, 1, 0], [-np.sin(ang), 0, np.cos(ang)]], f32)
    tcw = np.array([0.02, -0.01, 0.03], f32)
    I, z = np.eye(3, dtype=f32), np.zeros(3, f32)
    valid = (rng.uniform(size=n) > 0.1).astype(np.uint8)
    outl = (rng.uniform(size=n) > 0.9).astype(np.uint8)
    obs = (rng.uniform(size=n) > 0.3).astype(np.uint8)
    uright = np.where(rng.uniform(size=len(kb)) > 0.5, kb["x"] - 5.0, -1.0).astype(f32)
    owner0 = ((rng.uniform(size=len(kb)) > 0.9) * rng.integers(1, 3, len(kb))).astype(np.uint8)
    for k, cs in enumerate((dict(th=15.0, mono=True, check_level=True, check_ori=True), dict(th=7.0, mono=False, check_level=True, check_ori=True, u_right=uright, mb=0.11, mbf=50.0),
                            dict(th=30.0, mono=True, check_level=False, check_ori=False), dict(th=15.0, mono=False, check_level=True, check_ori=True, tz=0.5, mb=0.11, mbf=50.0))):
        cam = dict(CAM, mb=cs.get("mb", 0.0), mbf=cs.get("mbf", 0.0))
        t = tcw.copy()
        t[2] += cs.get("tz", 0.0)
        kw = dict(mp_valid=valid, outlier=outl, mp_has_obs=obs, u_right=cs.get("u_right"), cur_owner=owner0)
        out.append(("last%d" % k, "search_by_projection_last", (kb, db, sf, W, H, cam, ka, world, da, Rcw, t, I, z, cs["th"], cs["mono"], cs["check_level"], cs["check_ori"]), kw))
    rng = np.random.default_rng(7)
    px = (ka["x"] - 3.0 + rng.normal(0, 1.0, n)).astype(f32)
    py = (ka["y"] + 2.0 + rng.normal(0, 1.0, n)).astype(f32)
    vc = rng.uniform(0.99, 1.0, n).astype(f32)
    vc[::5] = rng.uniform(0.9, 0.998, len(vc[::5]))
    lvl = np.clip(ka["octave"] + rng.integers(-1, 2, n), 0, 7).astype(np.int32)
    tiv = (rng.uniform(size=n) > 0.15).astype(np.uint8)
    bad = (rng.uniform(size=n) > 0.95).astype(np.uint8)
    obs2 = (rng.uniform(size=n) > 0.2).astype(np.uint8)
    pxr = (px - 4.0).astype(f32)
    ur2 = np.where(rng.uniform(size=len(kb)) > 0.5, kb["x"] - 4.0, -1.0).astype(f32)
    own2 = ((rng.uniform(size=len(kb)) > 0.9) * rng.integers(1, 3, len(kb))).astype(np.uint8)
    for k, cs in enumerate((dict(th=1.0, check_level=False, nnratio=0.8), dict(th=3.0, check_level=True, nnratio=0.8),
                            dict(th=5.0, check_level=False, nnratio=0.6, stereo=True), dict(th=8.0, check_level=True, nnratio=0.9, stereo=True))):
        kw = dict(is_bad=bad, mp_has_obs=obs2, owner=own2)
        if cs.get("stereo"):
            kw.update(proj_xr=pxr, u_right=ur2)
        out.append(("mappoints%d" % k, "search_by_projection_mappoints", (kb, db, sf, W, H, CAM, tiv, px, py, vc, lvl, da, cs["th"], cs["check_level"], cs["nnratio"]), kw))
    prev = np.stack([ka["x"], ka["y"]], -1).astype(f32)
    for k, (window, ratio, ori) in enumerate(((100, 0.9, True), (30, 0.9, True), (100, 0.6, False))):
        out.append(("init%d" % k, "search_for_initialization", (ka, da, kb, db, sf, W, H, CAM, prev, window, ratio, ori), {}))
    rng = np.random.default_rng(5)
    kfv = (rng.uniform(size=n) > 0.1).astype(np.uint8)
    for k, (bits, ratio, ori) in enumerate(((4, 0.7, False), (6, 0.75, True), (1, 0.9, True))):
        na, nb = da[:, 0].astype(np.int32) >> (8 - bits), db[:, 0].astype(np.int32) >> (8 - bits)
        ko, fo, ki, fi = [0], [0], [], []
        for node in sorted(set(na.tolist()) & set(nb.tolist())):
            ki.extend(np.nonzero(na == node)[0]); fi.extend(np.nonzero(nb == node)[0])
            ko.append(len(ki)); fo.append(len(fi))
        out.append(("bow%d" % k, "search_by_bow", (np.array(ko, np.int32), np.array(ki, np.int32), np.array(fo, np.int32), np.array(fi, np.int32), kfv, ka, da, kb, db, ratio, ori), {}))
    return out


def reduce(fn, res):
    """What is compared: (nmatches, canonical match array[, owner nonzero / owner][, updated prevMatched])."""
    if fn == "search_for_initialization":
        return {"n": np.int32(res[0]), "m": np.asarray(res[1], np.int32), "p": np.asarray(res[2], np.float32)}
    if fn == "search_by_bow":
        return {"n": np.int32(res[0]), "m": canon(np.asarray(res[1], np.int32))}
    return {"n": np.int32(res[0]), "m": canon(np.asarray(res[1], np.int32)), "o": np.asarray(res[2], np.uint8)}
