#!/usr/bin/env python3
"""Golden vectors of the ORACLE for the matcher / aligner / DSO / stereo / direct-projection / frustum / distinctive paths on seeded
inputs -> tests/golden/paths_golden.npz (sha256 digests + a few raw vectors).  Like extract_golden.npz these pin OUR oracle's
definition (the reference ships no vectors: parity unpinned); tests/test_oracle_paths_golden.py replays them on the CPU tier and
tests/test_gpu_golden.py holds the HIP path to the same committed bytes."""
import hashlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def sha(*arrs):
    h = hashlib.sha256()
    for a in arrs:
        h.update(np.ascontiguousarray(a).tobytes())
    return np.frombuffer(h.digest(), np.uint8)


def cases(O):
    """Yields (name, dict of result arrays).  Shared by the generator and the replaying tests (O = oracle module or a GPU adapter)."""
    from orb_ygz_slam_amd.capi import EUROC
    from orb_ygz_slam_amd.scene import two_view_scene, stereo_scene, rotvec_to_quat
    from orb_ygz_slam_amd.synth import synth_frame
    w, h = 752, 480
    ex = O.Extractor(1000, 1.2, 8, 20, 7)
    sf, inv = ex.tables()["scale"], ex.tables()["inv_scale"]
    base = synth_frame(77, w + 16, h + 16)
    a, b = base[8:8 + h, 8:8 + w], base[10:10 + h, 5:5 + w]
    ka, da = ex.extract(a)
    kb, db = ex.extract(b)
    rng = np.random.default_rng(77)
    n = len(ka)
    f32 = np.float32
    depth = rng.uniform(2.0, 8.0, n).astype(f32)
    world = np.stack([(ka["x"] - f32(EUROC["cx"])) / f32(EUROC["fx"]) * depth, (ka["y"] - f32(EUROC["cy"])) / f32(EUROC["fy"]) * depth, depth], -1).astype(f32)
    ang = f32(np.deg2rad(0.5))
    Rcw = np.array([[np.cos(ang), 0, np.sin(ang)], [0, 1, 0], [-np.sin(ang), 0, np.cos(ang)]], f32)
    tcw = np.array([0.02, -0.01, 0.03], f32)
    I, z = np.eye(3, dtype=f32), np.zeros(3, f32)
    yield "match_last", dict(zip("nmo", O.search_by_projection_last(kb, db, sf, w, h, EUROC, ka, world, da, Rcw, tcw, I, z, 15.0)))
    tiv = (rng.uniform(size=n) > 0.2).astype(np.uint8)
    px, py = (ka["x"] + f32(0.5)).astype(f32), (ka["y"] - f32(0.25)).astype(f32)
    vc = np.where(np.arange(n) % 3, f32(0.9995), f32(0.99)).astype(f32)
    yield "match_mappoints", dict(zip("nmo", O.search_by_projection_mappoints(kb, db, sf, w, h, EUROC, tiv, px, py, vc, ka["octave"], da, 3.0, True, 0.8)))
    prev = np.stack([ka["x"], ka["y"]], -1).astype(f32)
    yield "match_init", dict(zip("nmp", O.search_for_initialization(ka, da, kb, db, sf, w, h, EUROC, prev, 100, 0.9, True)))
    na, nb = da[:, 0].astype(np.int32) >> 3, db[:, 0].astype(np.int32) >> 3
    ko, fo, ki, fi = [0], [0], [], []
    for nd in sorted(set(na.tolist()) & set(nb.tolist())):
        ki.extend(np.nonzero(na == nd)[0]); fi.extend(np.nonzero(nb == nd)[0])
        ko.append(len(ki)); fo.append(len(fi))
    yield "match_bow", dict(zip("nm", O.search_by_bow(ko, ki, fo, fi, tiv, ka, da, kb, db, 0.7, True)))
    normal = (world / np.linalg.norm(world, axis=1, keepdims=True)).astype(f32)
    dist = np.linalg.norm(world, axis=1).astype(f32)
    mf = (dist * sf[ka["octave"]]).astype(f32)
    Ow = (-Rcw.T @ tcw).astype(f32)
    fr = O.is_in_frustum(kb, db, sf, w, h, EUROC, world, normal, (f32(1.2) * mf).astype(f32), (f32(0.8) * mf / sf[7]).astype(f32), mf, Rcw, tcw, Ow,
                         np.log(f32(1.2)), 0.5)
    iv = fr[0].astype(bool)
    yield "frustum", {"iv": fr[0], "px": fr[1][iv], "py": fr[2][iv], "lv": fr[4][iv], "vc": fr[5][iv]}
    off = np.concatenate([[0], np.cumsum(rng.integers(1, 30, 100))]).astype(np.int32)
    yield "distinctive", {"best": O.distinctive_descriptors(off, da[np.arange(off[-1]) % n])}
    # DSO on frame a with some of its own keys as existing ones
    kd, dd, g = ex.extract_dso(a)
    lx, ly = ka["x"] * inv[ka["octave"]], ka["y"] * inv[ka["octave"]]
    sizes = np.array([ex.level_size(w, h, l) for l in range(8)])
    keep = (lx >= 16) & (ly >= 16) & (lx < sizes[ka["octave"], 0] - 16) & (ly < sizes[ka["octave"], 1] - 16)
    kd2, dd2, g2 = ex.extract_dso(a, existing=ka[keep][::5], grid_size=g)
    yield "dso", {"k": kd, "d": dd, "g": np.array([g, g2]), "k2": kd2, "d2": dd2}
    # stereo
    left, right, _, _ = stereo_scene(78, w, h)
    kl, dl = ex.extract(left)
    kr, dr = ex.extract(right)
    ur, dp = ex.compute_stereo_matches(left, right, kl, dl, kr, dr, 0.11, 47.9)
    yield "stereo", {"ur": ur, "dp": dp}
    # aligner + direct projection on a rendered plane
    rv, tr = (0.004, -0.006, 0.003), (0.03, -0.02, 0.015)
    A, B, (R, t), bp = two_view_scene(79, w, h, EUROC, rotvec=rv, trans=tr)
    k, _ = ex.extract(A)
    wp = bp(k["x"], k["y"])
    ident = np.array([0, 0, 0, 1, 0, 0, 0], f32)
    pa, pb = ex.pyramid(A), ex.pyramid(B)
    r = O.sparse_img_align(k, wp, ident, pa, ident, pb, inv, EUROC, 7, 1)
    yield "align", {"ret": np.array([r[0]]), "T": np.asarray(r[1], f32)}
    q = rotvec_to_quat(rv)
    T7 = np.array([q[0], q[1], q[2], q[3], *tr], f32)
    px0 = (np.stack([k["x"], k["y"]], -1) + rng.uniform(-3, 3, (len(k), 2))).astype(f32)
    dpj = ex.find_direct_projection_batch([A], B, T7, EUROC, np.zeros(len(k), np.int32), np.tile(ident, (len(k), 1)), k, wp, px0)
    yield "direct", {"px": dpj[0], "sl": dpj[1], "ok": dpj[2], "patch": dpj[3]}


def main():
    from oracle import oracle_py as O
    out = {}
    for name, res in cases(O):
        for key, arr in res.items():
            arr = np.asarray(arr)
            out["%s_%s_sha" % (name, key)] = sha(arr)
            if arr.size <= 16:
                out["%s_%s" % (name, key)] = arr
        print(name, {k: np.asarray(v).shape for k, v in res.items()})
    np.savez_compressed(os.path.join(ROOT, "tests/golden/paths_golden.npz"), **out)


if __name__ == "__main__":
    main()
