"""GPU test of the drop-in boundary: tests/cpp/bin/boundary_frame is the REFERENCE's own src/Frame.cc (real include/Frame.h, its
constructors, ExtractFeatures with its two extraction threads, ComputeStereoMatches, AssignFeaturesToGrid, isInFrustum) linked against the
PRODUCT's class shells (orb_ygz_slam_amd/csrc/host: ORBextractor.h in place of the reference's header, ORBmatcher.cc / SparseImageAlign.cc
defining the members of the reference's own, unchanged class declarations) and libygzf.so -- built by tests/cpp/build_boundary.sh where the
reference checkout exists.  The OpenCV stand-in's compute primitives abort in that binary, so every pyramid, keypoint and descriptor below
came out of the HIP library.  Everything the reference code produced through that path must equal the oracle."""
import os
import subprocess

import numpy as np
import pytest

from orb_ygz_slam_amd.capi import EUROC, KP_DTYPE
from orb_ygz_slam_amd.scene import two_view_scene
from tests.conftest import ROOT

pytestmark = pytest.mark.gpu
EXE = os.path.join(ROOT, "tests", "cpp", "bin", "boundary_frame")


def test_reference_frame_over_product_shells(oracle, tmp_path):
    if os.path.isdir("/root/reference"):
        subprocess.check_call([os.path.join(ROOT, "tests", "cpp", "build_boundary.sh")])
    assert os.path.exists(EXE), "tests/cpp/bin/boundary_frame missing: run tests/cpp/build_boundary.sh where /root/reference exists"
    from orb_ygz_slam_amd import load_library
    load_library()
    w, h, NF, L, depth = 752, 480, 1000, 8, np.float32(4.0)
    imgL, imgN, (R, t), _ = two_view_scene(17, w, h, EUROC, Z=float(depth))
    imgR = np.zeros_like(imgL)                      # right eye: bands of the left image shifted by known disparities
    for bnd, dsp in enumerate((5, 11, 17, 24, 31, 8)):
        imgR[bnd * 80:(bnd + 1) * 80, :w - dsp] = imgL[bnd * 80:(bnd + 1) * 80, dsp:]
    np.array([w, h, NF, L], np.int32).tofile(tmp_path / "size.i32")
    imgL.tofile(tmp_path / "left.u8"); imgR.tofile(tmp_path / "right.u8"); imgN.tofile(tmp_path / "next.u8")
    voc = oracle.make_vocabulary(23, 10, 5)          # ORBvoc's shape (k = 10) one level shallower: 111 111 nodes, levelsup = 4 -> FeatureVector keys at level 1
    oracle.write_vocabulary_text(voc, str(tmp_path / "voc.txt"))
    # the settings file Tracking::Tracking reads (src/Tracking.cc:83-213; Examples/Monocular/EuRoC.yaml's keys)
    (tmp_path / "settings.yaml").write_text("%%YAML:1.0\nCamera.fx: %r\nCamera.fy: %r\nCamera.cx: %r\nCamera.cy: %r\nCamera.k1: 0.0\nCamera.k2: 0.0\n"
                                            "Camera.p1: 0.0\nCamera.p2: 0.0\nCamera.bf: 47.9\nCamera.fps: 20.0\nCamera.RGB: 1\nORBextractor.nFeatures: %d\n"
                                            "ORBextractor.scaleFactor: 1.2\nORBextractor.nLevels: %d\nORBextractor.iniThFAST: 20\nORBextractor.minThFAST: 7\n"
                                            % (EUROC["fx"], EUROC["fy"], EUROC["cx"], EUROC["cy"], NF, L))
    out = subprocess.run([EXE, str(tmp_path)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "boundary ok" in out.stdout
    lat = [l for l in out.stdout.splitlines() if l.startswith("latency ")]
    assert len(lat) == 5
    tri = [l for l in out.stdout.splitlines() if l.startswith("info search_for_triangulation")]
    assert len(tri) == 2 and int(tri[0].split()[5]) >= 50, tri      # device pairs == the reference's own body (the driver exits 7 otherwise)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "boundary_latency.txt"), "w") as fh:
        fh.write("# tests/cpp/boundary_frame: the reference's own loop bodies (per-call members) against the batch bindings of TrackingBatched.cc, 752x480, ~1000 local points\n")
        fh.write("\n".join(lat) + "\n")
    for l in lat:
        tok = l.split()
        if tok[1].startswith("search_local_points_direct"):
            assert float(tok[5]) < float(tok[3]), l          # one launch chain per frame beats one per candidate
    rd = lambda name, dt: np.fromfile(tmp_path / name, dt)
    f = np.float32
    oex = oracle.Extractor(NF, 1.2, L, 20, 7)
    tab = oex.tables()
    # ---- stereo frame: reference constructor + ExtractFeatures over the shells ----
    assert (rd("s_scale.bin", np.float32) == tab["scale"]).all()
    for l, lvl in enumerate(oex.pyramid(imgL)):
        assert (rd("s_pyr%d.bin" % l, np.uint8).reshape(lvl.shape) == lvl).all(), "Frame::mvImagePyramid[%d]" % l
    kl, dl = oex.extract(imgL)
    kr, dr = oex.extract(imgR)
    assert (rd("s_keys.bin", KP_DTYPE) == kl).all() and (rd("s_desc.bin", np.uint8).reshape(-1, 32) == dl).all()
    assert (rd("s_keysr.bin", KP_DTYPE) == kr).all() and (rd("s_descr.bin", np.uint8).reshape(-1, 32) == dr).all()
    mbf = f(47.9)
    mb = mbf / f(EUROC["fx"])
    our, odp = oex.compute_stereo_matches(imgL, imgR, kl, dl, kr, dr, float(mb), float(mbf))
    sur, sdp = rd("s_uright.bin", np.float32), rd("s_depth.bin", np.float32)
    # Frame::ComputeStereoMatches as ExtractFeatures called it = the product's strong member (FrameStereo.cc -> device); the reference's own CPU body on the
    # same Frame (the binary has already demanded equality); the extractor shell's entry point called directly
    assert (sur.view(np.uint32) == our.view(np.uint32)).all() and (sdp.view(np.uint32) == odp.view(np.uint32)).all()
    assert (rd("s_uright_ref.bin", np.float32).view(np.uint32) == our.view(np.uint32)).all()                               # reference CPU code on HIP pyramids
    assert (rd("s_depth_ref.bin", np.float32).view(np.uint32) == odp.view(np.uint32)).all()
    assert (rd("s_uright_dev.bin", np.float32).view(np.uint32) == our.view(np.uint32)).all()                               # device ComputeStereoMatches
    assert (rd("s_depth_dev.bin", np.float32).view(np.uint32) == odp.view(np.uint32)).all()
    assert (our >= 0).sum() > 100
    # Frame::ComputeBoW -> DeviceORBVocabulary::transform (the binary has already compared it with the reference's CPU DBoW2 class)
    leaf, nid = oracle.bow_descend(voc, dl, 4)
    o_ids, o_vals, o_fv = oracle.bow_vectors(voc, leaf, nid)
    bow = rd("s_bow.bin", np.float64).reshape(-1, 2)
    assert (bow[:, 0].astype(np.int32) == o_ids).all() and (bow[:, 1].view(np.uint64) == o_vals.view(np.uint64)).all() and len(o_ids) > 100
    fvv, pos = rd("s_featvec.bin", np.int32), 0
    for key in sorted(o_fv):
        assert fvv[pos] == key and fvv[pos + 1] == len(o_fv[key]) and (fvv[pos + 2:pos + 2 + fvv[pos + 1]] == o_fv[key]).all()
        pos += 2 + fvv[pos + 1]
    assert pos == len(fvv)
    g = rd("s_grid.bin", np.int32)
    pos = 0
    for k in range(12):
        exp = oracle.features_in_area(kl, tab["scale"], w, h, 60.0 + 55.0 * k, 40.0 + 33.0 * k, 25.0, -1 if k % 3 else 0, -1 if k % 3 else 2)
        n = int(g[pos]); pos += 1
        assert n == len(exp) and (g[pos:pos + n] == exp).all()
        pos += n
    # ---- monocular pair ----
    kn, dn = oex.extract(imgN)
    assert (rd("m_keys_last.bin", KP_DTYPE) == kl).all() and (rd("m_keys_cur.bin", KP_DTYPE) == kn).all()
    assert (rd("m_desc_cur.bin", np.uint8).reshape(-1, 32) == dn).all()
    world = np.stack([(kl["x"] - f(EUROC["cx"])) / f(EUROC["fx"]) * depth, (kl["y"] - f(EUROC["cy"])) / f(EUROC["fy"]) * depth,
                      np.full(len(kl), depth, np.float32)], -1).astype(np.float32)
    ident = np.array([0, 0, 0, 1, 0, 0, 0], np.float32)
    o_ret, o_T, _, o_H = oracle.sparse_img_align(kl, world, ident, oex.pyramid(imgL), ident, oex.pyramid(imgN), tab["inv_scale"], EUROC, L - 1, 1)
    t7 = rd("m_tcr.bin", np.float32)
    assert int(t7[7]) == o_ret and o_ret > 100
    assert np.abs(t7[:7] - o_T).max() <= 1e-5                      # north_star tolerance on the SE3
    assert np.abs(t7[4:7] - t).max() < 5e-3
    fisher = rd("m_fisher.bin", np.float32)
    o_H = np.asarray(o_H, np.float32).reshape(-1)
    assert np.allclose(fisher, o_H / f(5e-4 * 255 * 255), rtol=2e-3, atol=1e-3 * np.abs(o_H).max() / (5e-4 * 255 * 255))
    pose = rd("m_pose.bin", np.float32)
    Rcw, tcw = pose[:9].reshape(3, 3), pose[9:]
    I, z = np.eye(3, dtype=np.float32), np.zeros(3, np.float32)
    e_n, e_m, _ = oracle.search_by_projection_last(kn, dn, tab["scale"], w, h, EUROC, kl, world, dl, Rcw, tcw, I, z, 15.0)
    assert int(rd("m_nmatch.bin", np.int32)[0]) == e_n and e_n > 100
    assert (rd("m_match.bin", np.int32) == np.where(e_m >= 0, e_m, -1)).all()
    # FindDirectProjection through the reference's class declaration, one candidate per call
    dr = rd("m_direct.bin", np.float32).reshape(-1, 4)
    nd = len(dr)
    i5, i3 = np.arange(nd) % 5, np.arange(nd) % 3
    px0 = np.stack([kl["x"][:nd] + (i5 - 2).astype(np.float32) * f(0.75), kl["y"][:nd] + (i3 - 1).astype(np.float32) * f(0.5)], -1).astype(np.float32)
    opx, osl, ook, _ = oex.find_direct_projection_batch([imgL], imgN, rd("m_pose7.bin", np.float32), EUROC, np.zeros(nd, np.int32), np.tile(ident, (nd, 1)),
                                                        kl[:nd], world[:nd], px0)
    assert nd == 120 and (dr[:, 2].astype(np.int32) == osl).all() and (dr[:, 3].astype(np.uint8) == ook).all() and ook.sum() > 60
    assert np.array_equal(dr[:, :2].copy().view(np.uint32), opx.view(np.uint32))
    # the link: hot-path members are the product's (strong), the LocalMapping / LoopClosing members still the reference's own (weak)
    sym = open(EXE + ".symbols").read().splitlines()
    strong = [l for l in sym if l.startswith("T ")]
    weak = [l for l in sym if l.startswith("W ")]
    assert all(any(n in l for l in strong) for n in ("FindDirectProjection", "SearchForInitialization", "SearchForTriangulation", "DescriptorDistance",
                                                       "SearchByBoW(ygz::KeyFrame*, ygz::Frame&",
                                                       "SparseImgAlign::run", "ORBextractor::operator()(ygz::Frame*", "Frame::ComputeStereoMatches"))
    assert sum("ORBmatcher::" in l for l in strong) == 8
    # the reference's own src/Tracking.cc is in the binary, unchanged, and its hot-path callers are there to call the product's definitions above
    for member in ("TrackWithSparseAlignment", "TrackWithMotionModel", "SearchLocalPoints()", "MonocularInitialization", "Relocalization", "SearchLocalPointsDirect",
                   "TrackReferenceKeyFrame"):
        assert any(l.startswith("T ygz::Tracking::" + member) for l in strong), member
    outside = open(EXE + ".outside").read().split()
    assert 30 < len(outside) < 120 and all(x.startswith(("_ZN3ygz", "_ZNK3ygz")) for x in outside)       # what aborts when reached: members of classes outside the hot path only
    assert not any(("ORBmatcher" in x) or ("SparseImgAlign" in x) or ("ORBextractor" in x) or ("3ygz5Frame" in x) for x in outside)
    assert len(weak) == 5 and all(any(n in l for l in weak) for n in ("Fuse", "SearchBySim3", "SearchByBoW(ygz::KeyFrame*, ygz::KeyFrame*"))
    # SearchLocalPoints: the reference's isInFrustum (CPU) marks, the shell searches
    fr = rd("m_frustum.bin", np.float32).reshape(-1, 5)
    Ow = -(Rcw.T @ tcw)
    mf_max = (depth * tab["scale"][kl["octave"]]).astype(np.float32)
    iv, px, py, _, lv, vc = oracle.is_in_frustum(kn, dn, tab["scale"], w, h, EUROC, world, np.tile(np.array([0, 0, 1], np.float32), (len(kl), 1)),
                                                 (f(1.2) * mf_max).astype(np.float32), (f(0.8) * mf_max / tab["scale"][L - 1]).astype(np.float32), mf_max,
                                                 Rcw, tcw, Ow.astype(np.float32), np.log(f(1.2)), 0.5)
    assert (fr[:, 0].astype(np.uint8) == iv).all() and iv.sum() > 100
    sel = iv.astype(bool)
    assert (fr[sel, 1] == px[sel]).all() and (fr[sel, 2] == py[sel]).all() and (fr[sel, 3].astype(np.int32) == lv[sel]).all()
    e_n2, e_m2, _ = oracle.search_by_projection_mappoints(kn, dn, tab["scale"], w, h, EUROC, iv, fr[:, 1].copy(), fr[:, 2].copy(), fr[:, 4].copy(),
                                                          fr[:, 3].astype(np.int32), dl, 3.0, False, 0.8)
    assert int(rd("m_nmatch2.bin", np.int32)[0]) == e_n2 and e_n2 > 50
    assert (rd("m_match2.bin", np.int32) == np.where(e_m2 >= 0, e_m2, -1)).all()
    # ---- the reference's src/Tracking.cc drove the same two calls itself ----
    # Tracking::TrackWithSparseAlignment: UpdateLastFrame, motion-model pose (identity velocity), mpAlign->run, SetPose(TCR * last): same inputs as
    # the direct call above, so the same SE3 bit for bit, and within 1e-5 of the oracle
    p7 = rd("t_pose7.bin", np.float32)
    assert p7[7] == 1.0 and np.array_equal(p7[:7], t7[:7]) and np.abs(p7[:7] - o_T).max() <= 1e-5
    # Tracking::SearchLocalPoints: IncreaseVisible bookkeeping, the reference's Frame::isInFrustum per local point, then
    # ORBmatcher(0.8).SearchByProjection(mCurrentFrame, mvpLocalMapPoints, th, false) with th = 5 right after a relocalisation id, 1 otherwise (:1578-1590)
    fid, visible, tn = rd("t_info.bin", np.int32)
    assert tn == len(kn) and (rd("t_keys.bin", KP_DTYPE) == kn).all() and (rd("t_desc.bin", np.uint8).reshape(-1, 32) == dn).all()
    th = 5.0 if fid < 2 else 1.0
    e_n3, e_m3, _ = oracle.search_by_projection_mappoints(kn, dn, tab["scale"], w, h, EUROC, iv, fr[:, 1].copy(), fr[:, 2].copy(), fr[:, 4].copy(),
                                                          fr[:, 3].astype(np.int32), dl, th, False, 0.8)
    a3 = rd("t_match.bin", np.int32)
    assert (a3 == np.where(e_m3 >= 0, e_m3, -1)).all() and (a3 >= 0).sum() == e_n3 and e_n3 > 30
    assert visible == int(iv.sum())                                                      # every point the frustum test accepted was counted visible once
    # ---- Tracking::SearchLocalPointsDirect, batch binding (TrackingBatched.cc; the binary has already demanded that it equals the reference's own body
    # calling FindDirectProjection once per candidate): against the oracle -- frustum of every point, one candidate per point (its observation in
    # KF1 = the last frame), keys appended for the successes at least 20 px inside the image
    x7 = rd("x_pose7.bin", np.float32)
    assert np.array_equal(x7, p7[:7])
    xp = rd("x_pose.bin", np.float32)                       # mRcw / mtcw as the Frame holds them
    Rx, tx = xp[:9].reshape(3, 3), xp[9:]
    Owx = -(Rx.T @ tx)
    ivx, pxx, pyx, _, lvx, _ = oracle.is_in_frustum(kn, dn, tab["scale"], w, h, EUROC, world, np.tile(np.array([0, 0, 1], np.float32), (len(kl), 1)),
                                                    (f(1.2) * mf_max).astype(np.float32), (f(0.8) * mf_max / tab["scale"][L - 1]).astype(np.float32), mf_max,
                                                    Rx, tx, Owx.astype(np.float32), np.log(f(1.2)), 0.5)
    sel = np.nonzero(ivx)[0]
    px0 = np.stack([pxx[sel], pyx[sel]], -1).astype(np.float32)
    opx, _, ook, _ = oex.find_direct_projection_batch([imgL], imgN, x7, EUROC, np.zeros(len(sel), np.int32), np.tile(ident, (len(sel), 1)), kl[sel], world[sel], px0)
    inside = (opx[:, 0] >= 20) & (opx[:, 1] >= 20) & (opx[:, 0] < w - 20) & (opx[:, 1] < h - 20)
    good = ook.astype(bool) & inside
    k0 = rd("x0_keys.bin", KP_DTYPE)
    assert len(k0) == good.sum() > 100
    assert np.array_equal(np.stack([k0["x"], k0["y"]], -1).view(np.uint32), opx[good].view(np.uint32)) and (k0["size"] == 7).all() and (k0["octave"] == 0).all()
    assert np.array_equal(rd("x0_mp.bin", np.int32), sel[good]) and (rd("x0_from.bin", np.int32) == 3).all()
    assert np.array_equal(rd("x0_cache.bin", np.int32), sel[good])                    # every matched point went into the cache (a std::set: ascending addresses)
    # second frame on that cache (the function's first half): coverage grid of 5-px cells, carried from point to point in cache order
    gcols, taken, exp_mp, exp_xy, exp_cache = w // 5, set(), [], [], []
    res = {int(i): (bool(g), opx[j]) for j, (i, g) in enumerate(zip(sel, good))}
    for i in sel[good]:
        i = int(i)                                                                    # (every cached point is in view again: same pose)
        cell = int(pyx[i] / f(5)) * gcols + int(pxx[i] / f(5))
        if cell in taken:
            exp_cache.append(i)
            continue
        g, q = res[i]
        assert g
        exp_mp.append(i); exp_xy.append(q); exp_cache.append(i)
        taken.add(int(q[1] / f(5)) * gcols + int(q[0] / f(5)))
    k1 = rd("x1_keys.bin", KP_DTYPE)
    assert len(exp_mp) > 10 and np.array_equal(rd("x1_mp.bin", np.int32), np.array(exp_mp, np.int32))
    assert np.array_equal(np.stack([k1["x"], k1["y"]], -1).view(np.uint32), np.array(exp_xy, np.float32).view(np.uint32))
    assert np.array_equal(rd("x1_cache.bin", np.int32), np.array(exp_cache, np.int32)) and len(exp_mp) < len(exp_cache)   # some points were skipped by the grid
    # direct-tracked frame: Frame::ExtractORB took the DSO_KEYPOINT branch
    ko, do, _ = oracle.Extractor(NF, 1.2, L, 20, 7).extract_dso(imgN, existing=kn[:120])
    kd = rd("d_keys.bin", KP_DTYPE)
    assert len(kd) == len(ko) > 120 and (kd == ko).all()
    assert (rd("d_desc.bin", np.uint8).reshape(-1, 32) == do).all()


def test_reference_mappoint_over_product_batch(oracle):
    """tests/cpp/bin/libboundary_mappoint.so: the reference's own src/MapPoint.cc (REAL include/MapPoint.h: private members, mutexes) under the product's
    MapPointBatch.cc.  On the same MapPoint objects: the reference's own CPU body of ComputeDistinctiveDescriptors, the product's strong member called
    per point (one device call each) and ygz::ComputeDistinctiveDescriptorsBatch (ONE device call) must pick the same observation -- and the oracle's.
    Includes points without observations, with one, with bad KeyFrames among them, with identical descriptors (first minimum wins) and with more
    than 256 observations (the histogram kernel)."""
    import ctypes as C
    if os.path.isdir("/root/reference"):
        subprocess.check_call([os.path.join(ROOT, "tests", "cpp", "build_boundary.sh")])
    lib = os.path.join(ROOT, "tests", "cpp", "bin", "libboundary_mappoint.so")
    assert os.path.exists(lib), "tests/cpp/bin/libboundary_mappoint.so missing: run tests/cpp/build_boundary.sh where /root/reference exists"
    from orb_ygz_slam_amd import load_library
    load_library()
    B = C.CDLL(lib)
    rng = np.random.default_rng(11)
    counts = [0, 1, 2, 3, 7, 40, 64, 65, 200, 256, 257, 300, 700] + [int(c) for c in rng.integers(1, 30, 60)]
    off = np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)
    total = int(off[-1])
    centres = rng.integers(0, 256, (len(counts), 32), dtype=np.uint8)
    desc = np.zeros((total, 32), np.uint8)
    for p, n in enumerate(counts):
        noise = (rng.random((n, 256)) < 0.12)
        desc[off[p]:off[p + 1]] = centres[p] ^ np.packbits(noise, axis=1)
    desc[off[6]:off[6] + 5] = desc[off[6]]                      # ties: several identical rows
    bad = (rng.random(total) < 0.1).astype(np.uint8)
    bad[off[1]] = 0
    bad[off[8]:off[9]] = 1                                      # a point whose KeyFrames are all bad: the descriptor stays untouched
    out = np.full((len(counts), 3), -9, np.int32)
    fails = B.bm_distinctive(len(counts), off.ctypes.data_as(C.c_void_p), desc.ctypes.data_as(C.c_void_p), bad.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p))
    assert fails == 0
    assert (out[:, 0] == out[:, 1]).all() and (out[:, 0] == out[:, 2]).all(), out[(out[:, 0] != out[:, 1]) | (out[:, 0] != out[:, 2])]
    assert out[0, 0] == -1 and out[8, 0] == -1 and out[1, 0] == 0
    # and the oracle, on the rows of the KeyFrames that are not bad
    keep = bad == 0
    goff = np.concatenate([[0], np.cumsum([int(keep[off[p]:off[p + 1]].sum()) for p in range(len(counts))])]).astype(np.int32)
    ob = oracle.distinctive_descriptors(goff, desc[keep])
    for p in range(len(counts)):
        if goff[p + 1] == goff[p]:
            assert out[p, 0] == -1
            continue
        win = desc[keep][goff[p] + ob[p]]
        first = next(i for i in range(off[p], off[p + 1]) if (desc[i] == win).all())
        assert out[p, 0] == first - off[p], p
    sym = open(lib[:-3] + ".symbols").read()
    assert "T ygz::MapPoint::ComputeDistinctiveDescriptors()" in sym and "T ygz::ComputeDistinctiveDescriptorsBatch" in sym and "ygz_ref_MapPoint_ComputeDistinctiveDescriptors" in sym


def test_shell_failures_are_observable():
    """The reference's signatures have no error channel, so a shell whose device call fails can only return "nothing": ygzf_host::failure_count() /
    last_failure() / set_failure_callback make that visible.  Provoked here with a device that does not exist."""
    import ctypes as C
    lib = os.path.join(ROOT, "tests", "cpp", "bin", "libboundary_mappoint.so")
    assert os.path.exists(lib)
    from orb_ygz_slam_amd import load_library
    load_library()
    B = C.CDLL(lib)
    msg = C.create_string_buffer(512)
    ncb = C.c_int(0)
    n = B.bm_provoke_failure(msg, 512, C.byref(ncb))
    assert n >= 2 and ncb.value == n                  # the batch call and the member's own call both failed, both were counted, the callback saw both
    text = msg.value.decode()
    assert "ygz::" in text or "libygzf" in text
    assert "device" in text.lower()
