"""GPU test of the host-side mirror of the reference interface: the C++ class shells ygz::ORBextractor / ygz::ORBmatcher /
ygz::SparseImgAlign (orb_ygz_slam_amd/csrc/host, reference signatures) are driven the way Tracking.cc drives them
(tests/cpp/test_shells.cc) and their outputs compared with the oracle."""
import os
import subprocess

import numpy as np
import pytest

from orb_ygz_slam_amd.capi import EUROC, KP_DTYPE
from orb_ygz_slam_amd.scene import two_view_scene
from tests.conftest import ROOT

pytestmark = pytest.mark.gpu


def _build(tmp):
    host = os.path.join(ROOT, "orb_ygz_slam_amd", "csrc", "host")
    lib = os.path.join(ROOT, "orb_ygz_slam_amd", "lib")
    exe = os.path.join(tmp, "test_shells")
    srcs = [os.path.join(ROOT, "tests", "cpp", "test_shells.cc")] + [os.path.join(host, f) for f in
                                                                     ("ORBextractor.cc", "ORBmatcher.cc", "SparseImageAlign.cc", "ygzf_pool.cc")]
    # stand-alone build of the shells: host/ (ORBextractor.h, ygz_compat.h) + host/standalone/ (ORBmatcher.h, SparseImageAlign.h)
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-Wall", "-pthread", "-I", host, "-I", os.path.join(host, "standalone")] + srcs +
                          ["-L", lib, "-lygzf", "-Wl,-rpath," + lib, "-o", exe])
    return exe


def _R_from_q(q):  # Eigen toRotationMatrix, float32, same operation order as ygz_compat::se3_to_Rt
    f = np.float32
    tx, ty, tz = f(2) * q[0], f(2) * q[1], f(2) * q[2]
    twx, twy, twz = tx * q[3], ty * q[3], tz * q[3]
    txx, txy, txz = tx * q[0], ty * q[0], tz * q[0]
    tyy, tyz, tzz = ty * q[1], tz * q[1], tz * q[2]
    return np.array([[f(1) - (tyy + tzz), txy - twz, txz + twy], [txy + twz, f(1) - (txx + tzz), tyz - twx],
                     [txz - twy, tyz + twx, f(1) - (txx + tyy)]], np.float32)


def test_class_shells_end_to_end(oracle, tmp_path):
    from orb_ygz_slam_amd import load_library
    load_library()
    exe = _build(str(tmp_path))
    w, h, depth = 752, 480, np.float32(4.0)
    imgA, imgB, (R, t), _ = two_view_scene(9, w, h, EUROC, Z=float(depth))
    imgA.tofile(tmp_path / "a.u8")
    imgB.tofile(tmp_path / "b.u8")
    np.array([depth], np.float32).tofile(tmp_path / "depth.f32")
    imgR = np.zeros_like(imgA)                      # right eye of A: bands of A shifted by known disparities
    for bnd, dsp in enumerate((4, 9, 15, 22, 30, 6)):
        imgR[bnd * 80:(bnd + 1) * 80, :w - dsp] = imgA[bnd * 80:(bnd + 1) * 80, dsp:]
    imgR.tofile(tmp_path / "r.u8")
    # two-view geometry for SearchForTriangulation: KF1 = A at the origin, KF2 = B at (R, t); F12 as LocalMapping::ComputeF12 builds it
    K = np.array([[EUROC["fx"], 0, EUROC["cx"]], [0, EUROC["fy"], EUROC["cy"]], [0, 0, 1]], np.float64)
    R64, t64 = np.asarray(R, np.float64), np.asarray(t, np.float64).reshape(3)
    t12 = -R64.T @ t64
    F12 = (np.linalg.inv(K).T @ np.array([[0, -t12[2], t12[1]], [t12[2], 0, -t12[0]], [-t12[1], t12[0], 0]]) @ R64.T @ np.linalg.inv(K)).astype(np.float32)
    R2w, t2w, Cw1 = R64.astype(np.float32), t64.astype(np.float32), np.zeros(3, np.float32)
    np.concatenate([F12.ravel(), R2w.ravel(), t2w, Cw1]).astype(np.float32).tofile(tmp_path / "tri.f32")
    out = subprocess.run([exe, str(tmp_path)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "shells ok" in out.stdout
    oex = oracle.Extractor(600, 1.2, 8, 20, 7)
    tb = np.fromfile(tmp_path / "tables.bin", np.float32).reshape(4, 8)      # getters read right after construction, before any image
    ot = oex.tables()
    for row, key in enumerate(("scale", "inv_scale", "sigma2", "inv_sigma2")):
        assert (tb[row] == ot[key]).all(), key
    res = {}
    for name, img in (("a", imgA), ("b", imgB)):
        k = np.fromfile(tmp_path / (name + "_kps.bin"), KP_DTYPE)
        d = np.fromfile(tmp_path / (name + "_desc.bin"), np.uint8).reshape(-1, 32)
        ok, od = oex.extract(img)
        assert len(k) == len(ok) and (k == ok).all() and (d == od).all()
        res[name] = (k, d)
    ka, da = res["a"]
    kb, db = res["b"]
    f = np.float32
    world = np.stack([(ka["x"] - f(EUROC["cx"])) / f(EUROC["fx"]) * depth, (ka["y"] - f(EUROC["cy"])) / f(EUROC["fy"]) * depth,
                      np.full(len(ka), depth, np.float32)], -1).astype(np.float32)
    ident = np.array([0, 0, 0, 1, 0, 0, 0], np.float32)
    pyrA, pyrB = oex.pyramid(imgA), oex.pyramid(imgB)
    o_ret, o_T, _, _ = oracle.sparse_img_align(ka, world, ident, pyrA, ident, pyrB, oex.tables()["inv_scale"], EUROC, 7, 1)
    t7 = np.fromfile(tmp_path / "tcr.bin", np.float32)
    assert int(t7[7]) == o_ret and o_ret > 100
    assert np.abs(t7[:7] - o_T).max() <= 1e-5
    assert np.abs(t7[4:7] - t).max() < 5e-3
    # image cache: a deep copy of B (other buffers, same id) gives B's answer; B's id + B's buffers with A's pixels give the identity, not a stale hit
    t_copy = np.fromfile(tmp_path / "tcr_copy.bin", np.float32)
    assert np.array_equal(t_copy, t7)
    assert np.array_equal(np.fromfile(tmp_path / "tcr_resident.bin", np.float32), t7)      # slot filled device to device from the extractor's context
    t_reuse = np.fromfile(tmp_path / "tcr_reuse.bin", np.float32)
    r_ret, r_T, _, _ = oracle.sparse_img_align(ka, world, ident, pyrA, ident, pyrA, oex.tables()["inv_scale"], EUROC, 7, 1)
    assert int(t_reuse[7]) == r_ret and np.abs(t_reuse[:7] - r_T).max() <= 1e-5 and np.abs(t_reuse[4:7]).max() < 1e-3
    # FindDirectProjection as a member, one candidate per call (KeyFrame = A at identity, current frame = B at TCR)
    dr = np.fromfile(tmp_path / "direct.bin", np.float32).reshape(-1, 4)
    nd = len(dr)
    i5, i3 = np.arange(nd) % 5, np.arange(nd) % 3
    px0 = np.stack([ka["x"][:nd] + (i5 - 2).astype(np.float32) * f(0.75), ka["y"][:nd] + (i3 - 1).astype(np.float32) * f(0.5)], -1).astype(np.float32)
    opx, osl, ook, _ = oex.find_direct_projection_batch([imgA], imgB, t7[:7], EUROC, np.zeros(nd, np.int32), np.tile(ident, (nd, 1)), ka[:nd], world[:nd], px0)
    assert nd == 160 and (dr[:, 2].astype(np.int32) == osl).all() and (dr[:, 3].astype(np.uint8) == ook).all()
    assert np.array_equal(dr[:, :2].copy().view(np.uint32), opx.view(np.uint32))
    assert ook.sum() > 100
    # matcher: CurrentFrame.mTcw = TCR (as computed on the device), LastFrame.mTcw = identity
    Rcw, tcw = _R_from_q(t7[:4]), t7[4:7]
    I, z = np.eye(3, dtype=np.float32), np.zeros(3, np.float32)
    e_n, e_m, _ = oracle.search_by_projection_last(kb, db, oex.tables()["scale"], w, h, EUROC, ka, world, da, Rcw, tcw, I, z, 15.0)
    nm = int(np.fromfile(tmp_path / "nmatch.bin", np.int32)[0])
    assigned = np.fromfile(tmp_path / "match.bin", np.int32)
    assert nm == e_n and nm > 100
    exp = np.where(e_m >= 0, e_m, -1)
    assert (assigned == exp).all()
    # SearchByProjection(F, MapPoints): the local points of A matched back into A
    M = len(ka)
    idx = np.arange(M)
    tiv = (idx % 7 != 0).astype(np.uint8)
    px = (ka["x"] + f(0.5)).astype(np.float32)
    py = (ka["y"] - f(0.25)).astype(np.float32)
    vc = np.where(idx % 3 != 0, f(0.9995), f(0.99)).astype(np.float32)
    e_n2, e_m2, _ = oracle.search_by_projection_mappoints(ka, da, oex.tables()["scale"], w, h, EUROC, tiv, px, py, vc, ka["octave"], da, 3.0,
                                                          True, 0.8)
    nm2 = int(np.fromfile(tmp_path / "nmatch2.bin", np.int32)[0])
    assigned2 = np.fromfile(tmp_path / "match2.bin", np.int32)
    assert nm2 == e_n2 and nm2 > 100
    assert (assigned2 == np.where(e_m2 >= 0, e_m2, -1)).all()
    # Frame overload with existing keys: DSO_KEYPOINT twice (grid size persists in the extractor), then ORBSLAM_KEYPOINT
    oex2 = oracle.Extractor(600, 1.2, 8, 20, 7)
    stale = kb[:150].copy()
    stale["angle"] = 0
    ko, do, g = oex2.extract_dso(imgB, existing=stale)
    kc = np.fromfile(tmp_path / "c_kps.bin", KP_DTYPE)
    dc = np.fromfile(tmp_path / "c_desc.bin", np.uint8).reshape(-1, 32)
    assert len(kc) == len(ko) > 150 and (kc == ko).all() and (dc == do).all()
    ko2, do2, _ = oex2.extract_dso(imgA, existing=ka[:80], grid_size=g)
    kc2 = np.fromfile(tmp_path / "c2_kps.bin", KP_DTYPE)
    dc2 = np.fromfile(tmp_path / "c2_desc.bin", np.uint8).reshape(-1, 32)
    assert len(kc2) == len(ko2) > 80 and (kc2 == ko2).all() and (dc2 == do2).all()
    stale_e = kb[5:125].copy()
    stale_e["angle"] = 0
    koe, doe = oex2.extract_fast(imgB, existing=stale_e)            # FAST_KEYPOINT branch
    ke = np.fromfile(tmp_path / "e_kps.bin", KP_DTYPE)
    de = np.fromfile(tmp_path / "e_desc.bin", np.uint8).reshape(-1, 32)
    assert len(ke) == len(koe) > 1000 and (ke == koe).all() and (de == doe).all()
    kd = np.fromfile(tmp_path / "d_kps.bin", KP_DTYPE)
    dd = np.fromfile(tmp_path / "d_desc.bin", np.uint8).reshape(-1, 32)
    _, dex = oex2.describe_keys(imgB, kb[10:70])
    assert len(kd) == 60 + len(kb) and (kd[:60] == kb[10:70]).all() and (kd[60:] == kb).all()
    assert (dd[:60] == dex).all() and (dd[:60] == db[10:70]).all() and (dd[60:] == db).all()
    # SearchByProjection(cur, KF, found, 10, 100): host prologue in the shell (projection, distance gate, PredictScale) + device search
    sf = oex.tables()["scale"]
    idx = np.arange(M)
    usable = ((idx % 11 != 0) & (idx % 5 != 0)).astype(np.uint8)
    mf_max = (depth * sf[ka["octave"]]).astype(np.float32)
    mf_min = (mf_max / sf[7]).astype(np.float32)
    owner0 = (np.arange(len(kb)) % 13 == 0).astype(np.uint8)
    e_n3, e_m3, _, _ = oracle.search_by_projection_kf(kb, db, sf, w, h, EUROC, usable, world, (f(1.2) * mf_max).astype(np.float32),
                                                      (f(0.8) * mf_min).astype(np.float32), mf_max, ka["angle"], da, Rcw, tcw,
                                                      np.log(f(1.2)), 10.0, 100, True, owner=owner0)
    nm3 = int(np.fromfile(tmp_path / "nmatch3.bin", np.int32)[0])
    assigned3 = np.fromfile(tmp_path / "match3.bin", np.int32)
    assert nm3 == e_n3 and nm3 > 50
    assert (assigned3 == np.where(e_m3 >= 0, e_m3, -1)).all()
    # SearchForInitialization(A, B, prevMatched = A's positions, 100)
    prev = np.stack([ka["x"], ka["y"]], -1).astype(np.float32)
    e_n4, e_m4, e_p4 = oracle.search_for_initialization(ka, da, kb, db, sf, w, h, EUROC, prev, 100, 0.9, True)
    assert int(np.fromfile(tmp_path / "nmatch4.bin", np.int32)[0]) == e_n4 and e_n4 > 50
    assert (np.fromfile(tmp_path / "match4.bin", np.int32) == e_m4).all()
    assert (np.fromfile(tmp_path / "prev4.bin", np.float32).reshape(-1, 2) == e_p4).all()
    # SearchByBoW(KF = A, F = B) with the stand-in FeatureVectors (node = first descriptor byte >> 3; node 7 absent in F)
    na, nb = da[:, 0].astype(np.int32) >> 3, db[:, 0].astype(np.int32) >> 3
    nodes = sorted((set(na.tolist()) & set(nb.tolist())) - {7})
    ko, fo, ki, fi = [0], [0], [], []
    for n in nodes:
        ki.extend(np.nonzero(na == n)[0]); fi.extend(np.nonzero(nb == n)[0])
        ko.append(len(ki)); fo.append(len(fi))
    e_n5, e_m5 = oracle.search_by_bow(ko, ki, fo, fi, (idx % 9 != 0).astype(np.uint8), ka, da, kb, db, 0.7, True)
    assert int(np.fromfile(tmp_path / "nmatch5.bin", np.int32)[0]) == e_n5 and e_n5 > 20
    assert (np.fromfile(tmp_path / "match5.bin", np.int32) == np.where(e_m5 >= 0, e_m5, -1)).all()
    # SearchForTriangulation(KF1 = A, KF2 = B): node = first descriptor byte >> 4, node 5 absent in KF1
    na, nb = da[:, 0].astype(np.int32) >> 4, db[:, 0].astype(np.int32) >> 4
    nodes = sorted((set(na.tolist()) - {5}) & set(nb.tolist()))
    o1, o2, i1, i2 = [0], [0], [], []
    for n in nodes:
        i1.extend(np.nonzero(na == n)[0]); i2.extend(np.nonzero(nb == n)[0])
        o1.append(len(i1)); o2.append(len(i2))
    ia, ib = np.arange(len(ka)), np.arange(len(kb))
    kf1 = dict(keys=ka, desc=da, has_mp=(ia % 5 == 0).astype(np.uint8), u_right=np.where(ia % 3 == 0, ka["x"] - f(4), f(-1)).astype(np.float32))
    kf2 = dict(keys=kb, desc=db, has_mp=(ib % 5 == 1).astype(np.uint8), u_right=np.where(ib % 3 == 0, kb["x"] - f(4), f(-1)).astype(np.float32))
    e_n6, e_m6 = oracle.search_for_triangulation(o1, i1, o2, i2, kf1, kf2, sf, (sf * sf).astype(np.float32), F12, Cw1, R2w, t2w,
                                                 (EUROC["fx"], EUROC["fy"], EUROC["cx"], EUROC["cy"]), False, True)
    assert int(np.fromfile(tmp_path / "nmatch6.bin", np.int32)[0]) == e_n6 and e_n6 > 20
    assert (np.fromfile(tmp_path / "match6.bin", np.int32) == np.where(e_m6 >= 0, e_m6, -1)).all()
    # stereo: right-eye extraction through the shell (leftEye = false) + ComputeStereoMatches
    kr, dr = oex.extract(imgR)
    assert (np.fromfile(tmp_path / "s_kpsr.bin", KP_DTYPE) == kr).all()
    our, odp = oex.compute_stereo_matches(imgA, imgR, ka, da, kr, dr, 0.11, 47.9)
    sur, sdp = np.fromfile(tmp_path / "s_uright.bin", np.float32), np.fromfile(tmp_path / "s_depth.bin", np.float32)
    assert (sur.view(np.uint32) == our.view(np.uint32)).all() and (sdp.view(np.uint32) == odp.view(np.uint32)).all()
    assert (our >= 0).sum() > 100


def test_c_abi_from_plain_c(tmp_path):
    """examples/extract_match.c: the boundary is usable from C99 without any C++ or Python in between."""
    from orb_ygz_slam_amd import load_library
    load_library()
    lib = os.path.join(ROOT, "orb_ygz_slam_amd", "lib")
    exe = str(tmp_path / "extract_match")
    subprocess.check_call(["gcc", "-std=c99", "-O2", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "extract_match.c"),
                           "-L", lib, "-lygzf", "-Wl,-rpath," + lib, "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "keypoints:" in out.stdout


def test_shell_latency_per_frame(tmp_path):
    """One Tracking iteration through the class shells, one frame at a time (host cv::Mat in, std::vector / cv::Mat out): prints the
    per-call latencies INTEGRATION.md quotes and checks that they stay in the range that makes the library a real-time drop-in."""
    from orb_ygz_slam_amd import load_library
    load_library()
    host = os.path.join(ROOT, "orb_ygz_slam_amd", "csrc", "host")
    lib = os.path.join(ROOT, "orb_ygz_slam_amd", "lib")
    exe = os.path.join(str(tmp_path), "shell_latency")
    srcs = [os.path.join(ROOT, "tests", "cpp", "shell_latency.cc")] + [os.path.join(host, f) for f in
                                                                       ("ORBextractor.cc", "ORBmatcher.cc", "SparseImageAlign.cc", "ygzf_pool.cc")]
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-Wall", "-pthread", "-I", host, "-I", os.path.join(host, "standalone")] + srcs +
                          ["-L", lib, "-lygzf", "-Wl,-rpath," + lib, "-o", exe])
    imgA, imgB, _, _ = two_view_scene(9, 752, 480, EUROC, Z=4.0)
    imgA.tofile(tmp_path / "a.u8")
    imgB.tofile(tmp_path / "b.u8")
    out = subprocess.run([exe, str(tmp_path), "200"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    print(out.stdout)
    try:                                                    # travels back from the GPU box with the run's other files (profiles/rNN_*_shell_latency.txt)
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        open(os.path.join(ROOT, "gpurun_out", "shell_latency.txt"), "w").write(out.stdout)
    except OSError:
        pass
    med = {l.split()[0]: float(l.split()[1]) for l in out.stdout.splitlines() if l and not l.startswith("info")}
    assert {"extract_image", "frame_pyramid_plus_extract", "sparse_img_align_run", "search_by_projection_last"} <= set(med)
    # a 30 Hz camera leaves 33 ms per frame; the CPU reference spends ~20 ms in the extractor alone
    assert med["extract_image"] < 3000 and med["frame_pyramid_plus_extract"] < 4000
    assert med["sparse_img_align_run"] < 5000 and med["search_by_projection_last"] < 3000
