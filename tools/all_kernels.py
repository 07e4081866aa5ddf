#!/usr/bin/env python3
"""tools/all_kernels.py -- drives every kernel of libygzf that the default bench does not reach, at the sizes a tracking frame produces, so that
one rocprofv3 run (tools/profile_round.sh) yields kernel stats and PMC counters for all of SURVEY 8a / 8f:
  k_sia_run (config 3: --align batch), k_stereo_* (config 5 shape), k_dso_cells + k_dso_* + k_describe_list (DSO_KEYPOINT), k_f10_* (libfast),
  k_direct_projection (2000 candidates over 20 KeyFrames), k_frustum + k_match_last mode 1 (4000 local MapPoints), k_match_last modes 2 / 3,
  k_distinctive (2000 MapPoints x 8 observations), k_features_in_area (1000 windows), k_bow_nodes (SearchByBoW), k_tri_nodes (SearchForTriangulation), k_bow_descend (Frame::ComputeBoW, k = 10 / L = 6 vocabulary),
  k_hamming_pairs.
Prints one JSON line with the library's own per-kernel event timings (ygzf_profile_*)."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import make_frames  # noqa: E402
from orb_ygz_slam_amd import Extractor, make_camera  # noqa: E402
from orb_ygz_slam_amd.capi import EUROC, KP_DTYPE  # noqa: E402
from orb_ygz_slam_amd.scene import rotvec_to_quat, two_view_scene  # noqa: E402


def full_vocabulary(k, L, seed=3):
    rng = np.random.default_rng(seed)
    sizes = [k ** l for l in range(L + 1)]
    n = sum(sizes)
    parent = np.full(n, -1, np.int32)
    start = np.cumsum([0] + sizes)
    for l in range(1, L + 1):
        parent[start[l]:start[l + 1]] = start[l - 1] + np.arange(sizes[l]) // k
    return parent, rng.integers(0, 256, (n, 32), dtype=np.uint8)


def main():
    import torch
    torch.cuda.init()
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    w, h = 752, 480
    cam = make_camera(w, h)
    rng = np.random.default_rng(0)
    frames = make_frames(64, w, h)
    ex = Extractor(1000, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=64)
    ex.profile_enable(True)
    ex.profile_reset()
    A, B, (R, t), bp = two_view_scene(9, w, h, EUROC, Z=4.0, rotvec=(0.01, -0.02, 0.03), trans=(0.1, -0.05, 0.2))
    ka, da = ex.extract(A)
    kb, db = ex.extract(B)
    world = bp(ka["x"], ka["y"])
    q = rotvec_to_quat((0.01, -0.02, 0.03))
    T7 = np.array([q[0], q[1], q[2], q[3], 0.1, -0.05, 0.2], np.float32)
    ident = np.array([0, 0, 0, 1, 0, 0, 0], np.float32)
    I, z = np.eye(3, dtype=np.float32), np.zeros(3, np.float32)
    Rm = R.astype(np.float32)
    tv = np.asarray(t, np.float32)
    for _ in range(reps):
        # config 3 / config 5 shapes on a resident batch
        ex.extract_batch_host(frames)
        ex.match_batch_prev(cam, 15.0, True, True, True)
        ex.align_batch_prev(cam, 7, 1, 10)
        ex.stereo_batch(0.11, 47.9)
        ex.sync()
        # DSO_KEYPOINT + libfast
        ex.extract_dso(B, existing=kb[:150])
        ex.fast10(A, 20)
        # direct projection: 2000 candidates, 20 KeyFrame slots + the current frame
        ex.image_cache_reserve(21, w, h)
        for s in range(20):
            ex.image_cache_put(s, A)
        ex.image_cache_put(20, B)
        n = 2000
        idx = rng.integers(0, len(ka), n)
        Xc = (R @ world[idx].T.astype(np.float64)).T + t
        px0 = np.stack([EUROC["fx"] * Xc[:, 0] / Xc[:, 2] + EUROC["cx"], EUROC["fy"] * Xc[:, 1] / Xc[:, 2] + EUROC["cy"]], -1) + rng.uniform(-1.5, 1.5, (n, 2))
        ex.find_direct_projection_batch(cam, 20, T7, rng.integers(0, 20, n).astype(np.int32), np.tile(ident, (n, 1)), ka[idx], world[idx], px0.astype(np.float32))
        # SearchLocalPoints: isInFrustum + SearchByProjection(F, MapPoints) over 4000 local MapPoints
        M = 4000
        mi = rng.integers(0, len(ka), M)
        sf = ex.tables()["scale"]
        mf = (4.0 * sf[ka["octave"][mi]]).astype(np.float32)
        Ow = -(Rm.T @ tv)
        ex.search_local_points(cam, kb, db, world[mi], np.tile(np.array([0, 0, 1], np.float32), (M, 1)), 1.2 * mf, 0.8 * mf / sf[7], mf, Rm, tv, Ow.astype(np.float32),
                               float(np.log(np.float32(1.2))), da[mi], 3.0, False, 0.8, 0.5)
        # relocalisation refinement (mode 2) and initialisation (mode 3)
        ex.search_for_initialization(cam, ka, da, kb, db, np.stack([ka["x"], ka["y"]], -1).astype(np.float32), 100, 0.9, True)
        # distinctive descriptors: 2000 MapPoints x 8 observations
        P = 2000
        off = np.arange(P + 1, dtype=np.int32) * 8
        obs = da[rng.integers(0, len(da), P * 8)] ^ (rng.integers(0, 256, (P * 8, 32), dtype=np.uint8) & rng.integers(0, 2, (P * 8, 32), dtype=np.uint8))
        ex.distinctive_descriptors_batch(off, obs)
        # SearchByBoW on a joined node list
        na, nb = da[:, 0].astype(np.int32) >> 3, db[:, 0].astype(np.int32) >> 3
        ko, fo, ki, fi = [0], [0], [], []
        for node in sorted(set(na.tolist()) & set(nb.tolist())):
            ki.extend(np.nonzero(na == node)[0]); fi.extend(np.nonzero(nb == node)[0])
            ko.append(len(ki)); fo.append(len(fi))
        ex.search_by_bow(ko, ki, fo, fi, np.ones(len(ka), np.uint8), ka, da, kb, db, 0.7, True)
        # SearchForTriangulation on the same node list (LocalMapping::CreateNewMapPoints): k_tri_nodes
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
        from tests.tri_cases import geometry
        F12, Cw1, R2w, t2w, cam2 = geometry(0.002)
        none = np.zeros(max(len(ka), len(kb)), np.uint8)
        ex.search_for_triangulation(ko, ki, fo, fi, dict(keys=ka, desc=da, has_mp=none[:len(ka)], u_right=None),
                                    dict(keys=kb, desc=db, has_mp=none[:len(kb)], u_right=None), None, None, F12, Cw1, R2w, t2w, cam2, False, True)
        ex.descriptor_distance(da[:1000], db[:1000])
        # Frame::GetFeaturesInArea: 1000 windows of the matcher's size around the frame's own keypoints
        ex.features_in_area(cam, kb, np.stack([ka["x"][:1000], ka["y"][:1000], np.full(min(1000, len(ka)), 15.0 * 1.2, np.float32)], 1), cap=256)
    # Frame::ComputeBoW with an ORBvoc-sized tree (k = 10, L = 6: 1 111 111 nodes, 35.6 MB of centroids)
    parent, vdesc = full_vocabulary(10, 6)
    ex.vocabulary_set(parent, vdesc, 6)
    for _ in range(reps):
        ex.bow_transform(np.concatenate([da, db]), 4)
    prof = {k: {"launches": n, "avg_us": round(1e3 * ms / n, 2)} for k, (ms, n) in ex.profile_read().items() if n}
    print(json.dumps({"kernels": prof, "candidates_direct": 2000, "mappoints_frustum": 4000, "points_distinctive": 2000, "bow_features": int(len(da) + len(db))}))


if __name__ == "__main__":
    main()
