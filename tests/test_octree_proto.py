"""The range/path-key formulation of DistributeOctTree (tools/octree_proto.py, the executable specification of the
HIP kernel k_octree) against the oracle's literal std::list restatement of reference src/ORBextractor.cc:533-723."""
import os
import sys

import numpy as np

from orb_ygz_slam_amd.synth import synth_frame

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
import octree_proto  # noqa: E402


def test_prototype_equals_oracle(oracle):
    ex = oracle.Extractor(1000, 1.2, 8, 20, 7)
    n = 0
    for seed, (w, h) in ((0, (640, 480)), (1, (333, 517))):
        img = synth_frame(seed, w, h)
        pyr = ex.pyramid(img)
        for l in (0, 3, 7):
            xs, ys, sc = ex.cell_candidates(l)
            if len(xs) == 0:
                continue
            lw, lh = pyr[l].shape[1], pyr[l].shape[0]
            for N in (int(ex.tables()["nfeat"][l]), 1, 7, 5000):
                a = ex.octree(xs, ys, sc, 16, lw - 16, 16, lh - 16, N)
                b = octree_proto.distribute(xs, ys, sc, 16, lw - 16, 16, lh - 16, N)
                assert len(a) == len(b) and (a == b).all()
                n += 1
    assert n >= 20


def test_octree_hand_cases(oracle):
    ex = oracle.Extractor(1000, 1.2, 8, 20, 7)
    # two roots (W/H = 2), one point each -> both kept, list order = root order
    r = ex.octree([10, 150], [10, 10], [50, 60], 0, 200, 0, 100, 5)
    assert list(r) == [0, 1]
    # four points in one root, N=1: one full subdivision pass still happens, best response per child survives
    xs, ys, sc = [10, 90, 10, 90], [10, 10, 90, 90], [10, 20, 30, 40]
    r = ex.octree(xs, ys, sc, 0, 100, 0, 100, 1)
    assert sorted(r.tolist()) == [0, 1, 2, 3] and list(r) == [3, 2, 1, 0]     # push_front: n4 first
    # equal responses: the first point in input order wins inside a node
    r = ex.octree([10, 11], [10, 10], [33, 33], 0, 100, 0, 100, 1)
    assert len(r) in (1, 2)
