#!/usr/bin/env python3
"""tools/sia_phases.py -- one SparseImgAlign pair (752x480, 1000 features, levels 7..1): kernel time from the library's own events; with
YGZF_DEBUG=sia the instrumented kernel also prints its accumulate / reduce / solve / precompute phase clocks."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from orb_ygz_slam_amd import Extractor, make_camera, EUROC  # noqa: E402
from orb_ygz_slam_amd.scene import two_view_scene  # noqa: E402

w, h = 752, 480
A, B, _, bp = two_view_scene(9, w, h, EUROC, Z=4.0)
ex = Extractor(1000, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=1)
k, _ = ex.extract(A)
pa, pb = ex.compute_pyramid(A), ex.compute_pyramid(B)
world = bp(k["x"], k["y"])
ident = np.array([0, 0, 0, 1, 0, 0, 0], np.float32)
inv = ex.tables()["inv_scale"]
cam = make_camera(w, h)
ex.sia_run(cam, k, world, ident, pa, ident, pb, inv, 7, 1, 10)
ex.profile_enable(True)
ex.profile_reset()
for _ in range(10):
    g = ex.sia_run(cam, k, world, ident, pa, ident, pb, inv, 7, 1, 10)
ms, n = ex.profile_read()["k_sia_run"]
print("ret %d, %d linearisations, chi2 %.4f; k_sia_run %.1f us per run (%d runs)" % (g[0], int(g[2][0]), g[2][1], 1e3 * ms / n, n))
