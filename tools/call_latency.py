#!/usr/bin/env python3
"""tools/call_latency.py -- wall-clock latency of the host-facing entry points at one-frame granularity (ctypes binding, host arrays in and
out, medians of 30 calls): the view a Tracking thread has of the library.  Looks for calls that fall off a fast path (a 10 ms pyramid
read-back and a 3 ms odd-width upload were found this way)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from orb_ygz_slam_amd import Extractor, make_camera, EUROC  # noqa: E402
from orb_ygz_slam_amd.scene import two_view_scene, stereo_scene  # noqa: E402


def med(f, n=30):
    for _ in range(3):
        f()
    t = []
    for _ in range(n):
        t0 = time.perf_counter()
        f()
        t.append(time.perf_counter() - t0)
    return 1e6 * float(np.median(t))


def main():
    w, h = 752, 480
    cam = make_camera(w, h, mb=0.11, mbf=47.9)
    A, B, (R, t), bp = two_view_scene(9, w, h, EUROC)
    ex = Extractor(1000, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=2)
    ka, da = ex.extract(A)
    kb, db = ex.extract(B)
    world = bp(ka["x"], ka["y"])
    ident = np.array([0, 0, 0, 1, 0, 0, 0], np.float32)
    I3, z3 = np.eye(3, dtype=np.float32), np.zeros(3, np.float32)
    inv = ex.tables()["inv_scale"]
    sf = ex.tables()["scale"]
    pa, pb = ex.compute_pyramid(A), ex.compute_pyramid(B)
    rows = []
    rows.append(("extract (image)", med(lambda: ex.extract(A))))
    rows.append(("compute_pyramid", med(lambda: ex.compute_pyramid(A))))
    rows.append(("extract_dso (150 existing keys)", med(lambda: ex.extract_dso(A, ka[:150]))))
    ex.extract(A)
    rows.append(("describe_keys (150 keys, after an extract)", med(lambda: ex.describe_keys(ka[:150]))))
    rows.append(("search_by_projection_last", med(lambda: ex.search_by_projection_last(cam, kb, db, ka, world, da, I3, z3, I3, z3, 15.0))))
    rows.append(("sia_run (levels 7..1)", med(lambda: ex.sia_run(cam, ka, world, ident, pa, ident, pb, inv, 7, 1))))
    n = len(ka)
    mf = (4.0 * sf[ka["octave"]]).astype(np.float32)
    nrm = np.tile(np.array([0, 0, 1], np.float32), (n, 1))
    rows.append(("is_in_frustum_batch (%d points)" % n, med(lambda: ex.is_in_frustum_batch(cam, world, nrm, 1.2 * mf, 0.8 * mf / sf[7], mf, I3, z3, z3,
                                                                                              float(np.log(np.float32(1.2)))))))
    rows.append(("search_local_points (%d points)" % n, med(lambda: ex.search_local_points(cam, kb, db, world, nrm, 1.2 * mf, 0.8 * mf / sf[7], mf, I3, z3, z3,
                                                                                              float(np.log(np.float32(1.2))), da, 3.0))))
    L, Rimg = stereo_scene(3, w, h)[:2]
    kl, dl = ex.extract(L)
    kr, dr = ex.extract(Rimg)
    rows.append(("compute_stereo_matches", med(lambda: ex.compute_stereo_matches(L, Rimg, kl, dl, kr, dr, 0.11, 47.9))))
    ex.image_cache_reserve(4, w, h)
    rows.append(("image_cache_put", med(lambda: ex.image_cache_put(1, A))))
    ex.image_cache_put(0, B)
    m = min(200, n)
    rows.append(("find_direct_projection_batch (%d candidates)" % m,
                 med(lambda: ex.find_direct_projection_batch(cam, 0, ident, np.ones(m, np.int32), np.tile(ident, (m, 1)), ka[:m], world[:m],
                                                             np.stack([ka["x"][:m], ka["y"][:m]], 1).astype(np.float32)))))
    rows.append(("features_in_area (100 windows)", med(lambda: ex.features_in_area(cam, kb, np.stack([ka["x"][:100], ka["y"][:100], np.full(100, 20, np.float32)], 1), cap=128))))
    rows.append(("descriptor_distance (1000 pairs)", med(lambda: ex.descriptor_distance(da[:1000], db[:1000]))))
    for name, us in rows:
        print("%-48s %8.0f us" % (name, us))


if __name__ == "__main__":
    main()
