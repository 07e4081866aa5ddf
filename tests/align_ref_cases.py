"""Four two-view scenes for SparseImgAlign(max_level, min_level, 10).run(ref, cur, TCR), shared by tools/make_golden_align_ref.py -- which runs them
through THE REFERENCE'S OWN src/SparseImageAlign.cc + NLSSolver (oracle/_ref/libref_orbmatcher.so, over the fixed-size matrix stand-in of
oracle/ref_shim/sia_stubs.h) and commits what came back as tests/golden/align_ref.npz -- and by the tests that hold the oracle (bit for bit) and the
device (1e-5 on the SE3: its sums run in another order) to it.  max_level <= 5: the reference indexes `int iterations[6]` with the level."""
import numpy as np

from orb_ygz_slam_amd.scene import two_view_scene

W, H = 752, 480
CAM = dict(fx=458.654, fy=457.296, cx=367.215, cy=248.375)
CASES = [(21, (0.004, -0.006, 0.003), (0.02, -0.01, 0.03), 5, 1), (22, (-0.01, 0.004, 0.0), (-0.03, 0.02, 0.01), 4, 0),
         (23, (0.002, 0.002, -0.008), (0.0, 0.0, 0.05), 5, 2), (24, (0.03, -0.02, 0.01), (0.3, -0.2, 0.4), 5, 1)]   # last: started too far away -> rollbacks
IDENT = np.array([0, 0, 0, 1, 0, 0, 0], np.float32)


def scene(j, extractor):
    seed, rv, tr, max_level, min_level = CASES[j]
    A, B, _, bp = two_view_scene(seed, W, H, CAM, Z=3.0, rotvec=rv, trans=tr)
    k, _ = extractor.extract(A)
    world = bp(k["x"], k["y"])
    rng = np.random.default_rng(seed)
    valid = (rng.uniform(size=len(k)) > 0.1).astype(np.uint8)
    outl = (rng.uniform(size=len(k)) > 0.95).astype(np.uint8)
    return A, B, k, world, valid, outl, max_level, min_level
