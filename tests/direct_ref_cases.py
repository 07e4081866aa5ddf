"""Two rendered two-view scenes for ORBmatcher::FindDirectProjection, shared by tools/make_golden_direct_ref.py -- which runs every keypoint of the
reference view through THE REFERENCE'S OWN FindDirectProjection / GetWarpAffineMatrix / WarpAffine / GetBestSearchLevel and src/Align.cc
(oracle/_ref/libref_orbmatcher.so) and commits what came back as tests/golden/direct_ref.npz -- and by the tests that hold the oracle (CPU tier) and
the device (GPU tier) to those bytes."""
import numpy as np

from orb_ygz_slam_amd.scene import rotvec_to_quat, two_view_scene

W, H = 752, 480
CAM = dict(fx=458.654, fy=457.296, cx=367.215, cy=248.375)
SCENES = ((9, (0.01, -0.02, 0.03), (0.1, -0.05, 0.2)), (4, (-0.03, 0.01, -0.02), (-0.2, 0.1, -0.3)))


def scene(j, extractor):
    seed, rv, tr = SCENES[j]
    A, B, (R, t), bp = two_view_scene(seed, W, H, CAM, Z=4.0, rotvec=rv, trans=tr)
    ka, _ = extractor.extract(A)
    world = bp(ka["x"], ka["y"])
    q = rotvec_to_quat(rv)
    cur7 = np.array([q[0], q[1], q[2], q[3], *tr], np.float32)
    ref7 = np.tile(np.array([0, 0, 0, 1, 0, 0, 0], np.float32), (len(ka), 1))
    Xc = (R @ world.T.astype(np.float64)).T + t
    u = CAM["fx"] * Xc[:, 0] / Xc[:, 2] + CAM["cx"]
    v = CAM["fy"] * Xc[:, 1] / Xc[:, 2] + CAM["cy"]
    rng = np.random.default_rng(seed)
    px0 = np.stack([u, v], -1) + rng.uniform(-2.5, 2.5, (len(ka), 2))
    px0[::17] += 400.0                                  # candidates that leave the image: Align2D breaks out, success = false
    return A, B, cur7, np.zeros(len(ka), np.int32), ref7, ka, world, px0
