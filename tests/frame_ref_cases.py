"""Scenarios for the Frame / MapPoint members of SURVEY 8f shared by tools/make_golden_frame_ref.py -- which runs them through THE REFERENCE'S OWN
src/Frame.cc and src/MapPoint.cc (oracle/_ref/libref_frame.so, libref_mappoint.so) and commits what they returned as tests/golden/frame_ref.npz --
and by the tests that hold the oracle (CPU tier) and the device (GPU tier) to those bytes: Frame::GetFeaturesInArea (300 windows),
Frame::isInFrustum (two viewing-cosine limits), MapPoint::ComputeDistinctiveDescriptors (twelve tracks)."""
import numpy as np

from orb_ygz_slam_amd.scene import synth_frame

W, H = 752, 480
CAM = dict(fx=458.654, fy=457.296, cx=367.215, cy=248.375, mb=0.11, mbf=47.9)


def frame(extractor):
    k, d = extractor.extract(synth_frame(31, W, H))
    return k, d, extractor.tables()["scale"]


def windows():
    rng = np.random.default_rng(1)
    out = []
    for _ in range(300):
        x, y = float(rng.uniform(-30, W + 30)), float(rng.uniform(-30, H + 30))
        r = float(rng.choice([3.0, 8.0, 15.0, 40.0, 120.0]))
        lo, hi = (-1, -1) if rng.uniform() < 0.3 else (int(rng.integers(-1, 7)), int(rng.integers(-1, 8)))
        out.append((x, y, r, lo, hi))
    return out


def frustum_inputs(ka, sf):
    f32 = np.float32
    rng = np.random.default_rng(90)
    n = len(ka)
    depth = rng.uniform(2.0, 8.0, n).astype(f32)
    world = np.stack([(ka["x"] - f32(CAM["cx"])) / f32(CAM["fx"]) * depth, (ka["y"] - f32(CAM["cy"])) / f32(CAM["fy"]) * depth, depth], -1).astype(f32)
    world[::17, 2] *= -1                                  # behind the camera
    world[5::23, 0] += 50                                 # outside the image
    normal = (world / np.linalg.norm(world, axis=1, keepdims=True)).astype(f32)
    normal[3::19] *= -1                                   # seen from behind
    dist = np.linalg.norm(world, axis=1).astype(f32)
    mf_max = (dist * sf[ka["octave"]]).astype(f32)
    mf_max[7::29] *= 0.3                                  # outside the scale-invariance range
    mf_min = (mf_max / sf[-1]).astype(f32)
    mx, mn = (f32(1.2) * mf_max).astype(f32), (f32(0.8) * mf_min).astype(f32)
    ang = f32(np.deg2rad(0.3))
    Rcw = np.array([[np.cos(ang), 0, np.sin(ang)], [0, 1, 0], [-np.sin(ang), 0, np.cos(ang)]], f32)
    tcw = np.array([0.02, -0.01, 0.03], f32)
    Ow = (-Rcw.T @ tcw).astype(f32)
    return world, normal, mx, mn, mf_max, Rcw, tcw, Ow, np.log(f32(1.2), dtype=f32)


def tracks():
    rng = np.random.default_rng(17)
    sizes = [1, 2, 3, 4, 5, 8, 13, 21, 40, 2, 7, 64]
    off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
    desc = np.zeros((int(off[-1]), 32), np.uint8)
    for p, n in enumerate(sizes):
        base = rng.integers(0, 256, 32).astype(np.uint8)
        for i in range(n):
            desc[off[p] + i] = base ^ np.packbits(rng.uniform(size=256) < rng.choice([0.02, 0.1, 0.3]))
    return off, desc


def frustum_reduce(res):
    """(in_view, projX, projY, projXR, level, viewCos): the fields are defined where in_view is set -- elsewhere they read zero here."""
    iv = np.asarray(res[0]).astype(bool)
    out = {"iv": iv.astype(np.uint8)}
    for name, a in zip(("px", "py", "pxr", "lvl", "vc"), res[1:]):
        a = np.asarray(a)
        out[name] = np.where(iv, a, np.zeros_like(a))
    return out
