"""Inputs for ORBmatcher::SearchForTriangulation (src/ORBmatcher.cc:596-741) shared by the CPU pin (tests/test_ref_matcher.py) and the
device parity test (tests/test_gpu_match.py): two extractions of one scene shifted by (3, -2) px, KF1 at the world origin, KF2 translated so
that a plane at depth Z moves by exactly that shift, F12 built the way LocalMapping::ComputeF12 does (src/LocalMapping.cc:
K1^-T [t12]x R12 K2^-1), in float64 and then rounded to float32.  The variants move the epipole from far outside the image (lateral motion:
most Hamming matches lie on their epipolar lines) into the image (forward motion: the epipole-distance and the epipolar-line tests reject most)."""
import numpy as np

CAM = dict(fx=458.654, fy=457.296, cx=367.215, cy=248.375)


def fake_feature_vector(desc, bits):
    """Stand-in for DBoW2's FeatureVector: node id = leading descriptor bits."""
    node = (desc[:, 0].astype(np.int32) >> (8 - bits)) if bits <= 8 else ((desc[:, 0].astype(np.int32) << (bits - 8)) | (desc[:, 1] >> (16 - bits)))
    return {int(n): np.nonzero(node == n)[0].astype(np.int32) for n in np.unique(node)}


def join(fv1, fv2):
    nodes = sorted(set(fv1) & set(fv2))
    o1, o2, i1, i2 = [0], [0], [], []
    for n in nodes:
        i1.extend(fv1[n]); i2.extend(fv2[n])
        o1.append(len(i1)); o2.append(len(i2))
    return np.array(o1, np.int32), np.array(i1, np.int32), np.array(o2, np.int32), np.array(i2, np.int32)


def skew(t):
    return np.array([[0, -t[2], t[1]], [t[2], 0, -t[0]], [-t[1], t[0], 0]], np.float64)


def geometry(tz, yaw_deg=0.0, Z=5.0, epipole=None):
    a = np.deg2rad(yaw_deg)
    R2w = np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]], np.float64)
    t2w = np.array([3.0 * Z / CAM["fx"], -2.0 * Z / CAM["fy"], tz], np.float64)
    if epipole is not None:     # KF1's centre projects to this pixel of KF2 (R2w = I)
        t2w = np.array([(epipole[0] - CAM["cx"]) / CAM["fx"] * tz, (epipole[1] - CAM["cy"]) / CAM["fy"] * tz, tz], np.float64)
    K = np.array([[CAM["fx"], 0, CAM["cx"]], [0, CAM["fy"], CAM["cy"]], [0, 0, 1]], np.float64)
    R12 = R2w.T
    t12 = -R2w.T @ t2w
    F12 = np.linalg.inv(K).T @ skew(t12) @ R12 @ np.linalg.inv(K)
    return (F12.astype(np.float32), np.zeros(3, np.float32), R2w.astype(np.float32), t2w.astype(np.float32),
            np.array([CAM["fx"], CAM["fy"], CAM["cx"], CAM["cy"]], np.float32))


def cases(ka, da, kb, db, seed=11):
    """-> list of (label, kwargs for search_for_triangulation(off1, idx1, off2, idx2, kf1, kf2, sf, sigma2, F12, Cw1, R2w, t2w, cam2, ...))."""
    rng = np.random.default_rng(seed)
    n1, n2 = len(ka), len(kb)
    mp1 = (rng.uniform(size=n1) < 0.2).astype(np.uint8)
    mp2 = (rng.uniform(size=n2) < 0.2).astype(np.uint8)
    ur1 = np.where(rng.uniform(size=n1) < 0.5, ka["x"] - 4.0, -1.0).astype(np.float32)
    ur2 = np.where(rng.uniform(size=n2) < 0.5, kb["x"] - 4.0, -1.0).astype(np.float32)
    # the keypoint of KF2 with the most level-0..2 neighbours within 8 px: an epipole placed there puts many candidates inside the
    # 100 * scaleFactor exclusion disc of :668-673, where every epipolar line passes close by
    xy = np.stack([kb["x"], kb["y"]], -1).astype(np.float64)
    low = kb["octave"] <= 2
    near = [int(((np.abs(xy[low] - p).max(-1) < 8)).sum()) if low[j] else 0 for j, p in enumerate(xy)]
    hot = xy[int(np.argmax(near))] + 0.5
    out = []
    for label, bits, tz, yaw, stereo, only_stereo, ori in (("lateral", 4, 0.002, 0.0, False, False, True), ("lateral-noori", 6, 0.002, 0.0, False, False, False),
                                                             ("one-node", 0, 0.01, 0.0, False, False, True), ("forward", 3, 0.6, 0.0, False, False, True),
                                                             ("forward-stereo", 3, 0.6, 0.0, True, False, True), ("only-stereo", 4, 0.002, 0.0, True, True, True),
                                                             ("yaw", 5, 0.05, 0.4, True, False, True), ("fine-nodes", 10, 0.002, 0.0, False, False, True),
                                                             ("zero-F", 4, 0.002, 0.0, False, False, True),
                                                             ("epipole-mono", 0, 1.0, 0.0, False, False, False),
                                                             ("epipole-stereo", 0, 1.0, 0.0, "all", False, False)):
        fv1 = fake_feature_vector(da, bits) if bits else {0: np.arange(n1, dtype=np.int32)}
        fv2 = fake_feature_vector(db, bits) if bits else {0: np.arange(n2, dtype=np.int32)}
        o1, i1, o2, i2 = join(fv1, fv2)
        F12, Cw1, R2w, t2w, cam2 = geometry(tz, yaw, epipole=hot if label.startswith("epipole") else None)
        if label == "zero-F":
            F12 = np.zeros((3, 3), np.float32)     # den == 0 -> CheckDistEpipolarLine false for every pair (:146-147)
        u1, u2 = (ur1, ur2) if stereo else (None, None)
        if stereo == "all":
            u1, u2 = np.abs(ur1), np.abs(ur2)
        kf1 = dict(keys=ka, desc=da, has_mp=mp1, u_right=u1)
        kf2 = dict(keys=kb, desc=db, has_mp=mp2, u_right=u2)
        out.append((label, dict(off1=o1, idx1=i1, off2=o2, idx2=i2, kf1=kf1, kf2=kf2, F12=F12, Cw1=Cw1, R2w=R2w, t2w=t2w, cam2=cam2,
                                only_stereo=only_stereo, check_ori=ori)))
    return out
