"""Synthetic two-view scenes with known geometry for the matcher / SparseImgAlign tests and benches (SURVEY.md 8d,
config 3): a fronto-parallel textured plane at depth Z seen by a pinhole camera; frame B is frame A re-rendered after
a known small SE3 motion.  Pure numpy (bilinear resampling of a larger texture)."""
import numpy as np

from .synth import synth_frame


def quat_to_R(q):
    x, y, z, w = [float(v) for v in q]
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]], np.float64)


def rotvec_to_quat(rv):
    rv = np.asarray(rv, np.float64)
    th = np.linalg.norm(rv)
    if th < 1e-12:
        return np.array([0, 0, 0, 1], np.float64)
    ax = rv / th
    return np.concatenate([ax * np.sin(th / 2), [np.cos(th / 2)]])


def render_plane(texture, cam, w, h, R_cw, t_cw, Z, tex_scale, tex_origin):
    """Image of the plane {world z = Z} (texture coords = world x,y * tex_scale + tex_origin) from pose (R_cw, t_cw)."""
    fx, fy, cx, cy = cam["fx"], cam["fy"], cam["cx"], cam["cy"]
    u, v = np.meshgrid(np.arange(w, dtype=np.float64), np.arange(h, dtype=np.float64))
    rays_c = np.stack([(u - cx) / fx, (v - cy) / fy, np.ones_like(u)], -1)          # camera rays
    R_wc = R_cw.T
    o_w = -R_wc @ t_cw                                                               # camera centre in world
    rays_w = rays_c @ R_wc.T
    lam = (Z - o_w[2]) / rays_w[..., 2]
    X = o_w[0] + lam * rays_w[..., 0]
    Y = o_w[1] + lam * rays_w[..., 1]
    tx = X * tex_scale + tex_origin[0]
    ty = Y * tex_scale + tex_origin[1]
    x0 = np.clip(np.floor(tx).astype(np.int64), 0, texture.shape[1] - 2)
    y0 = np.clip(np.floor(ty).astype(np.int64), 0, texture.shape[0] - 2)
    ax, ay = np.clip(tx - x0, 0, 1), np.clip(ty - y0, 0, 1)
    T = texture.astype(np.float64)
    img = (T[y0, x0] * (1 - ax) + T[y0, x0 + 1] * ax) * (1 - ay) + (T[y0 + 1, x0] * (1 - ax) + T[y0 + 1, x0 + 1] * ax) * ay
    return np.clip(np.rint(img), 0, 255).astype(np.uint8)


def two_view_scene(seed, w, h, cam, Z=4.0, rotvec=(0.004, -0.006, 0.003), trans=(0.03, -0.02, 0.015), smooth=True):
    """Returns (imgA, imgB, T_BA as (R, t), backproject(keys)->world points on the plane in A's frame (A = world))."""
    margin = 160
    tex = synth_frame(seed, w + 2 * margin, h + 2 * margin).astype(np.float64)
    if smooth:  # light low-pass so that photometric alignment has usable gradients at every pyramid level
        k = np.array([1, 4, 6, 4, 1], np.float64) / 16
        tex = np.apply_along_axis(lambda r: np.convolve(r, k, mode="same"), 1, tex)
        tex = np.apply_along_axis(lambda r: np.convolve(r, k, mode="same"), 0, tex)
    tex = np.clip(np.rint(tex), 0, 255).astype(np.uint8)
    tex_scale = cam["fx"] / Z                                  # 1 texture px per image px at depth Z
    tex_origin = (cam["cx"] + margin, cam["cy"] + margin)
    I3, z3 = np.eye(3), np.zeros(3)
    R = quat_to_R(rotvec_to_quat(rotvec))
    t = np.asarray(trans, np.float64)
    imgA = render_plane(tex, cam, w, h, I3, z3, Z, tex_scale, tex_origin)
    imgB = render_plane(tex, cam, w, h, R, t, Z, tex_scale, tex_origin)

    def backproject(xs, ys):
        X = (np.asarray(xs, np.float64) - cam["cx"]) / cam["fx"] * Z
        Y = (np.asarray(ys, np.float64) - cam["cy"]) / cam["fy"] * Z
        return np.stack([X, Y, np.full_like(X, Z)], -1).astype(np.float32)

    return imgA, imgB, (R, t), backproject


def stereo_scene(seed, w=752, h=480, disparities=(3, 7, 12, 20, 33, 48, 5, 9), noise=3):
    """A rectified stereo pair: horizontal bands of the left image re-appear in the right image shifted by a known disparity
    (+ small independent noise).  Returns (left, right, band height, disparities)."""
    from .synth import synth_frame
    pad = 64 + max(disparities)
    base = synth_frame(seed, w + 2 * pad, h)
    left = np.ascontiguousarray(base[:, pad:pad + w])
    right = np.zeros_like(left)
    bh = (h + len(disparities) - 1) // len(disparities)
    for b, d in enumerate(disparities):
        right[b * bh:(b + 1) * bh] = base[b * bh:(b + 1) * bh, pad + d:pad + d + w]
    rng = np.random.default_rng(seed + 1000)
    if noise:
        right = np.clip(right.astype(np.int32) + rng.integers(-noise, noise + 1, right.shape), 0, 255).astype(np.uint8)
    return left, right, bh, tuple(disparities)
