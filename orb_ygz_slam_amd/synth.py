"""Seeded synthetic gray frames for tests and bench (SURVEY.md §8d, config 2 generator).

sum of (i) 3-octave value noise, (ii) random filled rectangles with uniform gray U[0,255], (iii) checkerboard
patches 8-24 px, (iv) additive Gaussian noise sigma=2, clipped to u8.  Yields several thousand FAST candidates at
threshold 20 on 640x480 so that the octree actually prunes.  Pure numpy, deterministic per (seed, w, h).
"""
import numpy as np


def _value_noise(rng, w, h, cell):
    gw, gh = w // cell + 2, h // cell + 2
    g = rng.uniform(0.0, 1.0, size=(gh, gw)).astype(np.float32)
    ys = np.arange(h, dtype=np.float32) / cell
    xs = np.arange(w, dtype=np.float32) / cell
    y0 = np.floor(ys).astype(np.int64)
    x0 = np.floor(xs).astype(np.int64)
    fy = (ys - y0)[:, None]
    fx = (xs - x0)[None, :]
    a = g[y0][:, x0]
    b = g[y0][:, x0 + 1]
    c = g[y0 + 1][:, x0]
    d = g[y0 + 1][:, x0 + 1]
    return (a * (1 - fx) + b * fx) * (1 - fy) + (c * (1 - fx) + d * fx) * fy


def synth_frame(seed, w=640, h=480, n_rect=200, n_checker=150, noise_sigma=2.0):
    """Return a (h, w) uint8 image."""
    rng = np.random.default_rng(seed)
    img = np.zeros((h, w), np.float32)
    for cell, amp in ((64, 90.0), (32, 50.0), (16, 30.0)):
        img += amp * _value_noise(rng, w, h, cell)
    img += 40.0
    area = (w * h) / (640.0 * 480.0)
    for _ in range(int(n_rect * area)):
        rw, rh = rng.integers(6, 80), rng.integers(6, 80)
        x, y = rng.integers(-20, w), rng.integers(-20, h)
        g = rng.uniform(0, 255)
        img[max(y, 0):max(y + rh, 0), max(x, 0):max(x + rw, 0)] = g
    for _ in range(int(n_checker * area)):
        s = int(rng.integers(8, 25))
        n = int(rng.integers(2, 5))
        x, y = int(rng.integers(0, max(1, w - s * n))), int(rng.integers(0, max(1, h - s * n)))
        lo, hi = sorted(rng.uniform(0, 255, size=2))
        yy, xx = np.mgrid[0:s * n, 0:s * n]
        pat = np.where(((yy // s) + (xx // s)) % 2 == 0, lo, hi).astype(np.float32)
        sub = img[y:y + s * n, x:x + s * n]
        sub[...] = pat[:sub.shape[0], :sub.shape[1]]
    img += rng.normal(0.0, noise_sigma, size=img.shape).astype(np.float32)
    return np.clip(np.rint(img), 0, 255).astype(np.uint8)


def synth_batch(seed0, n, w=640, h=480):
    return np.stack([synth_frame(seed0 + i, w, h) for i in range(n)])
