"""CPU tests of the oracle's OpenCV-primitive restatements against independent definitions (SURVEY App. B sanity
properties).  These do not need OpenCV: every check is a property the primitive must have by definition."""
import math
import os

import numpy as np
import pytest

from orb_ygz_slam_amd.synth import synth_frame

RING = [(0, 3), (1, 3), (2, 2), (3, 1), (3, 0), (3, -1), (2, -2), (1, -3), (0, -3), (-1, -3), (-2, -2), (-3, -1),
        (-3, 0), (-3, 1), (-2, 2), (-1, 3)]


def brute_is_corner(img, x, y, t, arc=9):
    v = int(img[y, x])
    d = [int(img[y + dy, x + dx]) - v for dx, dy in RING]
    for s in range(16):
        seg = [d[(s + k) % 16] for k in range(arc)]
        if all(e > t for e in seg) or all(e < -t for e in seg):
            return True
    return False


def test_tables(oracle):
    ex = oracle.Extractor(1000, 1.2, 8, 20, 7)
    t = ex.tables()
    assert list(t["nfeat"]) == [217, 181, 151, 126, 105, 87, 73, 60]
    assert list(t["umax"]) == [15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3]
    assert [ex.level_size(752, 480, l) for l in range(8)] == [(752, 480), (627, 400), (522, 333), (435, 278),
                                                             (363, 231), (302, 193), (252, 161), (210, 134)]
    ex4 = oracle.Extractor(4000, 1.2, 8, 20, 7)
    assert list(ex4.tables()["nfeat"]) == [869, 724, 603, 503, 419, 349, 291, 242]
    assert ex4.level_size(1920, 1080, 7) == (536, 301)
    ex12 = oracle.Extractor(8000, 1.2, 12, 20, 7)
    assert list(ex12.tables()["nfeat"]) == [1502, 1251, 1043, 869, 724, 604, 503, 419, 349, 291, 243, 202]
    assert ex12.level_size(3840, 2160, 11) == (517, 291)


def test_cv_round_half_even(oracle):
    L = oracle.lib()
    assert [L.yo_cv_round(v) for v in (0.5, 1.5, 2.5, -0.5, -1.5, 2.4999, 2.5001)] == [0, 2, 2, 0, -2, 2, 3]


def test_fast9_score_is_max_threshold(oracle):
    rng = np.random.default_rng(1)
    img = rng.integers(0, 256, size=(24, 28), dtype=np.uint8)
    img[8:16, 8:20] //= 4  # structure so that there are corners at several thresholds
    for thr in (7, 20, 40):
        xs, ys, sc = oracle.fast9(img, thr, nonmax=False)
        got = {(int(x), int(y)): int(s) for x, y, s in zip(xs, ys, sc)}
        for y in range(3, img.shape[0] - 3):
            for x in range(3, img.shape[1] - 3):
                c = brute_is_corner(img, x, y, thr)
                assert ((x, y) in got) == c
                if c:
                    tmax = max(t for t in range(thr, 256) if brute_is_corner(img, x, y, t))
                    assert got[(x, y)] == tmax, (x, y, thr)


def test_fast9_nms_properties(oracle):
    img = synth_frame(3, 160, 120)
    for thr in (7, 20):
        xa, ya, sa = oracle.fast9(img, thr, nonmax=False)
        smap = np.zeros(img.shape, np.int32)
        smap[ya, xa] = sa
        xs, ys, sc = oracle.fast9(img, thr, nonmax=True)
        kept = set(zip(xs.tolist(), ys.tolist()))
        assert len(kept) > 5
        exp = set()
        for x, y, s in zip(xa.tolist(), ya.tolist(), sa.tolist()):
            nb = smap[y - 1:y + 2, x - 1:x + 2].copy()
            nb[1, 1] = -1
            if s > nb.max():
                exp.add((x, y))
        assert kept == exp
        order = list(zip(ys.tolist(), xs.tolist()))
        assert order == sorted(order)  # raster order


def test_resize_properties(oracle):
    const = np.full((40, 60), 137, np.uint8)
    assert (oracle.resize(const, 50, 33) == 137).all()
    img = synth_frame(5, 120, 90)
    assert (oracle.resize(img, 120, 90) == img).all()  # scale 1 is the identity
    dw, dh = 100, 75
    got = oracle.resize(img, dw, dh).astype(np.float64)
    sx, sy = img.shape[1] / dw, img.shape[0] / dh
    xs = (np.arange(dw) + 0.5) * sx - 0.5
    ys = (np.arange(dh) + 0.5) * sy - 0.5
    x0 = np.clip(np.floor(xs).astype(int), 0, img.shape[1] - 2)
    y0 = np.clip(np.floor(ys).astype(int), 0, img.shape[0] - 2)
    fx = np.clip(xs - x0, 0, 1)[None, :]
    fy = np.clip(ys - y0, 0, 1)[:, None]
    f = img.astype(np.float64)
    ref = (f[y0][:, x0] * (1 - fx) + f[y0][:, x0 + 1] * fx) * (1 - fy) + (f[y0 + 1][:, x0] * (1 - fx) + f[y0 + 1][:, x0 + 1] * fx) * fy
    assert np.abs(got - ref).max() <= 1.0
    # exact factor 2 takes the 2x2 area-average path
    a = oracle.resize(img, 60, 45).astype(np.int32)
    i = img.astype(np.int32)
    assert (a == ((i[0::2, 0::2] + i[0::2, 1::2] + i[1::2, 0::2] + i[1::2, 1::2] + 2) >> 2)).all()


def test_blur_properties(oracle):
    k = np.array([18, 34, 49, 55, 49, 34, 18], np.int64)
    assert k.sum() == 257
    for c in (0, 1, 128, 200, 255):
        out = oracle.blur(np.full((20, 30), c, np.uint8))
        assert (out == min(255, (c * 257 * 257 + 32768) >> 16)).all()
    img = synth_frame(7, 64, 48)
    out = oracle.blur(img)
    # independent numpy evaluation with REFLECT_101 padding
    p = np.pad(img.astype(np.int64), 3, mode="reflect")
    rows = sum(k[i] * p[:, i:i + img.shape[1]] for i in range(7))
    ref = sum(k[i] * rows[i:i + img.shape[0], :] for i in range(7))
    ref = np.clip((ref + 32768) >> 16, 0, 255)
    assert (out == ref).all()
    assert (oracle.blur(img[:, ::-1])[:, ::-1] == out).all()  # symmetric kernel


def test_fast_atan2(oracle):
    assert oracle.fast_atan2(0.0, 0.0) == 0.0
    assert oracle.fast_atan2(0.0, 5.0) == 0.0
    assert abs(oracle.fast_atan2(5.0, 0.0) - 90.0) < 1e-4
    assert abs(oracle.fast_atan2(0.0, -5.0) - 180.0) < 1e-4
    assert abs(oracle.fast_atan2(-5.0, 0.0) - 270.0) < 1e-4
    rng = np.random.default_rng(0)
    for y, x in rng.integers(-200000, 200000, size=(2000, 2)):
        a = oracle.fast_atan2(float(y), float(x))
        t = math.degrees(math.atan2(y, x)) % 360.0
        d = abs(a - t)
        assert min(d, 360 - d) <= 0.3
        assert 0.0 <= a <= 360.0


def test_sincos_is_correctly_rounded_float(oracle):
    """oracle sincos_deg == float(cos(double(angle_rad))) == the semantics glibc cosf/sinf aim for."""
    rng = np.random.default_rng(2)
    angles = np.concatenate([rng.uniform(0, 360, 200000).astype(np.float32),
                             np.arange(0, 361, 0.5, dtype=np.float32)])
    factor = np.float32(math.pi / np.float32(180.0))
    rad = (angles * factor).astype(np.float32)
    c_ref = np.cos(rad.astype(np.float64)).astype(np.float32)
    s_ref = np.sin(rad.astype(np.float64)).astype(np.float32)
    import ctypes as C
    L = oracle.lib()
    c, s = C.c_float(), C.c_float()
    for i in range(len(angles)):
        L.yo_sincos_deg(float(angles[i]), C.byref(c), C.byref(s))
        assert c.value == c_ref[i] and s.value == s_ref[i], (angles[i], c.value, c_ref[i], s.value, s_ref[i])


def test_descriptor_angle0_is_unrotated_pattern(oracle):
    import re
    ex = oracle.Extractor()
    img = synth_frame(11, 96, 96)
    bl = oracle.blur(img)
    d = ex.descriptor(bl, 48.0, 48.0, 0.0)
    txt = open(os.path.join(os.path.dirname(oracle.__file__), "orb_pattern_table.h")).read()
    nums = [int(v) for v in re.findall(r"-?\d+", txt[txt.index("{"):])]
    assert len(nums) == 1024
    bits = []
    for i in range(256):
        x0, y0, x1, y1 = nums[4 * i:4 * i + 4]
        bits.append(1 if int(bl[48 + y0, 48 + x0]) < int(bl[48 + y1, 48 + x1]) else 0)
    exp = np.packbits(np.array(bits, np.uint8).reshape(32, 8)[:, ::-1], axis=1).ravel()
    assert (d == exp).all()
    # 90 degrees: (x, y) -> (x*a - y*b, x*b + y*a) with a=cos=~0, b=sin=1 -> (-y, x)
    d90 = ex.descriptor(bl, 48.0, 48.0, 90.0)
    bits = []
    for i in range(256):
        x0, y0, x1, y1 = nums[4 * i:4 * i + 4]
        bits.append(1 if int(bl[48 + x0, 48 - y0]) < int(bl[48 + x1, 48 - y1]) else 0)
    exp = np.packbits(np.array(bits, np.uint8).reshape(32, 8)[:, ::-1], axis=1).ravel()
    assert (d90 == exp).all()


def test_ic_angle(oracle):
    ex = oracle.Extractor()
    um = ex.tables()["umax"]
    img = synth_frame(13, 80, 80)
    m10 = m01 = 0
    for v in range(-15, 16):
        for u in range(-um[abs(v)], um[abs(v)] + 1):
            m10 += u * int(img[40 + v, 40 + u])
            m01 += v * int(img[40 + v, 40 + u])
    assert ex.ic_angle(img, 40.0, 40.0) == oracle.fast_atan2(float(m01), float(m10))
    assert sum(2 * um[abs(v)] + 1 for v in range(-15, 16)) == 749


def test_hamming(oracle):
    rng = np.random.default_rng(4)
    for _ in range(200):
        a = rng.integers(0, 256, 32, dtype=np.uint8)
        b = rng.integers(0, 256, 32, dtype=np.uint8)
        assert oracle.hamming(a, b) == int(np.unpackbits(a ^ b).sum())
    assert oracle.hamming(np.zeros(32, np.uint8), np.full(32, 255, np.uint8)) == 256
