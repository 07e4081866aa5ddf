"""The launch plan of the one-launch pyramid chain (k_pyr_strips) is host arithmetic: checked here without a device.
A workgroup owns a strip of the last level and recomputes, level by level in LDS, every row above that the strip depends on; nothing is ever
read that another workgroup wrote.  That only holds if, for every strip and level, the rows it produces of level l - 1 contain the two source
rows cv::resize takes for every row it produces of level l (fy = (float)((y + 0.5) * scale_y - 0.5), sy = floor(fy), rows
sy and sy + 1 clamped to the level), and if the rows the strips OWN partition every level (each pyramid row is written exactly once)."""
import math

import numpy as np
import pytest

from orb_ygz_slam_amd.capi import pyramid_plan_host

CASES = [(752, 480, 8, 1.2), (640, 480, 8, 1.2), (1241, 376, 8, 1.2), (1920, 1080, 8, 1.2), (641, 479, 8, 1.2), (333, 517, 6, 1.25),
         (752, 480, 12, 1.1), (752, 480, 4, 1.2), (320, 240, 8, 1.2), (200, 150, 3, 1.2), (130, 110, 8, 1.2), (4128, 2000, 6, 1.2), (3840, 2160, 12, 1.2), (2560, 1440, 10, 1.2)]


def _yofs(sh, dh):
    scale_y = 1.0 / (dh / sh)
    return [int(math.floor(np.float32((dy + 0.5) * scale_y - 0.5))) for dy in range(dh)]


@pytest.mark.parametrize("w,h,nl,sf", CASES)
def test_strips_cover_what_they_read_and_partition_what_they_write(w, h, nl, sf):
    plan = pyramid_plan_host(1000, sf, nl, w, h)
    S, lv, rows, base = plan["strips"], plan["levels"], plan["rows"], plan["base"]
    assert lv[0] == (w, h)
    if S == 0:
        pytest.skip("this geometry takes one launch per level")
    assert S in (8, 16, 32, 48, 64) and 0 < plan["lds_bytes"] <= 160 * 1024 and 0 <= base <= nl - 3
    assert (rows[:, :base] == 0).all() and (rows[:, base, 2:] == 0).all()      # levels below the base are not the strips' business; the base is staged, never written
    for l in range(base + 1, nl):
        hl, sh = lv[l][1], lv[l - 1][1]
        yofs = _yofs(sh, hl)
        written = np.zeros(hl, np.int32)
        for s in range(S):
            ca, cb, wa, wb = rows[s, l]
            pa, pb = rows[s, l - 1][:2]
            assert 0 <= ca <= wa <= wb <= cb <= hl and wa < wb, (s, l, ca, cb, wa, wb)
            written[wa:wb] += 1
            need = [min(max(yofs[y] + d, 0), sh - 1) for y in range(ca, cb) for d in (0, 1)]
            assert pa <= min(need) and max(need) < pb, "strip %d: level %d rows [%d, %d) need level %d rows %d..%d, produced [%d, %d)" % (
                s, l, ca, cb, l - 1, min(need), max(need), pa, pb)
        assert (written == 1).all(), "level %d: rows written %s times" % (l, sorted(set(written.tolist())))
    # the halo rows the strips recompute stay a modest multiple of the pyramid
    total = sum(a * b for a, b in lv[base + 1:])
    produced = sum((rows[s, l, 1] - rows[s, l, 0]) * lv[l][0] for s in range(S) for l in range(base + 1, nl))
    assert produced <= 2.5 * total


def test_large_images_keep_one_launch_per_level():
    big = pyramid_plan_host(8000, 1.2, 12, 3840, 2160)                          # the chain from the image does not fit LDS: the strips start further up
    assert big["strips"] > 0 and 1 <= big["base"] <= 6
    assert pyramid_plan_host(4000, 1.2, 8, 1920, 1080)["base"] == 0 and pyramid_plan_host(1000, 1.2, 8, 752, 480)["base"] == 0
    assert pyramid_plan_host(500, 2.0, 4, 640, 480)["strips"] == 0              # exact 2x levels take the area kernel
    assert pyramid_plan_host(1000, 1.2, 2, 752, 480)["strips"] == 0             # one level to produce: nothing to chain


def _resize_rows(src, src_row0, sw, sh, dw, dh, ya, yb):
    """Rows [ya, yb) of cv::resize(INTER_LINEAR, 8UC1) of an sw x sh level to dw x dh, from a buffer that holds the level's rows from
    src_row0 on -- the integer arithmetic of k_pyr_resize_tiled / k_pyr_strips, vectorised."""
    f32 = np.float32
    x = np.arange(dw)
    fx = ((x + 0.5) * (1.0 / (dw / sw)) - 0.5).astype(f32)
    sx = np.floor(fx).astype(np.int64)
    fx = (fx - sx.astype(f32)).astype(f32)
    lo, hi = sx < 0, sx >= sw - 1
    sx = np.where(lo, 0, np.where(hi, sw - 1, sx))
    fx = np.where(lo | hi, f32(0), fx).astype(f32)
    a0 = np.rint((f32(1) - fx) * f32(2048)).astype(np.int64)
    a1 = np.rint(fx * f32(2048)).astype(np.int64)
    sx1 = np.minimum(sx + 1, sw - 1)
    y = np.arange(ya, yb)
    fy = ((y + 0.5) * (1.0 / (dh / sh)) - 0.5).astype(f32)
    sy = np.floor(fy).astype(np.int64)
    fy = (fy - sy.astype(f32)).astype(f32)
    b0 = np.rint((f32(1) - fy) * f32(2048)).astype(np.int64)
    b1 = np.rint(fy * f32(2048)).astype(np.int64)
    r0 = np.clip(sy, 0, sh - 1) - src_row0
    r1 = np.clip(sy + 1, 0, sh - 1) - src_row0
    assert r0.min() >= 0 and r1.max() < src.shape[0], "the strip does not hold the rows it reads"
    s = src.astype(np.int64)
    H0 = s[r0][:, sx] * a0 + s[r0][:, sx1] * a1
    H1 = s[r1][:, sx] * a0 + s[r1][:, sx1] * a1
    return (((((b0[:, None] * (H0 >> 4)) >> 16) + ((b1[:, None] * (H1 >> 4)) >> 16) + 2) >> 2)).astype(np.uint8)


@pytest.mark.parametrize("w,h,nl,sf", [(752, 480, 8, 1.2), (641, 479, 8, 1.2), (333, 517, 6, 1.25), (200, 150, 3, 1.2), (2560, 1440, 10, 1.2)])
def test_strips_assemble_the_oracle_pyramid(oracle, w, h, nl, sf):
    """The algorithm of k_pyr_strips replayed on the CPU from the library's own plan: every strip stages its level-0 rows, produces its rows
    of every level from its OWN rows of the level below, and writes the rows it owns -- the assembled levels are the oracle's pyramid."""
    from orb_ygz_slam_amd.synth import synth_frame
    img = synth_frame(5, w, h)
    want = oracle.Extractor(500, sf, nl, 20, 7).pyramid(img)
    plan = pyramid_plan_host(500, sf, nl, w, h)
    S, lv, rows, base = plan["strips"], plan["levels"], plan["rows"], plan["base"]
    assert S > 0
    assert (_resize_rows(img, 0, w, h, lv[1][0], lv[1][1], 0, lv[1][1]) == want[1]).all()   # the model itself against the oracle's resize
    got = [img] + [np.full((lv[l][1], lv[l][0]), 0xAA, np.uint8) for l in range(1, nl)]
    for l in range(1, base + 1):
        got[l] = want[l]                                                                     # (one launch each, k_pyr_resize_tiled: not the strips' business)
    for s in range(S):
        ca, cb = rows[s, base][:2]
        held, held0 = got[base][ca:cb], ca
        for l in range(base + 1, nl):
            ca, cb, wa, wb = rows[s, l]
            held = _resize_rows(held, held0, lv[l - 1][0], lv[l - 1][1], lv[l][0], lv[l][1], ca, cb)
            held0 = ca
            got[l][wa:wb] = held[wa - ca:wb - ca]
    for l in range(1, nl):
        assert (got[l] == want[l]).all(), "level %d" % l
