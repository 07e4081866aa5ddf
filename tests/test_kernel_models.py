"""CPU models of the arithmetic tricks the HIP kernels rely on, checked exhaustively / on large random samples against the plain
definitions.  They guard the algebra (saturation corners, rounding, exactness ranges) independently of a GPU:

* k_fast_quads' byte-sliced FAST-9 test: v_lerp_u8 as a four-byte comparator + bit-sliced 3-input arc trees
  (orb_ygz_slam_amd/csrc/extract_kernels.hip: fast9_quad) against the scalar "9 contiguous of 16" definition (cv::FAST 9/16);
* the (a * m) >> 16 small division (div_small / kRcp16);
* the horizontal blur as two v_dot4_u32_u8 and the pyramid row sum as v_dot2_u32_u16 (plain integer identities);
* the NMS shortcut "iniTh survivor == minTh survivor with score >= iniTh".
"""
import numpy as np


def lerp_u8(a, b, c):
    """V_LERP_U8 per byte: (a + b + (c & 1)) >> 1 on uint8 arrays."""
    return ((a.astype(np.uint16) + b.astype(np.uint16) + (c.astype(np.uint16) & 1)) >> 1).astype(np.uint8)


def fast9_quad_model(center, ring, t):
    """center: (n,) uint8, ring: (n, 16) uint8, t: int -> polarity per pixel (0 none, 1 bright, 2 dark), using only bit 7 of byte results."""
    c = center.astype(np.int32)
    nc = 255 - c                                           # ~b3
    nhi = np.maximum(nc - t, 0).astype(np.uint8)           # v_pk_sub_u16 clamp
    nlo = np.minimum(nc + t, 255).astype(np.uint8)         # v_pk_add_u16 + v_pk_min_u16
    zero, one = np.zeros_like(center), np.ones_like(center)
    B = [lerp_u8(ring[:, k], nhi, zero) for k in range(16)]
    N = [lerp_u8(ring[:, k], nlo, one) for k in range(16)]
    A3 = [B[k] & B[(k + 1) & 15] & B[(k + 2) & 15] for k in range(16)]
    O3 = [N[k] | N[(k + 1) & 15] | N[(k + 2) & 15] for k in range(16)]
    bright = np.zeros_like(center)
    ndark = np.full_like(center, 255)
    for k in range(16):
        bright |= A3[k] & A3[(k + 3) & 15] & A3[(k + 6) & 15]
        ndark &= O3[k] | O3[(k + 3) & 15] | O3[(k + 6) & 15]
    fb = (bright & 0x80) != 0
    fd = ((~(ndark | bright)) & 0x80) != 0
    return np.where(fb, 1, np.where(fd, 2, 0))


def fast9_scalar(center, ring, t):
    c = center.astype(np.int32)[:, None]
    r = ring.astype(np.int32)
    br, dk = r > c + t, r < c - t

    def arc(m):
        m2 = np.concatenate([m, m[:, :8]], axis=1)
        out = np.zeros(len(m), bool)
        for k in range(16):
            out |= m2[:, k:k + 9].all(axis=1)
        return out
    b, d = arc(br), arc(dk)
    return np.where(b, 1, np.where(d, 2, 0))


def test_fast9_byte_sliced_model_matches_definition():
    rng = np.random.default_rng(7)
    n = 400_000
    for t in (1, 7, 20, 100, 254):
        center = rng.integers(0, 256, n).astype(np.uint8)
        # rings biased towards the centre +- t so that arcs of 8 / 9 / 10 and the saturation corners are all frequent
        delta = rng.integers(-t - 3, t + 4, (n, 16))
        ring = np.clip(center[:, None].astype(np.int32) + delta * rng.integers(0, 3, (n, 16)), 0, 255).astype(np.uint8)
        edge = rng.integers(0, 4, n)
        center = np.where(edge == 0, rng.choice([0, 1, 254, 255], n), center).astype(np.uint8)
        assert (fast9_quad_model(center, ring, t) == fast9_scalar(center, ring, t)).all(), t


def test_fast9_model_structured_arcs():
    """Every arc start x every arc length 7..11 x both polarities, at the saturation corners of centre +- t."""
    rows_c, rows_r = [], []
    for c in (0, 5, 7, 8, 128, 247, 248, 250, 255):
        for pol in (+1, -1):
            for start in range(16):
                for length in (7, 8, 9, 10, 11, 16):
                    for margin in (7, 8):     # exactly t (not a corner pixel) and t + 1
                        ring = np.full(16, c, np.int32)
                        for i in range(length):
                            ring[(start + i) & 15] = c + pol * margin
                        rows_c.append(c)
                        rows_r.append(np.clip(ring, 0, 255))
    center = np.array(rows_c, np.uint8)
    ring = np.array(rows_r, np.uint8)
    assert (fast9_quad_model(center, ring, 7) == fast9_scalar(center, ring, 7)).all()
    assert fast9_scalar(center, ring, 7).max() == 2          # the set does contain bright and dark corners


def test_small_division_table_is_exact():
    for x in range(1, 65):
        m = 65536 // x + 1
        assert m < (1 << 24)                                 # __umul24 operand range
        for a in range(0, 65):
            assert (a * m) >> 16 == a // x, (a, x)


def test_dot_product_forms_of_blur_and_resize():
    rng = np.random.default_rng(3)
    q = rng.integers(0, 256, (100_000, 8)).astype(np.int64)
    ref = 18 * (q[:, 0] + q[:, 6]) + 34 * (q[:, 1] + q[:, 5]) + 49 * (q[:, 2] + q[:, 4]) + 55 * q[:, 3]
    w0, w1 = np.array([18, 34, 49, 55]), np.array([49, 34, 18, 0])     # 0x37312212, 0x00122231 (byte 0 first)
    assert (ref == (q[:, :4] * w0).sum(1) + (q[:, 4:] * w1).sum(1)).all()
    assert ref.max() <= 65535
    a1 = rng.integers(0, 2049, 100_000)
    a0 = 2048 - a1
    p = rng.integers(0, 256, (100_000, 2))
    H = p[:, 0] * a0 + p[:, 1] * a1
    assert H.max() < (1 << 24) and (H >> 4).max() <= 32640   # 24-bit multiply operands of the vertical step


def test_nms_threshold_shortcut():
    """survivor at iniTh: s >= iniTh and s > every neighbour with score >= iniTh  <=>  s >= iniTh and s > every neighbour."""
    rng = np.random.default_rng(11)
    s = rng.integers(0, 256, 200_000)
    nb = rng.integers(0, 256, (200_000, 8))
    nb[rng.integers(0, 2, nb.shape) == 0] = 0                # many empty neighbours, as in a real score map
    for ini in (1, 7, 20, 200):
        long_form = (s >= ini) & (s > np.where(nb >= ini, nb, 0).max(1))
        short_form = (s >= ini) & (s > nb.max(1))
        assert (long_form == short_form).all()


UMAX15 = [15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3]   # src/ORBextractor.cc:455-469 for HALF_PATCH_SIZE = 15


def test_centroid_on_masked_dwords_equals_ic_angle_sums():
    """describe_window's moments: 248 (row, dword) items, byte masks from umax, m10 = <bytes, column index> - 15 * sum, m01 = v * sum."""
    rng = np.random.default_rng(5)
    for _ in range(50):
        patch = rng.integers(0, 256, (31, 32)).astype(np.int64)      # rows v = -15..15, columns u + 15 = 0..31 (column 31 is outside)
        m10 = m01 = 0
        for v in range(-15, 16):                                     # IC_Angle (:83-110): all (u, v) with |u| <= umax[|v|]
            for u in range(-UMAX15[abs(v)], UMAX15[abs(v)] + 1):
                m10 += u * patch[v + 15, u + 15]
                m01 += v * patch[v + 15, u + 15]
        wsum = ssum = vsum = 0
        for item in range(256):
            row, j = item >> 3, item & 7
            if row >= 31:
                continue                                             # the kernel's table holds zero masks there
            v = row - 15
            px = [patch[row, 4 * j + k] if (4 * j + k - 15 <= 15 and abs(4 * j + k - 15) <= UMAX15[abs(v)]) else 0 for k in range(4)]
            sm = sum(px)
            wsum += sum((4 * j + k) * px[k] for k in range(4))
            ssum += sm
            vsum += v * sm
        assert (wsum - 15 * ssum, vsum) == (m10, m01)


def test_quad_record_round_trip():
    """k_fast_quads: record = pol bytes | y << 2 | q << 10; the expansion rebuilds (y << 8 | 4q << 2) with two mask-shifts."""
    for y in range(0, 59):
        for q in range(0, 15):
            for pb in (0x00000001, 0x02000000, 0x01020102, 0x00020000):
                rec = pb | (y << 2) | (q << 10)
                assert rec & 0x03030303 == pb
                yx = ((rec & 0xFC) << 6) | ((rec & 0x3C00) >> 6)
                assert yx == (y << 8) | ((4 * q) << 2)


def _dpp(v, ctrl, row_mask, old=0):
    """update_dpp(old, v, ctrl, row_mask, 0xf, bound_ctrl=false) on a 64-lane array for the controls wave_ops.h uses."""
    out = np.full(64, old, dtype=np.int64)
    for lane in range(64):
        row, i = lane >> 4, lane & 15
        if not (row_mask >> row) & 1:
            out[lane] = old
            continue
        if 0x111 <= ctrl <= 0x11F:                                   # row_shr:n
            n = ctrl - 0x110
            out[lane] = v[lane - n] if i >= n else old
        elif ctrl == 0x142:                                          # row_bcast:15 -> lane 15 of the previous row
            out[lane] = v[(row - 1) * 16 + 15] if row >= 1 else old
        elif ctrl == 0x143:                                          # row_bcast:31 -> lane 31 into rows 2 and 3
            out[lane] = v[31] if row >= 2 else old
        else:
            raise ValueError(ctrl)
    return out


def test_dpp_inclusive_scan_sequence():
    rng = np.random.default_rng(9)
    for _ in range(20):
        x = rng.integers(0, 1000, 64).astype(np.int64)
        v = x.copy()
        for ctrl, mask in ((0x111, 0xF), (0x112, 0xF), (0x114, 0xF), (0x118, 0xF), (0x142, 0xA), (0x143, 0xC)):
            v = v + _dpp(v, ctrl, mask)
        assert (v == np.cumsum(x)).all()


def test_row_bytes8_window_never_leaves_the_row():
    """SparseImgAlign's 8-byte row load: start min(c0, w - 8), shift c0 - start <= 3, the needed bytes are inside the loaded eight."""
    for w in range(8, 80):
        for u in range(3, w - 3):                                    # border = 3: u - 3 >= 0, u + 3 < w
            for c0, need in ((u - 3, 7), (u - 2, 5)):                # precompute (7 bytes); a 5-byte window one column further in
                s0 = min(c0, w - 8)
                sh = c0 - s0
                assert 0 <= s0 and s0 + 8 <= w and 0 <= sh <= 3 and sh + need <= 8


def test_packed_triple_scan_fields():
    """k_octree's block_scan_array3: three exclusive scans as 21-bit fields of one 64-bit running sum (no carries between fields while every
    total stays below 2^21)."""
    rng = np.random.default_rng(21)
    for n in (1, 7, 218, 1023, 2500):
        a, b, c = (rng.integers(0, 5, n).astype(np.uint64), rng.integers(0, 5, n).astype(np.uint64), rng.integers(0, 2, n).astype(np.uint64))
        packed = a | (b << np.uint64(21)) | (c << np.uint64(42))
        excl = np.concatenate([[np.uint64(0)], np.cumsum(packed)[:-1]]).astype(np.uint64)
        m = np.uint64(0x1FFFFF)
        assert ((excl & m) == np.concatenate([[0], np.cumsum(a)[:-1]])).all()
        assert (((excl >> np.uint64(21)) & m) == np.concatenate([[0], np.cumsum(b)[:-1]])).all()
        assert ((excl >> np.uint64(42)) == np.concatenate([[0], np.cumsum(c)[:-1]])).all()
    assert 4 * 400_000 < (1 << 21)          # children per pass <= candidates per level; 4K frames stay far below the field width


def test_pyramid_perm_selectors_and_dot2():
    """k_pyr_resize_tiled: per column the (left, right) source pixels are picked out of eight shifted bytes by one v_perm_b32 selector
    (byte index | 0x0c = zero), then one 16-bit dot product with (alpha0, alpha1)."""
    rng = np.random.default_rng(13)
    for _ in range(2000):
        row = rng.integers(0, 256, 64).astype(np.int64)
        scale = rng.uniform(1.0, 1.27)
        x0 = int(rng.integers(0, 40))
        sx = [int(np.floor((x0 + k + 0.5) * scale - 0.5)) for k in range(4)]
        lx0 = sx[0]
        e = row[lx0:lx0 + 8]                                # the two dwords after v_alignbyte_b32: bytes lx0 .. lx0 + 7
        for k in range(4):
            d, d1 = sx[k] - lx0, sx[k] + 1 - lx0
            assert 0 <= d <= 4 and d1 <= 5                  # scale < 1.28: everything inside the eight bytes
            sel = d | 0x0C00 | (d1 << 16) | 0x0C000000
            picked = [e[(sel >> (8 * b)) & 0xFF] if ((sel >> (8 * b)) & 0xFF) < 8 else 0 for b in range(4)]
            pair = picked[0] | (picked[1] << 8) | (picked[2] << 16) | (picked[3] << 24)
            a1 = int(rng.integers(0, 2049))
            a0 = 2048 - a1
            H = (pair & 0xFFFF) * a0 + (pair >> 16) * a1    # v_dot2_u32_u16
            assert H == row[sx[k]] * a0 + row[sx[k] + 1] * a1


def test_bilinear_weights_fp32():
    """align_kernels.hip forms the bilinear weights of SparseImgAlign in fp32 where the reference (src/SparseImageAlign.cc:93-96,182-185)
    multiplies in double and rounds to float: for pixel coordinates >= 3 (the patch border) the fractions are multiples of 2^-22, so
    1 - s is exact in fp32 and both forms round the same exact product once."""
    rng = np.random.default_rng(5)
    one = np.float32(1)
    for lo, hi in ((3, 8), (3, 64), (3, 4096)):
        u = rng.uniform(lo, hi, 400_000).astype(np.float32)
        v = rng.uniform(lo, hi, 400_000).astype(np.float32)
        u[:64] = np.nextafter(np.floor(u[:64]) + one, np.float32(0))   # fractions next to 1
        v[:64] = np.floor(v[:64])                                       # and exactly 0
        su, sv = (u - np.floor(u)).astype(np.float32), (v - np.floor(v)).astype(np.float32)
        sud, svd = su.astype(np.float64), sv.astype(np.float64)
        ref = [((1.0 - sud) * (1.0 - svd)).astype(np.float32), (sud * (1.0 - svd)).astype(np.float32), ((1.0 - sud) * svd).astype(np.float32)]
        got = [(one - su) * (one - sv), su * (one - sv), (one - su) * sv]
        for r, g in zip(ref, got):
            assert g.dtype == np.float32 and np.array_equal(r, g)


def _sequential_matcher(lists, blocking, two, accept):
    """The reference's loop over the queries, in order (SearchByProjection(Cur, Last) / (F, MapPoints), src/ORBmatcher.cc:43-126, 1218-1350) on
    abstract inputs: lists[q] = the candidate keypoints of query q in (distance, visiting order) order; a keypoint taken by an EARLIER blocking
    query (a MapPoint with observations) is skipped; `two`: best and runner-up feed an accept rule, else the first free candidate is taken."""
    taken, pick = set(), []
    for q, cand in enumerate(lists):
        free = [c for c in cand if c not in taken]
        p = -1
        if two:
            if free and accept(q, free[0], free[1] if len(free) > 1 else -1):
                p = free[0]
        elif free:
            p = free[0]
        pick.append(p)
        if p >= 0 and blocking[q]:
            taken.add(p)
    return pick


def _fixpoint_matcher(lists, blocking, two, accept, depth, max_rounds=1000):
    """k_match_last's block-wide form: every query holds a pick; claim[e] = lowest blocking query that picks e; everybody re-picks among the entries
    of his list with claim[e] >= q; lists are known `depth` entries at a time and extended by eight when they run out (scan_after)."""
    n = len(lists)
    known = [min(depth, len(c)) for c in lists]
    pick = [-1] * n
    rounds = 0
    while True:
        rounds += 1
        assert rounds <= max_rounds
        claim = {}
        for q in range(n):
            if pick[q] >= 0 and blocking[q]:
                claim[pick[q]] = min(claim.get(pick[q], n), q)
        changed, extend = False, []
        for q in range(n):
            free = [c for c in lists[q][:known[q]] if claim.get(c, n) >= q]
            need = 2 if two else 1
            if len(free) < need and known[q] < len(lists[q]):
                new = -2                       # list exhausted before the picks are complete: wait for the extension, take nothing
                extend.append(q)
            elif two:
                new = free[0] if free and accept(q, free[0], free[1] if len(free) > 1 else -1) else -1
            else:
                new = free[0] if free else -1
            if new != pick[q]:
                pick[q] = new
                changed = True
        for q in extend:
            known[q] = min(known[q] + 8, len(lists[q]))
        if not changed and not extend:
            return pick, rounds


def test_matcher_fixpoint_equals_the_sequential_loop():
    """The block-wide fixpoint of k_match_last (match_kernels.hip) against the sequential loop it replaces, on random instances: dense conflicts
    (many queries, few keypoints), MapPoints without observations (non-blocking takers), lists that run out and are extended, both modes."""
    rng = np.random.default_rng(2024)
    worst = 0
    for trial in range(300):
        n = int(rng.integers(1, 120))
        m = int(rng.integers(1, 60 if trial % 3 else 400))
        lists = []
        for _ in range(n):
            k = int(rng.integers(0, min(m, 30) + 1))
            lists.append([int(x) for x in rng.choice(m, size=k, replace=False)])
        blocking = [bool(b) for b in rng.uniform(size=n) < rng.choice([0.5, 0.9, 1.0])]
        table = rng.uniform(size=(n, m + 1)) < 0.8      # an arbitrary accept rule of (query, best, runner-up)
        accept = lambda q, b1, b2: bool(table[q, b1] ^ (b2 >= 0 and table[q, b2] and (b1 + b2) % 3 == 0))
        for two, depth in ((False, 4), (True, 8)):
            want = _sequential_matcher(lists, blocking, two, accept)
            got, rounds = _fixpoint_matcher(lists, blocking, two, accept, depth)
            assert [p if p >= 0 else -1 for p in got] == want, (trial, two)
            worst = max(worst, rounds)
    assert worst <= 121         # never more rounds than queries + 1 (each round settles at least the next query)
