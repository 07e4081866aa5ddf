"""Shared generators for the GaussianBlur-mode tests (CPU tier: tests/test_oracle_blur_modes.py, GPU tier: tests/test_gpu_blur_modes.py)."""
import numpy as np

LEGACY = np.array([18, 34, 49, 55, 49, 34, 18], np.int64)
CV4 = np.array([18, 34, 48, 56, 48, 34, 18], np.int64)


def blur_model(img, mode):
    """Integer model of the three cv::GaussianBlur(7x7, sigma 2, REFLECT_101) definitions: the exact sum of the separable fixed-point
    filter, (sum + 2^15) >> 16, and -- mode 0 -- exact ties rounded to even inside the SSE2 body [0, w & ~3)."""
    k = CV4 if mode == 2 else LEGACY
    h, w = img.shape
    p = np.pad(img.astype(np.int64), 3, mode="reflect")
    rows = sum(k[i] * p[:, i:i + w] for i in range(7))
    s = sum(k[i] * rows[i:i + h, :] for i in range(7))
    q = (s + 32768) >> 16
    if mode == 0:
        tie = (s & 0xFFFF) == 0x8000
        tie[:, (w & ~3):] = False
        q = np.where(tie, q & ~1, q)
    return np.clip(q, 0, 255).astype(np.uint8), s


def tie_tuples(rng, n, lo=96, hi=160):
    """n seven-pixel rows p with <LEGACY, p> == 32768: the centre pixel of such a run in a vertically constant image blurs to the exact tie
    257 * 32768 = 128.5 * 65536 (half up: 129, half to even: 128)."""
    out = []
    while len(out) < n:
        p = rng.integers(lo, hi, (200000, 7))
        hit = p[(p * LEGACY).sum(1) == 32768]
        out.extend(hit[: n - len(out)])
    return np.array(out, np.uint8)


def tie_image(seed, w, h, tail_tie=False):
    """Vertically constant image whose columns 7j+3 are exact ties in the legacy modes (and whose other columns land near 128).
    tail_tie: the last four pixels of a row are chosen so that the LAST column (REFLECT_101 window p[w-4..w-1..w-4]) is a tie as well --
    for w % 4 != 0 that column belongs to the scalar tail of the SSE2 column pass."""
    rng = np.random.default_rng(seed)
    t = tie_tuples(rng, (w + 6) // 7)
    row = t.reshape(-1)[:w].copy()
    if tail_tie:
        while True:
            p = rng.integers(96, 160, (200000, 4))     # p[w-4], p[w-3], p[w-2], p[w-1]
            hit = p[(p * np.array([36, 68, 98, 55])).sum(1) == 32768]
            if len(hit):
                row[w - 4:] = hit[0]
                break
    return np.ascontiguousarray(np.broadcast_to(row, (h, w))).astype(np.uint8)
