// ref_fast9_capi.cpp -- test infrastructure: flat C shim around the REFERENCE's own FAST-9 decision tree,
// Thirdparty/fast/include/fast/corner_9.h:1 (`is_corner_9<fast::Less>` / `is_corner_9<fast::Greater>` over the comparison policies of
// faster_corner_utilities.h:19-41), compiled from where it lies under /root/reference by oracle/Makefile into oracle/_ref/libfast_ref.so
// (the reference sources are never copied into this repository).
//
// What it pins: the PREDICATE of SURVEY a-3b -- "9 contiguous of the 16 ring pixels all > p + t or all < p - t, strict" -- that
// cv::FAST(..., 9/16) implements and that the oracle's fast9() restates from recollection of OpenCV; and, through the predicate's
// monotonicity in t, the SCORE ("the largest threshold for which the pixel is still a corner", what cv's cornerScore<16> computes).
// What it cannot pin: cv::FAST's 3x3 non-maximum suppression (strict >, row-buffered) -- the reference holds no code for that.
#include <cstddef>
#include <cstdint>

#include <fast/faster_corner_utilities.h>
#include <fast/corner_9.h>

extern "C" {

// flags[y * w + x] = 1 where the reference's tree says "corner" at `barrier`, for x in [3, w-3), y in [3, h-3); 0 elsewhere.  Returns the count.
int ref_fast9_corners(const uint8_t *img, int w, int h, int stride, int barrier, uint8_t *flags) {
    int n = 0;
    for (int i = 0; i < w * h; i++) flags[i] = 0;
    for (int y = 3; y < h - 3; y++)
        for (int x = 3; x < w - 3; x++) {
            const unsigned char *p = img + (std::size_t) y * stride + x;
            if (is_corner_9<fast::Less>(p, stride, (short) barrier) || is_corner_9<fast::Greater>(p, stride, (short) barrier)) {
                flags[(std::size_t) y * w + x] = 1;
                n++;
            }
        }
    return n;
}

// the largest barrier in [0, 255] at which (x, y) is a corner by the reference's tree, -1 if it is none at barrier 0 (the predicate is monotone
// in the barrier: a ring pixel beyond p +- t is beyond p +- t' for every t' < t)
void ref_fast9_max_barrier(const uint8_t *img, int stride, const int *xs, const int *ys, int n, int *out) {
    for (int i = 0; i < n; i++) {
        const unsigned char *p = img + (std::size_t) ys[i] * stride + xs[i];
        int lo = -1, hi = 255;           // corner at lo (or lo == -1), unknown in (lo, hi]
        while (lo < hi) {
            const int m = (lo + hi + 1) >> 1;
            if (is_corner_9<fast::Less>(p, stride, (short) m) || is_corner_9<fast::Greater>(p, stride, (short) m)) lo = m;
            else hi = m - 1;
        }
        out[i] = lo;
    }
}
}
