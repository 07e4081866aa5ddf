// ref_fast_capi.cpp -- test infrastructure: flat C shim around the REFERENCE's own Thirdparty/fast library
// (compiled from where it lies under /root/reference by oracle/Makefile into oracle/_ref/libfast_ref.so; the
// reference sources are never copied into this repository).  Interface: Thirdparty/fast/include/fast/fast.h:19-29.
#include <fast/fast.h>

#include <cstddef>
#include <cstdint>
#include <vector>

extern "C" {

// which: 0 = fast_corner_detect_10 (plain), 1 = fast_corner_detect_10_sse2
int ref_fast10_detect(int which, const uint8_t *img, int w, int h, int stride, int barrier, short *xy, int cap) {
    std::vector<fast::fast_xy> c;
    if (which == 0) fast::fast_corner_detect_10(img, w, h, stride, (short) barrier, c);
    else fast::fast_corner_detect_10_sse2(img, w, h, stride, (short) barrier, c);
    for (std::size_t i = 0; i < c.size() && (int) i < cap; i++) {
        xy[2 * i] = c[i].x;
        xy[2 * i + 1] = c[i].y;
    }
    return (int) c.size();
}

void ref_fast10_score(const uint8_t *img, int stride, const short *xy, int n, int threshold, int *scores) {
    std::vector<fast::fast_xy> c;
    c.reserve(n);
    for (int i = 0; i < n; i++) c.emplace_back(xy[2 * i], xy[2 * i + 1]);
    std::vector<int> s;
    fast::fast_corner_score_10(img, stride, c, threshold, s);
    for (int i = 0; i < n; i++) scores[i] = s[i];
}

int ref_fast_nonmax_3x3(const short *xy, const int *scores, int n, int *out_idx, int cap) {
    std::vector<fast::fast_xy> c;
    c.reserve(n);
    for (int i = 0; i < n; i++) c.emplace_back(xy[2 * i], xy[2 * i + 1]);
    std::vector<int> s(scores, scores + n), nm;
    fast::fast_nonmax_3x3(c, s, nm);
    for (std::size_t i = 0; i < nm.size() && (int) i < cap; i++) out_idx[i] = nm[i];
    return (int) nm.size();
}
}
