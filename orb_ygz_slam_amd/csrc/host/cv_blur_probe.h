// cv_blur_probe.h -- which generation of cv::GaussianBlur's 8-bit arithmetic does the OpenCV this process is linked against implement?
// (product code, host side, no OpenCV in this header.)
//
// The descriptors of src/ORBextractor.cc:1010 / :1083 sample GaussianBlur(7x7, sigma 2, BORDER_REFLECT_101) of an 8-bit image, and that is the one
// primitive of the hot path whose integers changed between the OpenCV versions the reference can be built against (CMakeLists.txt:40-46): the device
// follows one of three definitions (ygzf_cv_mode, include/ygzf.h).  VERDICT r5: "a user with OpenCV 3.4.11+ must set sCvMode = 2 or get different
// descriptors -- nothing detects it".  This does: a 16 x 12 probe image whose blur differs between the three generations -- one column is an exact
// rounding tie of the legacy kernel (half-to-even in the SSE2 column body, half-up without it), the Q8.8 kernel of >= 3.4.11 differs on most pixels --
// is blurred by the caller's OpenCV and classified by comparing with the three integer models below (the arithmetic of oracle/oracle_cvprims.cpp B4,
// restated; tests/test_cv_blur_probe.py holds the models against the oracle's and checks that each generation is recognised).
// ygz::ORBextractor runs it once per process when sCvMode is left at "detect" (host/ORBextractor.cc).
#ifndef YGZF_CV_BLUR_PROBE_H
#define YGZF_CV_BLUR_PROBE_H
#include <stdint.h>
#include <string.h>

namespace ygzf_host {

constexpr int kBlurProbeW = 16, kBlurProbeH = 12;

// vertically constant rows: columns 0..6 = a tuple p with <(18,34,49,55,49,34,18), p> = 32768, so that column 3 blurs to the exact tie
// 257 * 32768 = 128.5 * 65536 under the legacy kernel; the rest of the row is texture.  Rows 8.. carry a second texture (the models are 2-D).
inline void blur_probe_image(uint8_t *img) {
    static const uint8_t rowA[kBlurProbeW] = {105, 157, 150, 119, 131, 96, 109, 31, 222, 64, 190, 7, 143, 250, 88, 171};
    static const uint8_t rowB[kBlurProbeW] = {12, 240, 77, 199, 34, 160, 91, 215, 53, 128, 249, 3, 180, 66, 137, 101};
    for (int y = 0; y < kBlurProbeH; y++)
        for (int x = 0; x < kBlurProbeW; x++) img[y * kBlurProbeW + x] = y < 8 ? rowA[x] : rowB[x];
}

inline int blur_probe_reflect101(int i, int n) { return i < 0 ? -i : i >= n ? 2 * n - 2 - i : i; }

// mode: 0 = OpenCV 2.4 / 3.0-3.3 on x86 (SSE2 column body: exact ties to even in columns [0, w & ~3)), 1 = the same kernel, half-up everywhere,
// 2 = OpenCV >= 3.4.11 / 4.x (Q8.8 kernel {18,34,48,56,48,34,18})
inline void blur_probe_expected(int mode, uint8_t *out) {
    static const int legacy[7] = {18, 34, 49, 55, 49, 34, 18}, cv4[7] = {18, 34, 48, 56, 48, 34, 18};
    const int *k = mode == 2 ? cv4 : legacy;
    uint8_t img[kBlurProbeW * kBlurProbeH];
    blur_probe_image(img);
    int rows[kBlurProbeW * kBlurProbeH];
    for (int y = 0; y < kBlurProbeH; y++)
        for (int x = 0; x < kBlurProbeW; x++) {
            int s = 0;
            for (int t = -3; t <= 3; t++) s += k[t + 3] * img[y * kBlurProbeW + blur_probe_reflect101(x + t, kBlurProbeW)];
            rows[y * kBlurProbeW + x] = s;
        }
    for (int y = 0; y < kBlurProbeH; y++)
        for (int x = 0; x < kBlurProbeW; x++) {
            int s = 0;
            for (int t = -3; t <= 3; t++) s += k[t + 3] * rows[blur_probe_reflect101(y + t, kBlurProbeH) * kBlurProbeW + x];
            int q = (s + 32768) >> 16;
            if (mode == 0 && (s & 0xFFFF) == 0x8000 && x < (kBlurProbeW & ~3)) q &= ~1;
            out[y * kBlurProbeW + x] = (uint8_t) (q < 0 ? 0 : q > 255 ? 255 : q);
        }
}

// the generation whose model equals `blurred` (kBlurProbeW x kBlurProbeH, tight rows) byte for byte; -1: none of the three
inline int blur_probe_classify(const uint8_t *blurred) {
    for (int mode = 0; mode < 3; mode++) {
        uint8_t want[kBlurProbeW * kBlurProbeH];
        blur_probe_expected(mode, want);
        bool same = true;
        for (int i = 0; i < kBlurProbeW * kBlurProbeH && same; i++) same = want[i] == blurred[i];
        if (same) return mode;
    }
    return -1;
}

#ifdef YGZF_BLUR_PROBE_WITH_CV
// the probe through the cv:: namespace in scope (include the OpenCV headers first): the very call of src/ORBextractor.cc:1010 on the probe image
inline int blur_probe_run_opencv() {
    uint8_t img[kBlurProbeW * kBlurProbeH], out[kBlurProbeW * kBlurProbeH];
    blur_probe_image(img);
    cv::Mat src(kBlurProbeH, kBlurProbeW, CV_8UC1), dst;
    for (int y = 0; y < kBlurProbeH; y++) memcpy(src.ptr(y), img + y * kBlurProbeW, kBlurProbeW);
    cv::GaussianBlur(src, dst, cv::Size(7, 7), 2, 2, cv::BORDER_REFLECT_101);
    if (dst.rows != kBlurProbeH || dst.cols != kBlurProbeW) return -1;
    for (int y = 0; y < kBlurProbeH; y++) memcpy(out + y * kBlurProbeW, dst.ptr(y), kBlurProbeW);
    return blur_probe_classify(out);
}
#endif

}  // namespace ygzf_host
#endif
