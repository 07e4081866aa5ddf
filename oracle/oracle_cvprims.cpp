// oracle_cvprims.cpp -- CPU ORACLE (test infrastructure): restatements of the OpenCV primitives the reference's
// ORBextractor calls.  OpenCV is NOT vendored in /root/reference and its version is not pinned
// (CMakeLists.txt:40-46: "OpenCV 3.0, else 2.4.3"; README-ORB-SLAM2.md:64: "tested with 2.4.11 and 3.2"), so the
// arithmetic below is the published 2.4.11/3.2 non-IPP integer behaviour restated from the algorithm description
// (SURVEY.md Appendix B).  PARITY UNPINNED: no reference test or fixture covers any of it.
//
// Call sites restated:  cv::resize  src/ORBextractor.cc:1139   cv::GaussianBlur :1010,:1083
//                       cv::FAST    :765,:768                  cv::fastAtan2    :100      cvRound :80,:111,...
#include "ygz_oracle.h"

#include <cfloat>
#include <cmath>
#include <cstdlib>
#include <cstring>

namespace ygzo {

// B6: cvRound = round half to even (SSE2 cvtsd2si under the default rounding mode).
int cv_round(double v) { return (int) std::nearbyint(v); }

static inline short saturate_short_from_float(float v) {
    int i = cv_round((double) v);
    return (short) (i < -32768 ? -32768 : i > 32767 ? 32767 : i);
}

// B5: cv::fastAtan2 -- degree result in [0,360); float arithmetic in source order, no contraction
// (this file is built with -ffp-contract=off).
float fast_atan2_deg(float y, float x) {
    static const float k = (float) (180.0 / 3.14159265358979323846);
    static const float p1 = 0.9997878412794807f * k;
    static const float p3 = -0.3258083974640975f * k;
    static const float p5 = 0.1555786518463281f * k;
    static const float p7 = -0.04432655554792128f * k;
    float ax = std::fabs(x), ay = std::fabs(y);
    float a, c, c2;
    if (ax >= ay) {
        c = ay / (ax + (float) DBL_EPSILON);
        c2 = c * c;
        a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    } else {
        c = ax / (ay + (float) DBL_EPSILON);
        c2 = c * c;
        a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

// cosf/sinf stand-in.  The reference computes  angle = kpt.angle * (float)(CV_PI/180.f);  a = (float)cos(angle),
// b = (float)sin(angle)  with a float argument (src/ORBextractor.cc:103,108-109), i.e. glibc cosf/sinf.  The oracle
// DEFINES them as the float rounding of a double-precision evaluation (quadrant reduction + fdlibm-style kernel
// polynomials, plain * and + only) so that the HIP kernel can run the identical operation sequence; the test-suite
// checks this against libm's cos/sin (double) rounded to float and against cosf/sinf.
void sincos_deg(float angle_deg, float *c_out, float *s_out) {
    static const float factorPI = (float) (3.14159265358979323846 / 180.f);
    const float angle = angle_deg * factorPI;
    const double x = (double) angle;
    static const double TWO_OVER_PI = 6.36619772367581382433e-01;
    static const double PIO2_1 = 1.57079632673412561417e+00;   // first 33 bits of pi/2
    static const double PIO2_1T = 6.07710050650619224932e-11;  // pi/2 - PIO2_1
    const double kd = std::floor(x * TWO_OVER_PI + 0.5);
    const int k = (int) kd;
    const double r = (x - kd * PIO2_1) - kd * PIO2_1T;
    const double z = r * r;
    static const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03,
                        S3 = -1.98412698298579493134e-04, S4 = 2.75573137070700676789e-06,
                        S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    static const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03,
                        C3 = 2.48015872894767294178e-05, C4 = -2.75573143513906633035e-07,
                        C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    const double sr = r + (z * r) * (S1 + z * (S2 + z * (S3 + z * (S4 + z * (S5 + z * S6)))));
    const double rc = z * (C1 + z * (C2 + z * (C3 + z * (C4 + z * (C5 + z * C6)))));
    const double cr = 1.0 - (0.5 * z - z * rc);
    double s, c;
    switch (k & 3) {
        case 0: s = sr; c = cr; break;
        case 1: s = cr; c = -sr; break;
        case 2: s = -sr; c = -cr; break;
        default: s = -cr; c = sr; break;
    }
    *c_out = (float) c;
    *s_out = (float) s;
}

// B1: cv::resize(src, dst, dsize, 0, 0, INTER_LINEAR) for CV_8UC1 (fixed point, INTER_RESIZE_COEF_BITS = 11).
// When both scale factors are exactly 2 OpenCV reroutes INTER_LINEAR to the 2x2 area-average fast path.
void resize_linear_u8(const Image &src, Image &dst) {
    const int sw = src.w, sh = src.h, dw = dst.w, dh = dst.h;
    const double inv_scale_x = (double) dw / sw, inv_scale_y = (double) dh / sh;
    const double scale_x = 1. / inv_scale_x, scale_y = 1. / inv_scale_y;
    if (sw == dw * 2 && sh == dh * 2) {  // INTER_AREA fast path, ResizeAreaFastVec: (a+b+c+d+2)>>2
        for (int y = 0; y < dh; y++)
            for (int x = 0; x < dw; x++) {
                int s = src.at(2 * y, 2 * x) + src.at(2 * y, 2 * x + 1) + src.at(2 * y + 1, 2 * x) +
                        src.at(2 * y + 1, 2 * x + 1);
                dst.d[(size_t) y * dw + x] = (uint8_t) ((s + 2) >> 2);
            }
        return;
    }
    std::vector<int> xofs(dw), yofs(dh);
    std::vector<short> ialpha(2 * (size_t) dw), ibeta(2 * (size_t) dh);
    for (int dx = 0; dx < dw; dx++) {
        float fx = (float) ((dx + 0.5) * scale_x - 0.5);
        int sx = (int) std::floor(fx);
        fx -= sx;
        if (sx < 0) { fx = 0; sx = 0; }
        if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
        xofs[dx] = sx;
        ialpha[dx * 2] = saturate_short_from_float((1.f - fx) * 2048);
        ialpha[dx * 2 + 1] = saturate_short_from_float(fx * 2048);
    }
    for (int dy = 0; dy < dh; dy++) {
        float fy = (float) ((dy + 0.5) * scale_y - 0.5);
        int sy = (int) std::floor(fy);
        fy -= sy;
        yofs[dy] = sy;
        ibeta[dy * 2] = saturate_short_from_float((1.f - fy) * 2048);
        ibeta[dy * 2 + 1] = saturate_short_from_float(fy * 2048);
    }
    std::vector<int> H0(dw), H1(dw);
    for (int dy = 0; dy < dh; dy++) {
        int sy0 = yofs[dy], sy1 = yofs[dy] + 1;
        sy0 = sy0 < 0 ? 0 : (sy0 >= sh ? sh - 1 : sy0);  // clip(sy, 0, ssize.height)
        sy1 = sy1 < 0 ? 0 : (sy1 >= sh ? sh - 1 : sy1);
        const uint8_t *S0 = &src.d[(size_t) sy0 * sw], *S1 = &src.d[(size_t) sy1 * sw];
        for (int dx = 0; dx < dw; dx++) {
            const int sx = xofs[dx];
            const int sx1 = sx + 1 < sw ? sx + 1 : sw - 1;  // the a1 == 0 tap of the clamped column
            H0[dx] = S0[sx] * ialpha[dx * 2] + S0[sx1] * ialpha[dx * 2 + 1];
            H1[dx] = S1[sx] * ialpha[dx * 2] + S1[sx1] * ialpha[dx * 2 + 1];
        }
        const int b0 = ibeta[dy * 2], b1 = ibeta[dy * 2 + 1];
        uint8_t *D = &dst.d[(size_t) dy * dw];
        for (int dx = 0; dx < dw; dx++)
            D[dx] = (uint8_t) ((((b0 * (H0[dx] >> 4)) >> 16) + ((b1 * (H1[dx] >> 4)) >> 16) + 2) >> 2);
    }
}

static inline int reflect101(int i, int n) {
    if (n == 1) return 0;
    while (i < 0 || i >= n) {
        if (i < 0) i = -i;
        else i = 2 * n - 2 - i;
    }
    return i;
}

// B4: cv::GaussianBlur(src, dst, Size(7,7), 2, 2, BORDER_REFLECT_101), CV_8UC1 (call sites src/ORBextractor.cc:1010, :1083).
// OpenCV is neither vendored nor version-pinned by the reference (find_package(OpenCV 3.0) else 2.4.3, CMakeLists.txt:40-46), and the
// 8-bit Gaussian is the one primitive of this path whose arithmetic changed between the versions a user can build against.  Three
// recalled definitions, selected process-wide with set_cv_mode():
//   CV_MODE_LEGACY_SSE2 (0, default) OpenCV 2.4.x / 3.0 - 3.3 on x86 -- what the reference's own build produces (-march=native,
//       CMakeLists.txt:13-14; tested versions 2.4.11 / 3.2, README-ORB-SLAM2.md:64).  Separable fixed-point filter: the float kernel
//       exp(-x^2/8) normalised, both 1-D kernels converted to int32 with cvRound(k*256) = {18,34,49,55,49,34,18} (sum 257); row pass
//       u8 -> int32; column pass through SymmColumnVec_32s8u: the int kernel is turned back into floats k/65536, the row sums are
//       converted to float, s = c*f0, then s += (row[+k] + row[-k]) * fk for k = 1..3 (mulps / addps, no FMA), _mm_cvtps_epi32
//       (ROUND HALF TO EVEN) and unsigned saturation.  The vector body covers columns [0, width & ~3); the last width % 4 columns go
//       through the scalar FixedPtCastEx tail, (sum + 2^15) >> 16.  Every float operation is exact here (numerators stay below
//       2^24 until the result is >= 256, which saturates either way), so the two roundings differ only on exact ties,
//       sum mod 65536 == 32768.
//   CV_MODE_LEGACY_INT (1) the same kernel without the SSE2 column body (non-x86 / SIMD-less builds): (sum + 2^15) >> 16 everywhere.
//   CV_MODE_CV4 (2) OpenCV >= 3.4.11 / 4.x "bit-exact" 8-bit path: Q8.8 kernel from getGaussianKernelFixedPoint_ED (error diffusion,
//       centre = 256 - rest) = {18,34,48,56,48,34,18} (sum 256); horizontal pass into ufixedpoint16 (exact, <= 255*256), vertical pass
//       in Q16.16 with (sum + 2^15) >> 16.  SIMD and scalar forms agree by design.
// IPP / OpenCL dispatch of a particular OpenCV binary is outside all three.
static int g_cv_mode = CV_MODE_LEGACY_SSE2;
void set_cv_mode(int mode) { g_cv_mode = mode; }
int get_cv_mode() { return g_cv_mode; }

}  // namespace ygzo
// exported by every library that links this file (the oracle and the oracle/_ref builds of the reference's own sources)
extern "C" void yo_set_cv_mode(int mode) { ygzo::set_cv_mode(mode); }
extern "C" int yo_get_cv_mode() { return ygzo::get_cv_mode(); }
namespace ygzo {

void gaussian_kernel7_s2(int mode, int kq[7]) {
    if (mode == CV_MODE_CV4) {
        // getGaussianKernelFixedPoint_ED: normalised double kernel, v_i = round(k_i * 256 + err), err carried, centre takes the remainder
        double k[7], sum = 0;
        for (int i = 0; i < 7; i++) { const double x = i - 3.0; k[i] = std::exp(-0.5 / (2.0 * 2.0) * x * x); sum += k[i]; }
        double err = 0;
        int acc = 0;
        for (int i = 0; i < 3; i++) {
            const double adj = k[i] / sum * 256.0 + err;
            const int v = cv_round(adj);
            err = adj - v;
            kq[i] = kq[6 - i] = v;
            acc += 2 * v;
        }
        kq[3] = 256 - acc;
        return;
    }
    float cf[7];
    double sum = 0;
    for (int i = 0; i < 7; i++) {
        double x = i - 3.0;
        cf[i] = (float) std::exp(-0.5 / (2.0 * 2.0) * x * x);
        sum += cf[i];
    }
    sum = 1. / sum;
    for (int i = 0; i < 7; i++) {
        cf[i] = (float) (cf[i] * sum);
        kq[i] = cv_round((double) cf[i] * 256.0);
    }
}

void gaussian_blur7_s2_u8(const Image &src, Image &dst) { gaussian_blur7_s2_u8(src, dst, g_cv_mode); }

void gaussian_blur7_s2_u8(const Image &src, Image &dst, int mode) {
    int kq[7];
    gaussian_kernel7_s2(mode, kq);
    const int w = src.w, h = src.h;
    dst.w = w;
    dst.h = h;
    dst.d.resize((size_t) w * h);
    std::vector<int> rows((size_t) w * h);
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            int s = 0;
            for (int k = -3; k <= 3; k++) s += kq[k + 3] * src.at(y, reflect101(x + k, w));
            rows[(size_t) y * w + x] = s;
        }
    const int wvec = mode == CV_MODE_LEGACY_SSE2 ? (w & ~3) : 0;   // columns of the SSE2 body (16- and 4-wide loops)
    float fk[4];
    for (int k = 0; k < 4; k++) fk[k] = (float) kq[3 + k] * (1.f / 65536.f);   // kernel.convertTo(CV_32F, 1. / (1 << 16)): exact
    for (int y = 0; y < h; y++) {
        const int *R[7];
        for (int k = -3; k <= 3; k++) R[k + 3] = &rows[(size_t) reflect101(y + k, h) * w];
        for (int x = 0; x < w; x++) {
            int v;
            if (x < wvec) {
                float s = (float) R[3][x] * fk[0];                               // mulps (+ delta 0)
                for (int k = 1; k <= 3; k++) s = s + (float) (R[3 + k][x] + R[3 - k][x]) * fk[k];   // paddd, cvtdq2ps, mulps, addps
                v = (int) std::nearbyintf(s);                                    // cvtps2dq: round half to even
            } else {
                int s = 0;
                for (int k = -3; k <= 3; k++) s += kq[k + 3] * R[k + 3][x];
                v = (s + 32768) >> 16;
            }
            dst.d[(size_t) y * w + x] = (uint8_t) (v < 0 ? 0 : v > 255 ? 255 : v);
        }
    }
}

// B3: cv::FAST TYPE_9_16.  Ring offsets as in Thirdparty/fast/src/fast_10.cpp:16-33 (same Bresenham circle).
static const int kRing[16][2] = {{0, 3},  {1, 3},   {2, 2},   {3, 1},   {3, 0},  {3, -1}, {2, -2}, {1, -3},
                                 {0, -3}, {-1, -3}, {-2, -2}, {-3, -1}, {-3, 0}, {-3, 1}, {-2, 2}, {-1, 3}};

// cornerScore<16>: largest threshold for which the pixel is still a 9/16 corner, minus nothing:
// returns max over arcs of min |diff| - 1 (>= threshold for a corner).
static int corner_score16(const uint8_t *ptr, const int pixel[25], int threshold) {
    const int K = 8, N = K * 3 + 1;
    int k, v = ptr[0];
    short d[N];
    for (k = 0; k < N; k++) d[k] = (short) (v - ptr[pixel[k]]);
    int a0 = threshold;
    for (k = 0; k < 16; k += 2) {
        int a = std::min((int) d[k + 1], (int) d[k + 2]);
        a = std::min(a, (int) d[k + 3]);
        if (a <= a0) continue;
        a = std::min(a, (int) d[k + 4]);
        a = std::min(a, (int) d[k + 5]);
        a = std::min(a, (int) d[k + 6]);
        a = std::min(a, (int) d[k + 7]);
        a = std::min(a, (int) d[k + 8]);
        a0 = std::max(a0, std::min(a, (int) d[k]));
        a0 = std::max(a0, std::min(a, (int) d[k + 9]));
    }
    int b0 = -a0;
    for (k = 0; k < 16; k += 2) {
        int b = std::max((int) d[k + 1], (int) d[k + 2]);
        b = std::max(b, (int) d[k + 3]);
        b = std::max(b, (int) d[k + 4]);
        b = std::max(b, (int) d[k + 5]);
        if (b >= b0) continue;
        b = std::max(b, (int) d[k + 6]);
        b = std::max(b, (int) d[k + 7]);
        b = std::max(b, (int) d[k + 8]);
        b0 = std::min(b0, std::max(b, (int) d[k]));
        b0 = std::min(b0, std::max(b, (int) d[k + 9]));
    }
    return -b0 - 1;
}

void fast9(const uint8_t *img, int stride, int w, int h, int threshold, bool nonmax, std::vector<FastPt> &out) {
    out.clear();
    const int K = 8, N = 16 + K + 1;
    int pixel[25];
    for (int k = 0; k < 16; k++) pixel[k] = kRing[k][0] + kRing[k][1] * stride;
    for (int k = 16; k < 25; k++) pixel[k] = pixel[k - 16];
    threshold = std::min(std::max(threshold, 0), 255);
    uint8_t threshold_tab[512];
    for (int i = -255; i <= 255; i++) threshold_tab[i + 255] = (uint8_t) (i < -threshold ? 1 : i > threshold ? 2 : 0);
    if (w < 7 || h < 7) return;

    std::vector<uint8_t> score((size_t) w * h, 0);   // 0 outside the [3,w-3)x[3,h-3) detection domain
    std::vector<uint8_t> is_corner((size_t) w * h, 0);
    for (int i = 3; i < h - 3; i++) {
        const uint8_t *ptr = img + (size_t) i * stride + 3;
        for (int j = 3; j < w - 3; j++, ptr++) {
            const int v = ptr[0];
            const uint8_t *tab = &threshold_tab[0] - v + 255;
            int d = tab[ptr[pixel[0]]] | tab[ptr[pixel[8]]];
            if (d == 0) continue;
            d &= tab[ptr[pixel[2]]] | tab[ptr[pixel[10]]];
            d &= tab[ptr[pixel[4]]] | tab[ptr[pixel[12]]];
            d &= tab[ptr[pixel[6]]] | tab[ptr[pixel[14]]];
            if (d == 0) continue;
            d &= tab[ptr[pixel[1]]] | tab[ptr[pixel[9]]];
            d &= tab[ptr[pixel[3]]] | tab[ptr[pixel[11]]];
            d &= tab[ptr[pixel[5]]] | tab[ptr[pixel[13]]];
            d &= tab[ptr[pixel[7]]] | tab[ptr[pixel[15]]];
            bool corner = false;
            if (d & 1) {
                int vt = v - threshold, count = 0;
                for (int k = 0; k < N; k++) {
                    int x = ptr[pixel[k]];
                    if (x < vt) {
                        if (++count > K) { corner = true; break; }
                    } else
                        count = 0;
                }
            }
            if (!corner && (d & 2)) {
                int vt = v + threshold, count = 0;
                for (int k = 0; k < N; k++) {
                    int x = ptr[pixel[k]];
                    if (x > vt) {
                        if (++count > K) { corner = true; break; }
                    } else
                        count = 0;
                }
            }
            if (corner) {
                is_corner[(size_t) i * w + j] = 1;
                score[(size_t) i * w + j] = (uint8_t) corner_score16(ptr, pixel, threshold);
            }
        }
    }
    for (int i = 3; i < h - 3; i++)
        for (int j = 3; j < w - 3; j++) {
            if (!is_corner[(size_t) i * w + j]) continue;
            const int s = score[(size_t) i * w + j];
            if (nonmax) {
                const uint8_t *p = &score[(size_t) (i - 1) * w + j], *c = p + w, *n = c + w;
                if (!(s > c[1] && s > c[-1] && s > p[-1] && s > p[0] && s > p[1] && s > n[-1] && s > n[0] &&
                      s > n[1]))
                    continue;
            }
            out.push_back(FastPt{j, i, s});
        }
}

}  // namespace ygzo
