// oracle_fast10.cpp -- CPU ORACLE (test infrastructure): definition-level restatement of the reference's
// Thirdparty/fast library (Rosten FAST-10/16), which the reference calls at src/ORBextractor.cc:1220-1235,
// :1330-1337 and which the HIP build replaces outright.  Unlike the OpenCV primitives this one IS pinned: the
// reference's own sources compile stand-alone (oracle/_ref/libfast_ref.so) and tests/test_oracle_fast10.py checks
// this restatement against them on the reference's test image (167-corner KAT, Thirdparty/fast/test/test.cpp:52)
// and on random images.
//   detect : Thirdparty/fast/src/faster_corner_10_sse.cpp:14-198 (domain), fast_10.cpp:10-... (decision tree)
//   score  : Thirdparty/fast/src/fast_10_score.cpp:21-3147  == largest barrier at which the pixel is still a corner
//   nonmax : Thirdparty/fast/src/nonmax_3x3.cpp:18-111      suppress if any 8-neighbour corner has score >= own
#include <cstddef>
#include <cstdint>
#include <vector>

namespace ygzo {

static const int kRing10[16][2] = {{0, 3},  {1, 3},   {2, 2},   {3, 1},   {3, 0},  {3, -1}, {2, -2}, {1, -3},
                                   {0, -3}, {-1, -3}, {-2, -2}, {-3, -1}, {-3, 0}, {-3, 1}, {-2, 2}, {-1, 3}};

// max over the 16 arcs of `arc` contiguous ring pixels of the min signed margin, both polarities.
static inline int arc_margin(const uint8_t *p, int stride, int arc) {
    int d[32];
    const int v = p[0];
    for (int k = 0; k < 16; k++) d[k] = d[k + 16] = (int) p[kRing10[k][0] + kRing10[k][1] * stride] - v;
    int best = -256;
    for (int s = 0; s < 16; s++) {
        int mn = 255, mx = -255;
        for (int k = 0; k < arc; k++) {
            mn = d[s + k] < mn ? d[s + k] : mn;
            mx = d[s + k] > mx ? d[s + k] : mx;
        }
        if (mn > best) best = mn;      // all brighter by >= mn
        if (-mx > best) best = -mx;    // all darker by >= -mx
    }
    return best;  // pixel is a corner at barrier b  <=>  best > b
}

extern "C" {

// fast_corner_detect_10_sse2 semantics: width < 22 -> the plain detector, which scans the WHOLE window
// (x in [0,w), y in [0,h), reading 3 px outside it); otherwise x in [3,w-3), y in [3,h-3); nothing if h < 7.
int yo_fast10_detect(const uint8_t *img, int w, int h, int stride, int barrier, short *xy, int cap) {
    int x0 = 3, x1 = w - 3, y0 = 3, y1 = h - 3;
    if (w < 22) { x0 = 0; x1 = w; y0 = 0; y1 = h; }
    else if (h < 7) return 0;
    int n = 0;
    for (int y = y0; y < y1; y++)
        for (int x = x0; x < x1; x++)
            if (arc_margin(img + (std::size_t) y * stride + x, stride, 10) > barrier) {
                if (n < cap) { xy[2 * n] = (short) x; xy[2 * n + 1] = (short) y; }
                n++;
            }
    return n;
}

void yo_fast10_score(const uint8_t *img, int stride, const short *xy, int n, int *scores) {
    for (int i = 0; i < n; i++) scores[i] = arc_margin(img + (std::size_t) xy[2 * i + 1] * stride + xy[2 * i], stride, 10) - 1;
}

int yo_fast_nonmax_3x3(const short *xy, const int *scores, int n, int *out_idx, int cap) {
    if (n < 1) return 0;
    int maxx = 0, maxy = 0;
    for (int i = 0; i < n; i++) {
        if (xy[2 * i] > maxx) maxx = xy[2 * i];
        if (xy[2 * i + 1] > maxy) maxy = xy[2 * i + 1];
    }
    const int W = maxx + 3, H = maxy + 3;
    std::vector<int> map((std::size_t) W * H, -1);  // -1: not a corner
    for (int i = 0; i < n; i++) map[(std::size_t) (xy[2 * i + 1] + 1) * W + xy[2 * i] + 1] = scores[i];
    int m = 0;
    for (int i = 0; i < n; i++) {
        const int x = xy[2 * i] + 1, y = xy[2 * i + 1] + 1, s = scores[i];
        bool keep = true;
        for (int dy = -1; dy <= 1 && keep; dy++)
            for (int dx = -1; dx <= 1; dx++) {
                if (!dx && !dy) continue;
                const int v = map[(std::size_t) (y + dy) * W + x + dx];
                if (v != -1 && v >= s) { keep = false; break; }
            }
        if (keep) {
            if (m < cap) out_idx[m] = i;
            m++;
        }
    }
    return m;
}
}
}  // namespace ygzo
