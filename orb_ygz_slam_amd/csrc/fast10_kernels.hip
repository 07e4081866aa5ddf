// fast10_kernels.hip -- replacement of the reference's Thirdparty/fast library (Rosten FAST-10/16) on gfx950 (product code).
//
//   fast::fast_corner_detect_10_sse2   Thirdparty/fast/src/faster_corner_10_sse.cpp:14-198  (domain, w < 22 -> plain detector
//                                      fast_10.cpp:10-... which scans the whole window and reads 3 px outside it)
//   fast::fast_corner_score_10         Thirdparty/fast/src/fast_10_score.cpp:21-3147  == largest barrier that is still a corner
//   fast::fast_nonmax_3x3              Thirdparty/fast/src/nonmax_3x3.cpp:18-111      suppress if an 8-neighbour corner scores >= own
//   call sites: src/ORBextractor.cc:1220-1235, :1330-1340, :1440-1450 (FAST_KEYPOINT / DSO_KEYPOINT grid paths)
//
// The generated decision trees of libfast are replaced by their definition: a pixel is a corner at barrier b iff 10
// contiguous ring pixels are all > p+b or all < p-b; its score is (max over the 16 arcs of 10 of the min margin) - 1.
// Pinned against the reference's own library (oracle/_ref) incl. the 167-corner known answer on its test image.
#include "fast10_device.h"
#include "kernels.h"

namespace ygzf {

// score map over the window: -1 = not a corner (or outside the detection domain), else the libfast score
__global__ void k_f10_score(const uint8_t *__restrict__ img, int pitch, int x0, int y0, int w, int h, int dx0, int dx1, int dy0, int dy1,
                            int barrier, short *__restrict__ S) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= w) return;
    short s = -1;
    if (x >= dx0 && x < dx1 && y >= dy0 && y < dy1) {
        const int m = arc10_margin(img + (long long) (y0 + y) * pitch + x0 + x, pitch);
        if (m > barrier) s = (short) (m - 1);
    }
    S[(long long) y * w + x] = s;
}

// one wave per row: corners and non-max survivors per row
__global__ void k_f10_count(const short *__restrict__ S, int w, int h, int *__restrict__ rowCnt, int *__restrict__ rowKept) {
    const int y = blockIdx.x, lane = threadIdx.x;
    int nc = 0, nk = 0;
    for (int xb = 0; xb < w; xb += 64) {
        const int x = xb + lane;
        bool corner = false, keep = false;
        if (x < w) {
            const int s = S[(long long) y * w + x];
            corner = s >= 0;
            if (corner) {
                keep = true;
                for (int dy = -1; dy <= 1 && keep; dy++)
                    for (int dx = -1; dx <= 1; dx++) {
                        if (!dx && !dy) continue;
                        const int xx = x + dx, yy = y + dy;
                        if (xx < 0 || xx >= w || yy < 0 || yy >= h) continue;
                        const int n = S[(long long) yy * w + xx];
                        if (n >= 0 && n >= s) { keep = false; break; }
                    }
            }
        }
        nc += __popcll(__ballot(corner));
        nk += __popcll(__ballot(keep));
    }
    if (lane == 0) { rowCnt[y] = nc; rowKept[y] = nk; }
}

// exclusive scan of the two per-row arrays by one thread block (h rows), totals to out[0..1]
__global__ void k_f10_scan(int *rowCnt, int *rowKept, int h, int *totals) {
    __shared__ int s_carry[2];
    if (threadIdx.x == 0) { s_carry[0] = 0; s_carry[1] = 0; }
    __syncthreads();
    for (int base = 0; base < h; base += blockDim.x) {
        const int i = base + threadIdx.x;
        int a = i < h ? rowCnt[i] : 0, b = i < h ? rowKept[i] : 0;
        // Hillis-Steele inside the block through shared memory
        __shared__ int sa[1024], sb[1024];
        sa[threadIdx.x] = a; sb[threadIdx.x] = b;
        __syncthreads();
        for (int d = 1; d < (int) blockDim.x; d <<= 1) {
            int ta = threadIdx.x >= (unsigned) d ? sa[threadIdx.x - d] : 0, tb = threadIdx.x >= (unsigned) d ? sb[threadIdx.x - d] : 0;
            __syncthreads();
            sa[threadIdx.x] += ta; sb[threadIdx.x] += tb;
            __syncthreads();
        }
        if (i < h) { rowCnt[i] = s_carry[0] + sa[threadIdx.x] - a; rowKept[i] = s_carry[1] + sb[threadIdx.x] - b; }
        __syncthreads();
        if (threadIdx.x == blockDim.x - 1) { s_carry[0] += sa[threadIdx.x]; s_carry[1] += sb[threadIdx.x]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { totals[0] = s_carry[0]; totals[1] = s_carry[1]; }
}

// one wave per row: raster-ordered corner list (x, y, score) and the indices of the non-max survivors
__global__ void k_f10_emit(const short *__restrict__ S, int w, int h, const int *__restrict__ rowOff, const int *__restrict__ keptOff,
                           short *__restrict__ xy, int *__restrict__ scores, int *__restrict__ nonmax, int cap) {
    const int y = blockIdx.x, lane = threadIdx.x;
    int oc = rowOff[y], ok = keptOff[y];
    const unsigned long long lt = (1ull << lane) - 1ull;
    for (int xb = 0; xb < w; xb += 64) {
        const int x = xb + lane;
        bool corner = false, keep = false;
        int s = -1;
        if (x < w) {
            s = S[(long long) y * w + x];
            corner = s >= 0;
            if (corner) {
                keep = true;
                for (int dy = -1; dy <= 1 && keep; dy++)
                    for (int dx = -1; dx <= 1; dx++) {
                        if (!dx && !dy) continue;
                        const int xx = x + dx, yy = y + dy;
                        if (xx < 0 || xx >= w || yy < 0 || yy >= h) continue;
                        const int n = S[(long long) yy * w + xx];
                        if (n >= 0 && n >= s) { keep = false; break; }
                    }
            }
        }
        const unsigned long long mc = __ballot(corner), mk = __ballot(keep);
        const int ci = oc + __popcll(mc & lt);
        if (corner && ci < cap) { xy[2 * ci] = (short) x; xy[2 * ci + 1] = (short) y; scores[ci] = s; }
        if (keep) { const int ki = ok + __popcll(mk & lt); if (ki < cap) nonmax[ki] = ci; }
        oc += __popcll(mc);
        ok += __popcll(mk);
    }
}

void launch_fast10(hipStream_t st, const uint8_t *img, int pitch, int x0, int y0, int w, int h, int dx0, int dx1, int dy0, int dy1, int barrier,
                   short *S, int *rowCnt, int *rowKept, int *totals, short *xy, int *scores, int *nonmax, int cap) {
    hipLaunchKernelGGL(k_f10_score, dim3((w + 255) / 256, h), dim3(256), 0, st, img, pitch, x0, y0, w, h, dx0, dx1, dy0, dy1, barrier, S);
    hipLaunchKernelGGL(k_f10_count, dim3(h), dim3(64), 0, st, S, w, h, rowCnt, rowKept);
    hipLaunchKernelGGL(k_f10_scan, dim3(1), dim3(1024), 0, st, rowCnt, rowKept, h, totals);
    hipLaunchKernelGGL(k_f10_emit, dim3(h), dim3(64), 0, st, S, w, h, rowCnt, rowKept, xy, scores, nonmax, cap);
}

}  // namespace ygzf
