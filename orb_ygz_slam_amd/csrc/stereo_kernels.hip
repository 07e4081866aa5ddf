// stereo_kernels.hip -- ygz::Frame::ComputeStereoMatches on gfx950 (product code), reference src/Frame.cc:509-682.
//
//   k_stereo_prep    per right keypoint: the row band [floor(y - r), ceil(y + r)], r = 2 * scale[octave]   (:526-538); the records sorted by
//                    the band's first row (bins of 2^binShift rows) so that a left keypoint only meets the records near its row
//   k_stereo_match   one wave per left keypoint: best Hamming match among the right keypoints whose band covers its row (:552-593),
//                    11x11 SAD of the centre-subtracted patches over +-5 px on the keypoint's pyramid level (:596-640), parabola
//                    sub-pixel fit, disparity gates, depth (:646-668)
//   k_stereo_cut     per pair: median of the accepted SADs, matches with SAD >= 1.5 * 1.4 * median are dropped (:672-682)
//
// Every left keypoint is independent (no in-order dependence as in the SearchBy* functions).  The reference's row table is a
// list of right-keypoint indices per image row in ascending index order; scanning the right keypoints directly in index order with
// the band test gives the same candidate sequence, and a (distance << 16 | index) key min-reduced over the wave reproduces the
// strict `dist < bestDist` scan.  convertTo(CV_32F) / "minus centre" / cv::norm(NORM_L1) are integer arithmetic that float holds
// exactly (|sum| <= 121 * 510), evaluated here in int32.  Float expressions keep source order (library built -ffp-contract=off).
#include "kernels.h"
#include "wave_ops.h"

namespace ygzf {

// exclusive prefix sum of a[0..n) in LDS by the whole block (n <= a few thousand); tmp: 17 ints of LDS
__device__ __forceinline__ void block_scan_small(int *a, int n, int *tmp) {
    const int tid = threadIdx.x, nt = blockDim.x, lane = tid & 63, wave = tid >> 6, nw = nt >> 6;
    const int per = (n + nt - 1) / nt;
    const int lo = min(tid * per, n), hi = min(lo + per, n);
    int sum = 0;
    for (int i = lo; i < hi; i++) sum += a[i];
    int incl = sum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int o = __shfl_up(incl, d);
        if (lane >= d) incl += o;
    }
    if (lane == 63) tmp[wave] = incl;
    __syncthreads();
    if (tid == 0) {
        int acc = 0;
        for (int w = 0; w < nw; w++) { const int t = tmp[w]; tmp[w] = acc; acc += t; }
    }
    __syncthreads();
    int run = tmp[wave] + incl - sum;
    for (int i = lo; i < hi; i++) { const int t = a[i]; a[i] = run; run += t; }
    __syncthreads();
}

// One workgroup per pair.  The reference keeps, per image row, the list of right keypoints whose band covers it (:526-538); here the right
// keypoints are counting-sorted by the bin of their band's FIRST row (bins of 2^binShift rows): a left keypoint on row v then scans the bins of
// rows [v - bandMax, v] only (bandMax >= the longest band), a few hundred records instead of every right keypoint of the frame.  The order inside
// a bin is whatever the atomics make it: the matcher's (distance << 16 | index) minimum does not depend on the order of the scan.
constexpr int kStereoPrepBlock = 1024;
constexpr int kStereoMaxBins = 4096;
__global__ __launch_bounds__(kStereoPrepBlock) void k_stereo_prep(StereoArgs A) {
    __shared__ int s_bin[kStereoMaxBins + 1];
    __shared__ int s_tmp[20];
    const int pair = blockIdx.x, tid = threadIdx.x;
    const int nr = (A.cntR ? A.cntR : A.cnt)[(long long) pair * A.cntStride + A.cntOffR];
    const int nb = A.nBins;
    int *binStart = A.binStart + (long long) pair * (kStereoMaxBins + 1);
    for (int b = tid; b <= nb; b += kStereoPrepBlock) s_bin[b] = 0;
    __syncthreads();
    auto make = [&](int i, StereoRec *rec) {
        const ygzf_kp k = (A.keysR ? A.keysR : A.keys)[(long long) pair * A.keyStride + A.keyOffR + i];
        const float r = 2.0f * A.scale[k.octave];
        int maxr = (int) ceilf(k.y + r), minr = (int) floorf(k.y - r);
        minr = max(minr, 0);
        maxr = min(maxr, A.nRows - 1);
        rec->x = k.x;
        rec->band = (unsigned) (minr & 0xFFFF) | ((unsigned) (maxr & 0xFFFF) << 16);
        rec->octave = (maxr >= minr ? (k.octave & 0xFFFF) : 1000) | (i << 16);   // an empty band never matches; index of the keypoint above it
        return min(min(minr, A.nRows - 1) >> A.binShift, nb - 1);
    };
    for (int i = tid; i < nr; i += kStereoPrepBlock) {
        StereoRec rec;
        atomicAdd(&s_bin[make(i, &rec)], 1);
    }
    __syncthreads();
    block_scan_small(s_bin, nb + 1, s_tmp);            // exclusive: s_bin[b] = first sorted position of bin b, s_bin[nb] = nr
    for (int b = tid; b <= nb; b += kStereoPrepBlock) binStart[b] = s_bin[b];
    __syncthreads();
    StereoRec *out = A.rec + (long long) pair * A.recStride;
    for (int i = tid; i < nr; i += kStereoPrepBlock) {
        StereoRec rec;
        const int b = make(i, &rec);
        out[atomicAdd(&s_bin[b], 1)] = rec;
    }
}

__device__ __forceinline__ int s_wave_sum(int v) { return wave_sum(v); }
__device__ __forceinline__ unsigned s_wave_min(unsigned v) { return wave_min_u32(v); }

__global__ __launch_bounds__(256) void k_stereo_match(StereoArgs A) {
    const int pair = blockIdx.y, lane = threadIdx.x & 63;
    const int iL = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nl = A.cnt[(long long) pair * A.cntStride + A.cntOffL], nr = (A.cntR ? A.cntR : A.cnt)[(long long) pair * A.cntStride + A.cntOffR];
    if (iL >= nl) return;
    float *outU = A.uRight + (long long) pair * A.outStride, *outD = A.depth + (long long) pair * A.outStride;
    int *outS = A.sad + (long long) pair * A.outStride;
    float resU = -1.0f, resD = -1.0f;
    int resS = -1;
    const ygzf_kp kL = A.keys[(long long) pair * A.keyStride + A.keyOffL + iL];
    const int levelL = kL.octave;
    const float vL = kL.y, uL = kL.x;
    const float maxD = A.mbf / A.mb;
    const float minU = uL - maxD, maxU = uL - 0.f;
    const int TH_HIGH = 100, thOrbDist = 75;
    bool alive = vL >= 0 && vL < (float) A.nRows && !(maxU < 0);
    unsigned best = ((unsigned) TH_HIGH << 16);   // only distances < TH_HIGH replace the initial best (:565, :583)
    if (alive) {
        const int row = (int) vL;
        const unsigned long long *dL = (const unsigned long long *) (A.desc + ((long long) pair * A.keyStride + A.keyOffL + iL) * 32);
        const unsigned long long q0 = dL[0], q1 = dL[1], q2 = dL[2], q3 = dL[3];
        const StereoRec *rec = A.rec + (long long) pair * A.recStride;
        const uint8_t *descR = (A.descR ? A.descR : A.desc) + ((long long) pair * A.keyStride + A.keyOffR) * 32;
        const int *binStart = A.binStart + (long long) pair * (kStereoMaxBins + 1);
        const int b0 = max(row - A.bandMax, 0) >> A.binShift, b1 = min(row >> A.binShift, A.nBins - 1);
        const int lo = binStart[b0], hi = binStart[b1 + 1];
        for (int j = lo + lane; j < hi; j += 64) {
            const StereoRec rc = rec[j];
            const int minr = (int) (rc.band & 0xFFFFu), maxr = (int) (rc.band >> 16);
            const int oct = rc.octave & 0xFFFF, iR = (int) ((unsigned) rc.octave >> 16);
            if (row < minr || row > maxr) continue;
            if (oct < levelL - 1 || oct > levelL + 1) continue;
            if (!(rc.x >= minU && rc.x <= maxU)) continue;
            const unsigned long long *d = (const unsigned long long *) (descR + (long long) iR * 32);
            const unsigned dist = __popcll(q0 ^ d[0]) + __popcll(q1 ^ d[1]) + __popcll(q2 ^ d[2]) + __popcll(q3 ^ d[3]);
            const unsigned key = (dist << 16) | (unsigned) iR;
            best = key < best ? key : best;
        }
    }
    best = s_wave_min(best);
    const int bestDist = (int) (best >> 16);
    if (alive && bestDist < thOrbDist) {
        const int bestIdxR = (int) (best & 0xFFFFu);
        const float uR0 = (A.keysR ? A.keysR : A.keys)[(long long) pair * A.keyStride + A.keyOffR + bestIdxR].x;
        const float scaleFactor = A.invScale[levelL];
        const float scaleduL = roundf(kL.x * scaleFactor);
        const float scaledvL = roundf(kL.y * scaleFactor);
        const float scaleduR0 = roundf(uR0 * scaleFactor);
        const int w = 5, L = 5;
        const LevelGeom g = A.geom[levelL];
        int pitchL, pitchR;
        const uint8_t *imL = level_ptr(A.fs, g, levelL, A.frame0 + pair * A.frameStep, &pitchL);
        const uint8_t *imR = level_ptr(A.keysR ? A.fsR : A.fs, g, levelL, A.frame0 + pair * A.frameStep + 1, &pitchR);
        const int cxL = (int) scaleduL, cyL = (int) scaledvL, cxR0 = (int) scaleduR0;
        const float iniu = scaleduR0 + L - w, endu = scaleduR0 + L + w + 1;
        // rowRange / colRange outside the level throw in OpenCV (left keys sit >= 16 px inside their level); :618-620
        const bool inside = cxL - w >= 0 && cyL - w >= 0 && cxL + w < g.w && cyL + w < g.h && !(iniu < 0 || endu >= g.w) && cxR0 - L - w >= 0;
        if (inside) {
            // lane -> patch pixels p0 = lane, p1 = lane + 64 (< 121)
            const int p0 = lane, p1 = lane + 64;
            const int r0 = p0 / 11, c0 = p0 - 11 * r0, r1 = p1 / 11, c1 = p1 - 11 * r1;
            const bool has1 = p1 < 121;
            const int cL = imL[(long long) cyL * pitchL + cxL];
            const int a0 = (int) imL[(long long) (cyL - w + r0) * pitchL + cxL - w + c0] - cL;
            const int a1 = has1 ? (int) imL[(long long) (cyL - w + r1) * pitchL + cxL - w + c1] - cL : 0;
            const uint8_t *row0 = imR + (long long) (cyL - w + r0) * pitchR + cxR0 - w + c0;
            const uint8_t *row1 = imR + (long long) (cyL - w + (has1 ? r1 : 0)) * pitchR + cxR0 - w + (has1 ? c1 : 0);
            const uint8_t *rowC = imR + (long long) cyL * pitchR + cxR0;
            int sums[11];
#pragma unroll
            for (int k = 0; k < 11; k++) {
                const int incR = k - L;
                const int cR = rowC[incR];
                int s = abs(a0 - ((int) row0[incR] - cR));
                if (has1) s += abs(a1 - ((int) row1[incR] - cR));
                sums[k] = s_wave_sum(s);
            }
            int bestS = 0x7FFFFFFF, bestinc = 0;
#pragma unroll
            for (int k = 0; k < 11; k++)
                if (sums[k] < bestS) { bestS = sums[k]; bestinc = k - L; }
            if (!(bestinc == -L || bestinc == L)) {
                float dist1 = 0.f, dist2 = 0.f, dist3 = 0.f;
#pragma unroll
                for (int k = 1; k < 10; k++)
                    if (k - L == bestinc) { dist1 = (float) sums[k - 1]; dist2 = (float) sums[k]; dist3 = (float) sums[k + 1]; }
                const float deltaR = (dist1 - dist3) / (2.0f * (dist1 + dist3 - 2.0f * dist2));
                if (!(deltaR < -1 || deltaR > 1)) {
                    float bestuR = A.scale[levelL] * ((float) scaleduR0 + (float) bestinc + deltaR);
                    float disparity = (uL - bestuR);
                    if (disparity >= 0.f && disparity < maxD) {
                        if (disparity <= 0) {
                            disparity = (float) 0.01;
                            bestuR = (float) ((double) uL - 0.01);
                        }
                        resD = A.mbf / disparity;
                        resU = bestuR;
                        resS = bestS;
                    }
                }
            }
        }
    }
    if (lane == 0) { outU[iL] = resU; outD[iL] = resD; outS[iL] = resS; }
}

// median cut: the reference sorts (SAD, index) pairs and reads element size/2; only its SAD matters
__global__ __launch_bounds__(1024) void k_stereo_cut(StereoArgs A) {
    __shared__ int s_hist[256];
    __shared__ int s_sel[4];
    const int pair = blockIdx.x, tid = threadIdx.x;
    const int nl = A.cnt[(long long) pair * A.cntStride + A.cntOffL];
    float *outU = A.uRight + (long long) pair * A.outStride, *outD = A.depth + (long long) pair * A.outStride;
    const int *sad = A.sad + (long long) pair * A.outStride;
    // pass 1: histogram of SAD >> 8 (SAD <= 61710 -> 242 bins)
    if (tid < 256) s_hist[tid] = 0;
    __syncthreads();
    for (int i = tid; i < nl; i += 1024) {
        const int s = sad[i];
        if (s >= 0) atomicAdd(&s_hist[min(s >> 8, 255)], 1);
    }
    __syncthreads();
    if (tid == 0) {
        int m = 0;
        for (int b = 0; b < 256; b++) m += s_hist[b];
        int k = m / 2, acc = 0, hb = -1;
        for (int b = 0; b < 256 && m > 0; b++) {
            if (k < acc + s_hist[b]) { hb = b; break; }
            acc += s_hist[b];
        }
        s_sel[0] = m; s_sel[1] = hb; s_sel[2] = k - acc;
    }
    __syncthreads();
    const int m = s_sel[0], hb = s_sel[1], kin = s_sel[2];
    if (m == 0) return;   // empty list: the reference reads vDistIdx[0] of an empty vector; defined: nothing to cut
    __syncthreads();
    if (tid < 256) s_hist[tid] = 0;
    __syncthreads();
    for (int i = tid; i < nl; i += 1024) {
        const int s = sad[i];
        if (s >= 0 && min(s >> 8, 255) == hb) atomicAdd(&s_hist[s & 255], 1);
    }
    __syncthreads();
    if (tid == 0) {
        int acc = 0, lb = 0;
        for (int b = 0; b < 256; b++) {
            if (kin < acc + s_hist[b]) { lb = b; break; }
            acc += s_hist[b];
        }
        s_sel[3] = (hb << 8) | lb;
    }
    __syncthreads();
    const float median = (float) s_sel[3];
    const float thDist = 1.5f * 1.4f * median;
    for (int i = tid; i < nl; i += 1024) {
        const int s = sad[i];
        if (s >= 0 && !((float) s < thDist)) { outU[i] = -1; outD[i] = -1; }
    }
}

void launch_stereo(hipStream_t st, const StereoArgs &A, int nPairs, int maxLeft, int maxRight) {
    if (nPairs <= 0 || maxLeft <= 0) return;
    hipLaunchKernelGGL(k_stereo_prep, dim3(nPairs), dim3(kStereoPrepBlock), 0, st, A);
    hipLaunchKernelGGL(k_stereo_match, dim3((maxLeft + 3) / 4, nPairs), dim3(256), 0, st, A);
    hipLaunchKernelGGL(k_stereo_cut, dim3(nPairs), dim3(1024), 0, st, A);
}

}  // namespace ygzf
