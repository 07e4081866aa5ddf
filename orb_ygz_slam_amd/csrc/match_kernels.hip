// match_kernels.hip -- ORBmatcher hot path on gfx950 (product code).
//
//   k_backproject_unit   bench/test helper: MapPoint world positions = unit-depth back-projection of the Last keypoints
//   k_match_last         ORBmatcher::SearchByProjection(Frame &Cur, const Frame &Last, th, bMono, checkLevel)
//                        reference src/ORBmatcher.cc:1218-1350, with Frame::AssignFeaturesToGrid / PosInGrid /
//                        GetFeaturesInArea (src/Frame.cc:314-330, 483-493, 424-481) and DescriptorDistance (:1507-1523)
//
// One workgroup per (Last, Cur) frame pair.  The 64x48 feature grid of Cur is rebuilt in LDS (counting sort, index order
// inside a cell as the reference's push_back gives); all waves precompute the per-query projection, search radius and
// level range; then ONE wave walks the Last keypoints in index order -- the reference's loop carries state through
// Cur.mvpMapPoints ("already owned" test, :1292-1294), so acceptance must be resolved in order -- with its 64 lanes
// spread over the grid cells of the search window, Hamming distances by __popcll on 4 x u64 from LDS-resident
// descriptors, and a (distance, candidate order) key min-reduced across the wave so that ties break exactly like the
// reference's strict `dist < bestDist` scan.  Rotation-histogram voting (:1315-1345, including the factor = 1/30 quirk)
// is applied at the end.  Float expressions are evaluated in source order (library built with -ffp-contract=off).
#include "kernels.h"

namespace ygzf {

constexpr int GRID_COLS = 64, GRID_ROWS = 48, GRID_CELLS = GRID_COLS * GRID_ROWS;
constexpr int TH_HIGH = 100;
constexpr int HISTO_LENGTH = 30;

__device__ __forceinline__ int m_lane() { return threadIdx.x & 63; }

__device__ __forceinline__ int m_wave_incl_scan(int v) {
    const int lane = m_lane();
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        int t = __shfl_up(v, d, 64);
        if (lane >= d) v += t;
    }
    return v;
}
__device__ __forceinline__ unsigned m_wave_min(unsigned v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        unsigned t = (unsigned) __shfl_xor((int) v, d, 64);
        v = t < v ? t : v;
    }
    return v;
}

__global__ void k_backproject_unit(const ygzf_kp *__restrict__ keys, const int *__restrict__ cnt, long long kpStride, float fx,
                                   float fy, float cx, float cy, float *__restrict__ world) {
    const int f = blockIdx.y;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= cnt[f]) return;
    const ygzf_kp k = keys[(long long) f * kpStride + i];
    float *w = world + ((long long) f * kpStride + i) * 3;
    w[0] = (k.x - cx) / fx;
    w[1] = (k.y - cy) / fy;
    w[2] = 1.f;
}

struct QueryParam {  // per Last keypoint, precomputed by all waves
    float u, v, radius, invzc;
    short minCx, maxCx, minCy, maxCy;
    short minLevel, maxLevel;   // GetFeaturesInArea arguments
    int valid;
};

__global__ __launch_bounds__(256) void k_match_last(MatchArgs A) {
    extern __shared__ __attribute__((aligned(16))) unsigned char dyn[];
    __shared__ int s_tmp[20];
    __shared__ int s_hist[HISTO_LENGTH];
    const int tid = threadIdx.x, lane = m_lane(), wave = tid >> 6;
    const int pair = blockIdx.x;
    const int nt = A.curCnt[(long long) pair * A.cntStrideCur + A.cntOffCur];
    const int nq = A.lastCnt[(long long) pair * A.cntStrideLast + A.cntOffLast];
    const ygzf_kp *curKeys = A.curKeys + (long long) pair * A.kpStrideCur;
    const uint8_t *curDesc = A.curDesc + (long long) pair * A.kpStrideCur * 32;
    const float *uRight = A.curURight ? A.curURight + (long long) pair * A.kpStrideCur : nullptr;
    const ygzf_kp *lastKeys = A.lastKeys + (long long) pair * A.kpStrideLast;
    const uint8_t *mpDesc = A.mpDesc + (long long) pair * A.kpStrideLast * 32;
    const float *world = A.world + (long long) pair * A.kpStrideLast * 3;
    const uint8_t *mpValid = A.mpValid ? A.mpValid + (long long) pair * A.kpStrideLast : nullptr;
    const uint8_t *outlier = A.outlier ? A.outlier + (long long) pair * A.kpStrideLast : nullptr;
    const uint8_t *hasObs = A.hasObs ? A.hasObs + (long long) pair * A.kpStrideLast : nullptr;
    uint8_t *ownerOut = A.owner + (long long) pair * A.kpStrideCur;
    int *matchOut = A.match + (long long) pair * A.kpStrideCur;
    const float *pose = A.poses + (long long) pair * 24;  // Rcw[9] tcw[3] Rlw[9] tlw[3]

    // ---- LDS carve-up ----
    unsigned char *p = dyn;
    int *cellStart = (int *) p; p += sizeof(int) * (GRID_CELLS + 1);
    int *cellFill = (int *) p; p += sizeof(int) * GRID_CELLS;
    int *list = (int *) p; p += sizeof(int) * A.capCur;
    int *events = (int *) p; p += sizeof(int) * A.capLast;
    QueryParam *qp;
    if (A.qpInLds) { qp = (QueryParam *) p; p += sizeof(QueryParam) * A.capLast; }
    else qp = (QueryParam *) A.qpScratch + (long long) pair * A.capLast;
    unsigned long long *ldsDesc = (unsigned long long *) p; p += A.descInLds ? (size_t) 32 * A.capCur : 0;
    unsigned char *owner = p; p += A.capCur;
    unsigned char *octave = p; p += A.capCur;

    // ---- Frame::AssignFeaturesToGrid ----
    for (int i = tid; i < GRID_CELLS; i += 256) cellFill[i] = 0;
    if (tid < HISTO_LENGTH) s_hist[tid] = 0;
    __syncthreads();
    for (int i = tid; i < nt; i += 256) {
        const ygzf_kp k = curKeys[i];
        const int px = (int) roundf((k.x - A.minX) * A.gridInvW);
        const int py = (int) roundf((k.y - A.minY) * A.gridInvH);
        if (!(px < 0 || px >= GRID_COLS || py < 0 || py >= GRID_ROWS)) atomicAdd(&cellFill[px * GRID_ROWS + py], 1);
        owner[i] = A.ownerIn ? A.ownerIn[(long long) pair * A.kpStrideCur + i] : 0;
        octave[i] = (unsigned char) k.octave;
        matchOut[i] = -1;
        if (A.descInLds) {
            const unsigned long long *d = (const unsigned long long *) (curDesc + (size_t) i * 32);
            ldsDesc[4 * i] = d[0]; ldsDesc[4 * i + 1] = d[1]; ldsDesc[4 * i + 2] = d[2]; ldsDesc[4 * i + 3] = d[3];
        }
    }
    __syncthreads();
    {   // exclusive scan of 3072 counts: 12 per thread
        const int per = GRID_CELLS / 256;
        int s = 0;
        for (int k = 0; k < per; k++) s += cellFill[tid * per + k];
        int incl = m_wave_incl_scan(s);
        if (lane == 63) s_tmp[wave] = incl;
        __syncthreads();
        int woff = 0;
        for (int w2 = 0; w2 < wave; w2++) woff += s_tmp[w2];
        int off = woff + incl - s;
        for (int k = 0; k < per; k++) {
            const int c = cellFill[tid * per + k];
            cellStart[tid * per + k] = off;
            off += c;
        }
        if (tid == 255) cellStart[GRID_CELLS] = off;
    }
    __syncthreads();
    for (int i = tid; i < GRID_CELLS; i += 256) cellFill[i] = cellStart[i];
    __syncthreads();
    for (int i = tid; i < nt; i += 256) {
        const ygzf_kp k = curKeys[i];
        const int px = (int) roundf((k.x - A.minX) * A.gridInvW);
        const int py = (int) roundf((k.y - A.minY) * A.gridInvH);
        if (!(px < 0 || px >= GRID_COLS || py < 0 || py >= GRID_ROWS)) list[atomicAdd(&cellFill[px * GRID_ROWS + py], 1)] = i;
    }
    __syncthreads();
    for (int c = tid; c < GRID_CELLS; c += 256) {  // cells keep ascending keypoint index (push_back order)
        const int s = cellStart[c], e = cellStart[c + 1];
        for (int a = s + 1; a < e; a++) {
            const int v = list[a];
            int b = a - 1;
            while (b >= s && list[b] > v) { list[b + 1] = list[b]; b--; }
            list[b + 1] = v;
        }
    }
    // ---- per-query projection (:1243-1272) ----
    const float *Rcw = pose, *tcw = pose + 9, *Rlw = pose + 12, *tlw = pose + 21;
    float twc[3], tlc2;
    for (int i = 0; i < 3; i++) twc[i] = -1 * (Rcw[i] * tcw[0] + Rcw[3 + i] * tcw[1] + Rcw[6 + i] * tcw[2]);
    tlc2 = (Rlw[6] * twc[0] + Rlw[7] * twc[1] + Rlw[8] * twc[2]) + tlw[2];
    const bool bForward = tlc2 > A.mb && !A.bMono;
    const bool bBackward = -tlc2 > A.mb && !A.bMono;
    for (int i = tid; i < nq; i += 256) {
        QueryParam q;
        q.valid = 0;
        q.u = q.v = q.radius = q.invzc = 0;
        q.minCx = q.maxCx = q.minCy = q.maxCy = 0;
        q.minLevel = q.maxLevel = -1;
        const bool has = (mpValid ? mpValid[i] != 0 : true) && !(outlier ? outlier[i] != 0 : false);
        if (has) {
            const float *X = world + 3 * (size_t) i;
            const float xc = (Rcw[0] * X[0] + Rcw[1] * X[1] + Rcw[2] * X[2]) + tcw[0];
            const float yc = (Rcw[3] * X[0] + Rcw[4] * X[1] + Rcw[5] * X[2]) + tcw[1];
            const float zc = (Rcw[6] * X[0] + Rcw[7] * X[1] + Rcw[8] * X[2]) + tcw[2];
            const float invzc = (float) (1.0 / (double) zc);
            if (!(invzc < 0)) {
                const float u = A.fx * xc * invzc + A.cx;
                const float v = A.fy * yc * invzc + A.cy;
                if (!(u < A.minX || u > A.maxX) && !(v < A.minY || v > A.maxY)) {
                    const int oct = lastKeys[i].octave;
                    const float r = A.th * A.scaleFactors[oct];
                    int minL, maxL;
                    if (!A.checkLevel) { minL = -1; maxL = -1; }
                    else if (bForward) { minL = oct; maxL = -1; }
                    else if (bBackward) { minL = 0; maxL = oct; }
                    else { minL = oct - 1; maxL = oct + 1; }
                    // GetFeaturesInArea cell window (src/Frame.cc:429-447)
                    const int nMinCellX = max(0, (int) floorf((u - A.minX - r) * A.gridInvW));
                    const int nMaxCellX = min(GRID_COLS - 1, (int) ceilf((u - A.minX + r) * A.gridInvW));
                    const int nMinCellY = max(0, (int) floorf((v - A.minY - r) * A.gridInvH));
                    const int nMaxCellY = min(GRID_ROWS - 1, (int) ceilf((v - A.minY + r) * A.gridInvH));
                    if (!(nMinCellX >= GRID_COLS || nMaxCellX < 0 || nMinCellY >= GRID_ROWS || nMaxCellY < 0) &&
                        nMaxCellX >= nMinCellX && nMaxCellY >= nMinCellY) {
                        q.valid = 1;
                        q.u = u; q.v = v; q.radius = r; q.invzc = invzc;
                        q.minCx = (short) nMinCellX; q.maxCx = (short) nMaxCellX;
                        q.minCy = (short) nMinCellY; q.maxCy = (short) nMaxCellY;
                        q.minLevel = (short) minL; q.maxLevel = (short) maxL;
                    }
                }
            }
        }
        qp[i] = q;
    }
    __syncthreads();
    if (wave != 0) return;

    // ---- in-order resolution by one wave ----
    int nmatches = 0, nEvents = 0;
    const float factor = 1.0f / HISTO_LENGTH;
    QueryParam qn = nq > 0 ? qp[0] : QueryParam();
    unsigned long long n0 = 0, n1 = 0, n2 = 0, n3 = 0;
    if (nq > 0) { const unsigned long long *qd = (const unsigned long long *) mpDesc; n0 = qd[0]; n1 = qd[1]; n2 = qd[2]; n3 = qd[3]; }
    for (int i = 0; i < nq; i++) {
        const QueryParam q = qn;
        const unsigned long long q0 = n0, q1 = n1, q2 = n2, q3 = n3;
        if (i + 1 < nq) {  // prefetch the next query while this one is resolved
            qn = qp[i + 1];
            const unsigned long long *qd = (const unsigned long long *) (mpDesc + (size_t) (i + 1) * 32);
            n0 = qd[0]; n1 = qd[1]; n2 = qd[2]; n3 = qd[3];
        }
        if (!q.valid) continue;
        const int nCy = q.maxCy - q.minCy + 1;
        const int nc = (q.maxCx - q.minCx + 1) * nCy;
        const bool bCheckLevels = (q.minLevel > 0) || (q.maxLevel >= 0);
        unsigned best = (256u << 16) | 0xFFFFu;
        int bestI2 = -1;
        int orderBase = 0;
        for (int cbase = 0; cbase < nc; cbase += 64) {
            const int ci = cbase + lane;
            int s = 0, cnt = 0;
            if (ci < nc) {
                const int ix = q.minCx + ci / nCy, iy = q.minCy + ci % nCy;  // for ix: for iy: (src/Frame.cc:451-452)
                s = cellStart[ix * GRID_ROWS + iy];
                cnt = cellStart[ix * GRID_ROWS + iy + 1] - s;
            }
            const int incl = m_wave_incl_scan(cnt);
            const int ord0 = orderBase + incl - cnt;
            for (int k = 0; k < cnt; k++) {
                const int i2 = list[s + k];
                if (bCheckLevels) {
                    const int o = octave[i2];
                    if (o < q.minLevel) continue;
                    if (q.maxLevel >= 0 && o > q.maxLevel) continue;
                }
                const ygzf_kp kc = curKeys[i2];
                const float distx = kc.x - q.u, disty = kc.y - q.v;
                if (!(fabsf(distx) < q.radius && fabsf(disty) < q.radius)) continue;
                if (owner[i2] == 2) continue;  // mvpMapPoints[i2] && Observations() > 0
                if (uRight && uRight[i2] > 0) {
                    const float ur = q.u - A.mbf * q.invzc;
                    const float er = fabsf(ur - uRight[i2]);
                    if (er > q.radius) continue;
                }
                unsigned long long d0, d1, d2, d3;
                if (A.descInLds) {
                    d0 = ldsDesc[4 * i2]; d1 = ldsDesc[4 * i2 + 1]; d2 = ldsDesc[4 * i2 + 2]; d3 = ldsDesc[4 * i2 + 3];
                } else {
                    const unsigned long long *d = (const unsigned long long *) (curDesc + (size_t) i2 * 32);
                    d0 = d[0]; d1 = d[1]; d2 = d[2]; d3 = d[3];
                }
                const unsigned dist = __popcll(q0 ^ d0) + __popcll(q1 ^ d1) + __popcll(q2 ^ d2) + __popcll(q3 ^ d3);
                const unsigned key = (dist << 16) | (unsigned) (ord0 + k);
                if (key < best) { best = key; bestI2 = i2; }
            }
            orderBase += __shfl(incl, 63, 64);
        }
        const unsigned wbest = m_wave_min(best);
        const int bestDist = (int) (wbest >> 16);
        if (bestDist <= TH_HIGH) {
            const unsigned long long who = __ballot(best == wbest);
            const int src = __ffsll((long long) who) - 1;
            const int bestIdx2 = __shfl(bestI2, src, 64);
            if (lane == 0) {
                owner[bestIdx2] = (hasObs ? hasObs[i] != 0 : true) ? 2 : 1;
                matchOut[bestIdx2] = i;
            }
            nmatches++;
            if (A.checkOri) {
                float rot = lastKeys[i].angle - curKeys[bestIdx2].angle;
                if (rot < 0.0) rot += 360.0f;
                int bin = (int) roundf(rot * factor);
                if (bin == HISTO_LENGTH) bin = 0;
                if (lane == 0) {
                    events[nEvents] = (bin << 24) | bestIdx2;
                    s_hist[bin]++;
                }
                nEvents++;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
    }
    // ---- rotation consistency (:1327-1345) + ComputeThreeMaxima (:1471-1502) ----
    if (A.checkOri) {
        int ind1 = -1, ind2 = -1, ind3 = -1;
        int max1 = 0, max2 = 0, max3 = 0;
        for (int b = 0; b < HISTO_LENGTH; b++) {
            const int s = s_hist[b];
            if (s > max1) { max3 = max2; max2 = max1; max1 = s; ind3 = ind2; ind2 = ind1; ind1 = b; }
            else if (s > max2) { max3 = max2; max2 = s; ind3 = ind2; ind2 = b; }
            else if (s > max3) { max3 = s; ind3 = b; }
        }
        if (max2 < 0.1f * (float) max1) { ind2 = -1; ind3 = -1; }
        else if (max3 < 0.1f * (float) max1) { ind3 = -1; }
        int removed = 0;
        for (int e = lane; e < nEvents; e += 64) {
            const int ev = events[e];
            const int bin = ev >> 24, idx = ev & 0xFFFFFF;
            if (bin != ind1 && bin != ind2 && bin != ind3) {
                owner[idx] = 0;
                matchOut[idx] = -2;
                removed++;
            }
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) removed += __shfl_xor(removed, d, 64);
        nmatches -= removed;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    for (int i = lane; i < nt; i += 64) ownerOut[i] = owner[i];
    if (lane == 0) A.nmatches[pair] = nmatches;
}

static_assert(sizeof(QueryParam) == 32, "QueryParam is 32 bytes (qpScratch sizing)");

size_t match_lds_bytes(int capCur, int capLast, bool descInLds, bool qpInLds) {
    size_t b = sizeof(int) * (GRID_CELLS + 1) + sizeof(int) * GRID_CELLS + sizeof(int) * (size_t) capCur + sizeof(int) * (size_t) capLast +
               (qpInLds ? sizeof(QueryParam) * (size_t) capLast : 0) + (descInLds ? (size_t) 32 * capCur : 0) + 2 * (size_t) capCur + 64;
    return (b + 15) & ~(size_t) 15;
}

hipError_t match_prepare(size_t ldsBytes) {
    return hipFuncSetAttribute((const void *) k_match_last, hipFuncAttributeMaxDynamicSharedMemorySize, (int) ldsBytes);
}

void launch_backproject_unit(hipStream_t st, const ygzf_kp *keys, const int *cnt, long long kpStride, int maxKp, int nFrames, float fx,
                             float fy, float cx, float cy, float *world) {
    if (maxKp <= 0) return;
    hipLaunchKernelGGL(k_backproject_unit, dim3((maxKp + 255) / 256, nFrames), dim3(256), 0, st, keys, cnt, kpStride, fx, fy, cx, cy, world);
}

void launch_match_last(hipStream_t st, const MatchArgs &A, int nPairs, size_t ldsBytes) {
    hipLaunchKernelGGL(k_match_last, dim3(nPairs), dim3(256), ldsBytes, st, A);
}

}  // namespace ygzf
