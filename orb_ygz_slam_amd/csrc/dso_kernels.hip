// dso_kernels.hip -- the FAST-10 grid detector of the DSO_KEYPOINT path on gfx950 (product code).
//
//   ORBextractor::ComputeKeyPointsDSOSingleLevel   src/ORBextractor.cc:1275-1386
//   ORBextractor::ShiTomasiScore                   src/ORBextractor.cc:1152-1187
//
// One wave per inner grid cell: the cell (+5 px apron: 3 for the FAST ring, 5 for the Shi-Tomasi box) is staged in LDS, FAST-10
// runs at barrier 20 and -- only if that finds nothing -- at the hard-coded barrier 5 (:1337); corners closer than 20 px to the
// image edge or on an occupied pixel are dropped, the rest get their Shi-Tomasi score and the 3 best per cell survive.
// Definitions where the reference is undefined (see oracle/ygz_oracle.h): ties of the score sort keep raster order; a NaN score
// (float round-off can make the discriminant negative) ranks lowest.
#include "fast10_device.h"
#include "kernels.h"
#include "wave_ops.h"

namespace ygzf {

constexpr int kDsoWaves = 4;
constexpr int kDsoTileP = kDsoMaxGrid + 12;   // row pitch of the staged cell (grid + 10 columns used)

// occupancy bitmap over level 0 (bit y*w + x): pixels that already carry a keypoint (:1286-1291)
__global__ void k_dso_occ(const unsigned *__restrict__ xy, int n, int w, int h, unsigned *__restrict__ occ) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int x = xy[i] & 0xFFFFu, y = xy[i] >> 16;
    if (x >= w || y >= h) return;   // the reference would write out of bounds
    const long long b = (long long) y * w + x;
    atomicOr(&occ[b >> 5], 1u << (b & 31));
}

__device__ __forceinline__ unsigned score_rank(float s) {   // monotone u32 image of the float order; NaN lowest, -0 == +0
    if (s != s) return 0u;
    if (s == 0.f) s = 0.f;
    const unsigned b = __float_as_uint(s);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}

// th0 / th1: the two libfast barriers (20 / 5 hard-coded in ComputeKeyPointsDSOSingleLevel, iniThFAST / minThFAST in the multi-level
// ComputeKeyPointsDSO); take: winners per cell (3 / 2); occW: row length of the occupancy bitmap (the level-0 width: the multi-level detector
// indexes its level-0-sized map with LEVEL coordinates, :1466); mark: every winner sets its bit at once (:1482) -- cells own disjoint pixels,
// so no other cell of the pass reads a bit written here
__global__ __launch_bounds__(64 * kDsoWaves) void k_dso_cells(const uint8_t *__restrict__ img, int pitch, int w, int h, int grid, int nCols, int nRows,
                                                            unsigned *__restrict__ occ, int *__restrict__ cellCnt,
                                                            unsigned *__restrict__ cellXY, int *__restrict__ total, int th0, int th1, int take,
                                                            int occW, int mark) {
    __shared__ uint8_t tiles[kDsoWaves][(kDsoMaxGrid + 10) * kDsoTileP];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int innerCols = nCols - 2, nInner = innerCols * (nRows - 2);
    const int cell = blockIdx.x * kDsoWaves + wave;
    if (cell >= nInner) return;   // waves are independent: no block barrier below
    const int cyi = cell / innerCols + 1, cxi = cell - (cyi - 1) * innerCols + 1;   // border cells are skipped (:1316-1319)
    const int x_start = cxi * grid, y_start = cyi * grid;
    uint8_t *T = tiles[wave];
    const int TS = grid + 10;
    for (int idx = lane; idx < TS * TS; idx += 64) {
        const int r = idx / TS, c = idx - r * TS;
        T[r * kDsoTileP + c] = img[(long long) (y_start - 5 + r) * pitch + x_start - 5 + c];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // fast_corner_detect_10_sse2 domain: cells narrower than 22 px fall to the plain detector, which scans the whole cell
    const int dom0 = grid < 22 ? 0 : 3, D = grid < 22 ? grid : grid - 6, nPix = D * D;
    unsigned long long k0 = 0, k1 = 0, k2 = 0;   // this lane's three best (score rank << 32 | ~raster index); 0 = none
    int nc = 0;
    for (int pass = 0; pass < 2 && nc == 0; pass++) {
        const int barrier = pass == 0 ? th0 : th1;
        for (int base = 0; base < nPix; base += 64) {
            const int i = base + lane;
            const int yy = dom0 + i / D, xx = dom0 + i % D;
            bool corner = false;
            if (i < nPix) corner = arc10_margin(&T[(yy + 5) * kDsoTileP + xx + 5], kDsoTileP) > barrier;
            nc += __popcll(__ballot(corner));
            if (!corner) continue;
            const int x = x_start + xx, y = y_start + yy;
            if (x < 20 || y < 20 || x >= w - 20 || y >= h - 20) continue;   // :1343-1345
            const long long ob = (long long) y * occW + x;
            if ((occ[ob >> 5] >> (ob & 31)) & 1u) continue;                // :1347-1348
            // ShiTomasiScore: 8x8 box rows y-4..y+3, cols x-4..x+3; the sums are exact integers (< 2^24) in the reference's floats.
            // Its own border guard (:1162-1163) can never fire 20 px inside the image.
            int sxx = 0, syy = 0, sxy = 0;
            const uint8_t *c0 = &T[(yy + 5 - 4) * kDsoTileP + xx + 5 - 4];
#pragma unroll
            for (int r = 0; r < 8; r++) {
#pragma unroll
                for (int q = 0; q < 8; q++) {
                    const int dx = (int) c0[r * kDsoTileP + q + 1] - (int) c0[r * kDsoTileP + q - 1];
                    const int dy = (int) c0[(r + 1) * kDsoTileP + q] - (int) c0[(r - 1) * kDsoTileP + q];
                    sxx += dx * dx;
                    syy += dy * dy;
                    sxy += dx * dy;
                }
            }
            const float a = (float) sxx * (1.f / 128.f), b = (float) syy * (1.f / 128.f), c = (float) sxy * (1.f / 128.f);
            const float score = 0.5f * (a + b - sqrtf((a + b) * (a + b) - 4.f * (a * b - c * c)));
            const unsigned long long key = ((unsigned long long) score_rank(score) << 32) | (0xFFFFFFFFu - (unsigned) i);
            if (key > k0) { k2 = k1; k1 = k0; k0 = key; }
            else if (key > k1) { k2 = k1; k1 = key; }
            else if (key > k2) k2 = key;
        }
    }
    int taken = 0;
    for (int r = 0; r < take; r++) {
        const unsigned long long best = wave_max_u64(k0);
        if (best == 0) break;
        if (k0 == best) {   // keys are unique
            const int i = (int) (0xFFFFFFFFu - (unsigned) best);
            const int yy = dom0 + i / D, xx = dom0 + i % D;
            cellXY[cell * 3 + r] = (unsigned) (x_start + xx) | ((unsigned) (y_start + yy) << 16);
            if (mark) {
                const long long ob = (long long) (y_start + yy) * occW + x_start + xx;
                atomicOr(&occ[ob >> 5], 1u << (ob & 31));
            }
            k0 = k1; k1 = k2; k2 = 0;
        }
        taken++;
    }
    if (lane == 0) {
        cellCnt[cell] = taken;
        if (taken) atomicAdd(total, taken);
    }
}

// cell-order compaction of the per-cell winners into the describe list (after the nExisting entries of the frame's own keys)
__global__ __launch_bounds__(1024) void k_dso_compact(const int *__restrict__ cellCnt, const unsigned *__restrict__ cellXY, int nInner, int nExisting,
                                                     int4 *__restrict__ list, unsigned *__restrict__ newXY, int level) {
    __shared__ int s[1024];
    const int tid = threadIdx.x;
    const int per = (nInner + 1023) / 1024, b = tid * per, e = min(nInner, b + per);
    int sum = 0;
    for (int i = b; i < e; i++) sum += cellCnt[i];
    s[tid] = sum;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
        const int t = tid >= d ? s[tid - d] : 0;
        __syncthreads();
        s[tid] += t;
        __syncthreads();
    }
    int off = s[tid] - sum;
    for (int i = b; i < e; i++) {
        const int n = cellCnt[i];
        for (int k = 0; k < n; k++) {
            const unsigned xy = cellXY[i * 3 + k];
            newXY[off] = xy;
            list[nExisting + off] = make_int4((int) (xy & 0xFFFFu), (int) (xy >> 16), level, 0);
            off++;
        }
    }
}

void launch_dso_occ(hipStream_t st, const unsigned *xy, int n, int w, int h, unsigned *occ) {
    if (n > 0) hipLaunchKernelGGL(k_dso_occ, dim3((n + 255) / 256), dim3(256), 0, st, xy, n, w, h, occ);
}

void launch_dso_cells(hipStream_t st, const uint8_t *img, int pitch, int w, int h, int grid, int nCols, int nRows, unsigned *occ, int *cellCnt,
                      unsigned *cellXY, int *total, int th0, int th1, int take, int occW, bool mark) {
    const int nInner = (nCols - 2) * (nRows - 2);
    if (nInner <= 0) return;
    hipLaunchKernelGGL(k_dso_cells, dim3((nInner + kDsoWaves - 1) / kDsoWaves), dim3(64 * kDsoWaves), 0, st, img, pitch, w, h, grid, nCols, nRows, occ,
                       cellCnt, cellXY, total, th0, th1, take, occW, mark ? 1 : 0);
}

void launch_dso_compact(hipStream_t st, const int *cellCnt, const unsigned *cellXY, int nInner, int nExisting, void *list, unsigned *newXY, int level) {
    if (nInner <= 0) return;
    hipLaunchKernelGGL(k_dso_compact, dim3(1), dim3(1024), 0, st, cellCnt, cellXY, nInner, nExisting, (int4 *) list, newXY, level);
}

// ---- ComputeKeyPointsFast (FAST_KEYPOINT, src/ORBextractor.cc:1189-1273): per 5x5-px cell of the level-0 grid the corner with the largest
// Shi-Tomasi score among the non-max-suppressed libfast corners of ALL levels; `s > response` keeps the first of equal scores in (level,
// libfast order), which an atomicMax over (score bits << 32 | ~order) reproduces: positive floats order like their bit patterns.
__device__ __forceinline__ float shi_tomasi_global(const uint8_t *__restrict__ img, int pitch, int w, int h, int u, int v) {   // :1152-1187
    if (u - 4 < 1 || u + 4 >= w - 1 || v - 4 < 1 || v + 4 >= h - 1) return 0.f;
    int sxx = 0, syy = 0, sxy = 0;
    for (int r = 0; r < 8; r++) {
        const uint8_t *row = img + (long long) (v - 4 + r) * pitch + (u - 4);
#pragma unroll
        for (int q = 0; q < 8; q++) {
            const int dx = (int) row[q + 1] - (int) row[q - 1];
            const int dy = (int) row[q + pitch] - (int) row[q - pitch];
            sxx += dx * dx; syy += dy * dy; sxy += dx * dy;
        }
    }
    const float a = (float) sxx * (1.f / 128.f), b = (float) syy * (1.f / 128.f), c = (float) sxy * (1.f / 128.f);
    return 0.5f * (a + b - sqrtf((a + b) * (a + b) - 4.f * (a * b - c * c)));
}

__global__ void k_fgrid_vote(const uint8_t *__restrict__ img, int pitch, int w, int h, int level, float scale, const short *__restrict__ xy,
                             const int *__restrict__ nonmax, const int *__restrict__ totals, int gridCols, long long nCells,
                             const uint8_t *__restrict__ occ, unsigned long long *__restrict__ cellKey) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= totals[1]) return;
    const int c = nonmax[i];
    const int x = (int) xy[2 * c] + 20, y = (int) xy[2 * c + 1] + 20;                  // xy.x += boarder (:1238-1239)
    const int gy = (int) (((float) y * scale) / 5.0f), gx = (int) (((float) x * scale) / 5.0f);   // :1242-1243
    const long long k = (long long) gy * gridCols + gx;
    if (k < 0 || k >= nCells) return;                                                   // defined: outside the grid -> ignored (the reference indexes out of bounds)
    if (occ[k]) return;
    const float s = shi_tomasi_global(img, pitch, w, h, x, y);
    if (!(s > 0.f)) return;                                                             // `s > gridFeatures[k].response` with response 0 at first
    const unsigned order = ((unsigned) level << 24) | (unsigned) i;
    atomicMax(&cellKey[k], ((unsigned long long) __float_as_uint(s) << 32) | (0xFFFFFFFFu - order));
}

// winners -> their level coordinates: cellXY[k] = x | y << 16 of the corner whose key stands in cell k (0xFFFFFFFF where the cell is empty)
__global__ void k_fgrid_gather(const unsigned long long *__restrict__ cellKey, long long nCells, const short *__restrict__ xyAll,
                               const int *__restrict__ nmAll, const long long *__restrict__ lvlOff, unsigned *__restrict__ cellXY) {
    const long long k = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= nCells) return;
    const unsigned long long key = cellKey[k];
    if (key == 0) { cellXY[k] = 0xFFFFFFFFu; return; }
    const unsigned order = 0xFFFFFFFFu - (unsigned) key;
    const int level = (int) (order >> 24), i = (int) (order & 0xFFFFFFu);
    const long long off = lvlOff[level];
    const int c = nmAll[off + i];
    cellXY[k] = (unsigned) ((int) xyAll[2 * (off + c)] + 20) | ((unsigned) ((int) xyAll[2 * (off + c) + 1] + 20) << 16);
}

void launch_fgrid_vote(hipStream_t st, const uint8_t *img, int pitch, int w, int h, int level, float scale, const short *xy, const int *nonmax,
                       const int *totals, int maxNonmax, int gridCols, long long nCells, const uint8_t *occ, unsigned long long *cellKey) {
    if (maxNonmax <= 0) return;
    hipLaunchKernelGGL(k_fgrid_vote, dim3((maxNonmax + 255) / 256), dim3(256), 0, st, img, pitch, w, h, level, scale, xy, nonmax, totals, gridCols, nCells, occ,
                       cellKey);
}
void launch_fgrid_gather(hipStream_t st, const unsigned long long *cellKey, long long nCells, const short *xyAll, const int *nmAll, const long long *lvlOff,
                         unsigned *cellXY) {
    hipLaunchKernelGGL(k_fgrid_gather, dim3((unsigned) ((nCells + 255) / 256)), dim3(256), 0, st, cellKey, nCells, xyAll, nmAll, lvlOff, cellXY);
}

}  // namespace ygzf
