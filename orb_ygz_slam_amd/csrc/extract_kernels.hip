// extract_kernels.hip -- hand-written gfx950 kernels of the ORB extractor hot path (product code).
//
//   k_pyr_resize   K1  ORBextractor::ComputePyramid          (reference src/ORBextractor.cc:1129-1150, cv::resize)
//   k_fast_quads   K2  ComputeKeyPointsOctTree cell loop     (:747-781, cv::FAST 9/16 + per-cell NMS + threshold fallback)
//   k_octree       K4  DistributeOctTree                      (:533-723) as sort-by-path-key + breadth-first on ranges
//   k_describe     K5+K6+K7  IC_Angle (:77-101), GaussianBlur 7x7 s=2 (:1010) on the 37x37 patch only,
//                            computeOrbDescriptor (:105-149)
//
// Everything here is integer / compare-select work bounded by HBM traffic and launch count, not by MFMA.
// Built with -ffp-contract=off: the few float expressions (fastAtan2, the rBRIEF rotation) must round exactly like
// the reference's scalar C++ (no FMA), see DESIGN.md "float details inside bit-exact descriptors".
#include <cstdlib>

#include <cstring>
#include <mutex>
#include "kernels.h"
#include "wave_ops.h"
#include "orb_pattern_table.h"

namespace ygzf {

// ------------------------------------------------------------------------------------------------------------------
// small wave / block primitives (wave = 64 lanes)
// ------------------------------------------------------------------------------------------------------------------
typedef short v2s __attribute__((ext_vector_type(2)));
typedef unsigned short v2u __attribute__((ext_vector_type(2)));

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }
__device__ __forceinline__ int wave_id() { return __builtin_amdgcn_readfirstlane(threadIdx.x >> 6); }   // wave-uniform by construction

// Phase clock of the two kernels that carry half of the chain (k_fast_tab, k_describe): a wave stamps s_memtime (shader clock) at the phase
// borders of its one cell / keypoint and one wave in 64 leaves its per-phase cycles in a record of g_phase_rec[kernel].  Compiled in only with -DYGZF_PHASE_CLOCK (python -m orb_ygz_slam_amd.build --phase-clock builds
// lib_ab/libygzf_clk.so; tools/fast_phases.py reads it through ygzf_phase_clocks): the product library carries none of it -- the struct is empty
// and every call folds away.  A stamp first drains the wave's outstanding memory / LDS operations, so that a phase is charged with its own waits.
#ifdef YGZF_PHASE_CLOCK
// No atomics: a sampled wave (one in 64) stores its record into slot (sample index % kPhaseSlots) of its kernel's table -- later samples overwrite
// earlier ones, the host averages the slots that were written.  (A first version added every 16th wave's cycles to ten global counters: the
// contended atomics backed up the memory pipeline and stretched k_fast_tab 5.5x -- the clock measured itself.)
constexpr int kPhaseSlots = 8192;
__device__ unsigned g_phase_rec[2][kPhaseSlots][10];   // 8 phases, [8] pass-1 runs, [9] written flag
struct PhaseClk {
    unsigned long long t;
    unsigned a[8], extra;
    __device__ __forceinline__ void bump() { extra++; }
    __device__ __forceinline__ void start() {
        extra = 0;
        __builtin_amdgcn_sched_barrier(0);
        t = __builtin_amdgcn_s_memtime();
#pragma unroll
        for (int k = 0; k < 8; k++) a[k] = 0;
        __builtin_amdgcn_sched_barrier(0);
    }
    __device__ __forceinline__ void mark(int k) {
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        const unsigned long long n = __builtin_amdgcn_s_memtime();
        a[k] += (unsigned) (n - t);
        t = n;
        __builtin_amdgcn_sched_barrier(0);
    }
    // `index`: the wave's running number inside the launch (any numbering that spreads over the grid)
    __device__ __forceinline__ void flush(int kernel, int lane, unsigned index) {
        if ((index & 63u) != 0 || lane >= 10) return;
        unsigned v = lane == 9 ? 1u : lane == 8 ? extra : 0u;
#pragma unroll
        for (int k = 0; k < 8; k++) v = lane == k ? a[k] : v;
        g_phase_rec[kernel][(index >> 6) % kPhaseSlots][lane] = v;
    }
};
hipError_t phase_clocks_read(int kernel, unsigned long long *out16, bool reset) {
    static unsigned host[kPhaseSlots][10];
    const size_t bytes = sizeof host, off = (size_t) kernel * bytes;
    hipError_t e = hipMemcpyFromSymbol(host, HIP_SYMBOL(g_phase_rec), bytes, off);
    if (e != hipSuccess) return e;
    for (int k = 0; k < 16; k++) out16[k] = 0;
    for (int s = 0; s < kPhaseSlots; s++) {
        if (!host[s][9]) continue;
        for (int k = 0; k < 8; k++) out16[k] += host[s][k];
        out16[14] += host[s][8];
        out16[15] += 1;
    }
    if (!reset) return e;
    memset(host, 0, bytes);
    return hipMemcpyToSymbol(HIP_SYMBOL(g_phase_rec), host, bytes, off);
}
#else
struct PhaseClk {
    __device__ __forceinline__ void start() {}
    __device__ __forceinline__ void mark(int) {}
    __device__ __forceinline__ void bump() {}
    __device__ __forceinline__ void flush(int, int, unsigned) {}
};
hipError_t phase_clocks_read(int, unsigned long long *, bool) { return hipErrorNotSupported; }
#endif

// Exclusive prefix of v over the threads of the block (thread order); *total = block sum.  tmp: >= 17 ints of LDS.
// Contains block barriers: must be called by all threads.  The caller orders its reads of tmp's results before the next call (a barrier).
__device__ __forceinline__ int block_excl_scan(int v, int *tmp, int *total) {
    const int incl = wave_incl_scan(v);
    const int nw = (blockDim.x + 63) >> 6;
    // (no barrier on entry: the only caller, block_scan_array, ends with one after the last read of tmp)
    if (lane_id() == 63) tmp[wave_id()] = incl;
    __syncthreads();
    if (threadIdx.x < 64) {
        int w = threadIdx.x < nw ? tmp[threadIdx.x] : 0;
        int wi = wave_incl_scan(w);
        if (threadIdx.x < nw) tmp[threadIdx.x] = wi - w;
        if (threadIdx.x == nw - 1) tmp[16] = wi;
    }
    __syncthreads();
    *total = tmp[16];
    return tmp[wave_id()] + incl - v;
}

// In-place exclusive scan of an LDS array a[0..n); returns the total.  Each thread owns a contiguous chunk.
__device__ __forceinline__ int block_scan_array(int *a, int n, int *tmp) {
    const int per = (n + blockDim.x - 1) / blockDim.x;
    const int b = threadIdx.x * per, e = min(n, b + per);
    int s = 0;
    for (int i = b; i < e; i++) s += a[i];
    int total;
    int off = block_excl_scan(s, tmp, &total);
    for (int i = b; i < e; i++) {
        int v = a[i];
        a[i] = off;
        off += v;
    }
    __syncthreads();
    return total;
}

// In-place exclusive scans of three LDS arrays a, b, c [0..n) at once: the three running sums travel as 21-bit fields of one 64-bit word
// (every sum < 2^21: callers scan child / node counts bounded by the candidate count), so the block pays one set of barriers instead of
// three.  tmp64: >= 17 u64 of LDS.  Returns the totals.
__device__ __forceinline__ void block_scan_array3(int *a, int *b, int *c, int n, unsigned long long *tmp64, int *totA, int *totB, int *totC) {
    const int per = (n + blockDim.x - 1) / blockDim.x;
    const int lo = threadIdx.x * per, hi = min(n, lo + per);
    unsigned long long s = 0;
    for (int i = lo; i < hi; i++) s += (unsigned long long) (unsigned) a[i] | ((unsigned long long) (unsigned) b[i] << 21) | ((unsigned long long) (unsigned) c[i] << 42);
    const unsigned long long incl = wave_incl_scan_u64(s);
    const int nw = (blockDim.x + 63) >> 6;
    // (no barrier on entry: the previous call's reads of tmp64 lie before its closing barrier)
    if (lane_id() == 63) tmp64[wave_id()] = incl;
    __syncthreads();
    if (threadIdx.x < 64) {
        const unsigned long long w = threadIdx.x < nw ? tmp64[threadIdx.x] : 0ull;
        const unsigned long long wi = wave_incl_scan_u64(w);
        if (threadIdx.x < nw) tmp64[threadIdx.x] = wi - w;
        if (threadIdx.x == nw - 1) tmp64[16] = wi;
    }
    __syncthreads();
    const unsigned long long total = tmp64[16];
    unsigned long long off = tmp64[wave_id()] + incl - s;
    for (int i = lo; i < hi; i++) {
        const unsigned long long v = (unsigned long long) (unsigned) a[i] | ((unsigned long long) (unsigned) b[i] << 21) | ((unsigned long long) (unsigned) c[i] << 42);
        a[i] = (int) (off & 0x1FFFFFu); b[i] = (int) ((off >> 21) & 0x1FFFFFu); c[i] = (int) (off >> 42);
        off += v;
    }
    __syncthreads();
    *totA = (int) (total & 0x1FFFFFu); *totB = (int) ((total >> 21) & 0x1FFFFFu); *totC = (int) (total >> 42);
}

// Stable LSD radix sort (kRadixBits-bit digits) of n (key,val) pairs on `bits` key bits by the whole block.
// k0/v0 hold the input; result is left in *rk/*rv (one of the two buffers).  Buffers may be LDS or global.
// histT: kRadixHist ints of LDS (digit-major, one counter per wave), tmp: 17 ints of LDS.
// Seven-bit digits: the octree's 21-bit path keys take three passes (4-bit digits took six, at ~3.4 us of mostly barrier time each).
constexpr int kRadixBits = 7;
constexpr int kRadixHist = (1 << kRadixBits) * 16;   // up to 16 waves
__device__ void block_radix_sort(unsigned *k0, unsigned *v0, unsigned *k1, unsigned *v1, int n, int bits,
                                 volatile int *histT, int *tmp, unsigned **rk, unsigned **rv) {
    const int lane = lane_id(), wave = wave_id();
    const int nw = blockDim.x >> 6;  // <= 16
    const int seg = (((n + nw - 1) / nw) + 63) & ~63;
    const int start = wave * seg;
    const int end = min(n, start + seg);
    const unsigned long long lt = (1ull << lane) - 1ull;
    constexpr int NB = 1 << kRadixBits;
    for (int shift = 0; shift < bits; shift += kRadixBits) {
        for (int t = threadIdx.x; t < NB * 16; t += blockDim.x) histT[t] = 0;
        __syncthreads();
        for (int i = start + lane; i < end; i += 64) {
            const int d = (k0[i] >> shift) & (NB - 1);
            atomicAdd((int *) &histT[d * 16 + wave], 1);
        }
        __syncthreads();
        block_scan_array((int *) histT, NB * 16, tmp);   // exclusive, digit-major then wave: the scatter base of every (digit, wave)
        for (int base = start; base < end; base += 64) {
            const int i = base + lane;
            const bool valid = i < end;
            const unsigned key = valid ? k0[i] : 0u;
            const unsigned val = valid ? v0[i] : 0u;
            const int d = (key >> shift) & (NB - 1);
            unsigned long long m = __ballot(valid);
#pragma unroll
            for (int b = 0; b < kRadixBits; b++) {
                const bool bit = (d >> b) & 1;
                const unsigned long long bal = __ballot(bit);
                m &= bit ? bal : ~bal;
            }
            const int rank = __popcll(m & lt);
            const int cnt = __popcll(m);
            int pos = 0;
            if (valid) pos = histT[d * 16 + wave] + rank;
            __builtin_amdgcn_wave_barrier();
            if (valid) {
                k1[pos] = key;
                v1[pos] = val;
                if (rank == cnt - 1) histT[d * 16 + wave] = pos + 1;
            }
            __builtin_amdgcn_wave_barrier();
        }
        __syncthreads();
        unsigned *t = k0; k0 = k1; k1 = t;
        t = v0; v0 = v1; v1 = t;
    }
    *rk = k0;
    *rv = v0;
}

__device__ __forceinline__ void wave_lds_sync();

// ------------------------------------------------------------------------------------------------------------------
// K1  pyramid level from the previous level: cv::resize INTER_LINEAR, 8UC1, 11-bit fixed point coefficients that the
// host precomputed exactly as OpenCV does (xofs/ialpha, yofs/ibeta).  One thread per output pixel.
// ------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_pyr_resize(FrameSet fs, const LevelGeom *__restrict__ geom, int level,
                                                    const int *__restrict__ xofs, const short *__restrict__ xalpha,
                                                    const int *__restrict__ yofs, const short *__restrict__ ybeta) {
    const LevelGeom g = geom[level];
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y, f = blockIdx.z;
    if (x >= g.w) return;
    int sp;
    const uint8_t *src = level_ptr(fs, geom[level - 1], level - 1, f, &sp);
    const int sw = geom[level - 1].w, sh = geom[level - 1].h;
    uint8_t *dst = fs.pyr + (long long) f * fs.pyr_stride + g.off;
    if (g.area2x) {
        const uint8_t *r0 = src + (long long) (2 * y) * sp + 2 * x, *r1 = r0 + sp;
        dst[(long long) y * g.pitch + x] = (uint8_t) ((r0[0] + r0[1] + r1[0] + r1[1] + 2) >> 2);
        return;
    }
    int sy = yofs[g.ytab + y];
    const int sy0 = min(max(sy, 0), sh - 1), sy1 = min(max(sy + 1, 0), sh - 1);
    const int b0 = ybeta[2 * (g.ytab + y)], b1 = ybeta[2 * (g.ytab + y) + 1];
    const int sx = xofs[g.xtab + x];
    const int sx1 = min(sx + 1, sw - 1);
    const int a0 = xalpha[2 * (g.xtab + x)], a1 = xalpha[2 * (g.xtab + x) + 1];
    const uint8_t *S0 = src + (long long) sy0 * sp, *S1 = src + (long long) sy1 * sp;
    const int H0 = S0[sx] * a0 + S0[sx1] * a1;
    const int H1 = S1[sx] * a0 + S1[sx1] * a1;
    dst[(long long) y * g.pitch + x] = (uint8_t) ((((b0 * (H0 >> 4)) >> 16) + ((b1 * (H1 >> 4)) >> 16) + 2) >> 2);
}

// Tiled variant (the one normally launched).  A workgroup produces one tile of 8192 output pixels (host-built list, PyrTileRec): the source
// region goes to LDS in 16-byte chunks, every lane owns 4 adjacent output columns -- selectors and alpha pairs from its PyrColRec stay in
// registers -- and walks down 8 rows.  Tiles are 256 columns x 32 rows; what is left of a level's width after the 256-column tiles is cut
// into 128- / 64- / 32-column tiles whose waves fold 2 / 4 / 8 rows into one pass (lane = row-in-pass x column group), so that a 522-column
// level does not pay a third 256-column tile for its last 10 columns (752x480 / 1.2: 75 % of the lanes of plain 256-column tiles carry a pixel,
// 95 % of these).  Everything a row or a column needs is a table record (PyrRowRec: scalar loads at fold 1): no clamp, table search or
// coefficient unpacking per wave.  Round 6 measured the previous form at 641 vector instructions per wave of which 384 were the row loop.
constexpr int kPyrWaves = 4, kPyrThreads = 64 * kPyrWaves, kPyrPasses = 8;
static_assert(kPyrPasses * kPyrWaves == kPyrTileRows, "tile rows");
constexpr int kPyrLdsBytes = 16384;
static_assert(pyr_fold_rows(0) * pyr_fold_pitch(0) + 16 <= kPyrLdsBytes && pyr_fold_rows(1) * pyr_fold_pitch(1) + 16 <= kPyrLdsBytes &&
              pyr_fold_rows(2) * pyr_fold_pitch(2) + 16 <= kPyrLdsBytes && pyr_fold_rows(3) * pyr_fold_pitch(3) + 16 <= kPyrLdsBytes, "staging area");
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef u32x4 u32x4_a4 __attribute__((aligned(4)));
typedef unsigned u32x8 __attribute__((ext_vector_type(8)));

// the last chunk of the last row of a caller-owned level-0 buffer: never read past its end
__device__ __noinline__ u32x4 pyr_tail_chunk(const uint8_t *src, unsigned o, unsigned total) {
    u32x4 v = {0u, 0u, 0u, 0u};
    for (unsigned b = 0; b < 16; b++)
        if (o + b < total) v[b >> 2] |= (unsigned) src[o + b] << (8 * (b & 3));
    return v;
}

template <int FL>
__device__ __forceinline__ void pyr_tile(uint8_t *tile, uint8_t *__restrict__ dstf, const int gw, const int gh, const int gpitch, const uint8_t *__restrict__ src,
                                         const int sp, const int sh, const int x0, const int y0, const int sxa, const int nc,
                                         const PyrColRec *__restrict__ cols, const PyrRowRec *__restrict__ rows) {
    constexpr int kLX = 64 >> FL, kPitch = pyr_fold_pitch(FL), kStageLanes = 32 >> FL, kStageRows = kPyrThreads / kStageLanes;
    constexpr int kIters = (pyr_fold_rows(FL) + kStageRows - 1) / kStageRows;
    const int tid = threadIdx.x;
    const int yl = min(y0 + (kPyrTileRows << FL), gh) - 1;
    const int sya = (int) (rows[y0].r01 & 0xffffu), syb = (int) (rows[yl].r01 >> 16);
    const int nr = syb - sya + 1;
    const int lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int yw = y0 + ((wv * kPyrPasses) << FL);   // first row of this wave
    const int cg = lane & (kLX - 1), rj = lane >> (6 - FL);
    const int xb = x0 + 4 * cg;
    const bool colIn = xb < gw;
    // requested before the staging, needed after it: the lane's column record, and at fold 1 the records of the wave's 8 rows -- wave-uniform and
    // adjacent (the host pads a level's table, so no clamp), four wide scalar loads (in assembly: the compiler sinks plain loads into the row loop,
    // where every row would wait for its own)
    const PyrColRec *cr = cols + (min(xb, gw - 1) >> 2);
    const int sx0 = cr->sx0;
    const u32x4 s4 = *(const u32x4 *) cr->sel, a4 = *(const u32x4 *) cr->ap;
    u32x8 R01, R23, R45, R67;
    if constexpr (FL == 0) {
        const PyrRowRec *rp = rows + min(yw, gh - 1);
        asm volatile("s_load_dwordx8 %0, %4, 0x0\n\ts_load_dwordx8 %1, %4, 0x20\n\ts_load_dwordx8 %2, %4, 0x40\n\ts_load_dwordx8 %3, %4, 0x60"
                     : "=&s"(R01), "=&s"(R23), "=&s"(R45), "=&s"(R67) : "s"(rp) : "memory");
    }
    {   // a source row is nc <= kStageLanes chunks: kStageLanes lanes per row, kStageRows rows per round, every load in flight before the first LDS store
        const int c = tid & (kStageLanes - 1), r = tid >> (5 - FL);
        const unsigned total = (unsigned) sh * (unsigned) sp;
        const bool colOk = c < nc;
        const unsigned o = (unsigned) (sya + r) * (unsigned) sp + (unsigned) (sxa + 16 * c);
        u32x4 v[kIters];
#pragma unroll
        for (int it = 0; it < kIters; it++) {
            const int rr = r + it * kStageRows;
            if (colOk && rr < nr) {
                const unsigned oo = o + (unsigned) (it * kStageRows) * (unsigned) sp;
                if (oo + 16 <= total) v[it] = *(const u32x4_a4 *) (src + oo);
                else v[it] = pyr_tail_chunk(src, oo, total);
            }
        }
#pragma unroll
        for (int it = 0; it < kIters; it++) {
            const int rr = r + it * kStageRows;
            if (colOk && rr < nr) *(u32x4 *) (tile + rr * kPitch + 16 * c) = v[it];
        }
    }
    __syncthreads();
    if (yw >= gh) {
        if constexpr (FL == 0) asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(R01), "+s"(R23), "+s"(R45), "+s"(R67));
        return;
    }
    // The four columns of a lane read source bytes sx0 .. sx0 + 7 at most: per visited source row three aligned dwords are shifted to start
    // at sx0 (v_alignbyte), one v_perm_b32 per column picks its (left, right) pixel pair into 16-bit halves and one v_dot2_u32_u16 applies
    // (alpha0, alpha1).  H & ~15 keeps what (H >> 4) keeps: v_mul_hi_u32(H & ~15, beta << 12) = (beta * (H >> 4)) >> 16, one instruction.
    unsigned sel[4], ap[4];
#pragma unroll
    for (int k = 0; k < 4; k++) { sel[k] = s4[k]; ap[k] = a4[k]; }
    const int lx0 = sx0 - sxa;
    const int bd4 = lx0 & ~3;
    const unsigned osh = (unsigned) lx0 & 3u;
    auto hrow = [&](int rowByte, unsigned (&H)[4]) {
        const unsigned *p = (const unsigned *) (tile + rowByte + bd4);
        const unsigned d0 = p[0], d1 = p[1], d2 = p[2];
        const unsigned e0 = __builtin_amdgcn_alignbyte(d1, d0, osh), e1 = __builtin_amdgcn_alignbyte(d2, d1, osh);
#pragma unroll
        for (int k = 0; k < 4; k++)
            H[k] = __builtin_amdgcn_udot2(__builtin_bit_cast(v2u, __builtin_amdgcn_perm(e1, e0, sel[k])), __builtin_bit_cast(v2u, ap[k]), 0u, false) & ~15u;
    };
    auto vout = [&](const unsigned (&H0)[4], const unsigned (&H1)[4], unsigned b0, unsigned b1) -> unsigned {
        unsigned sum[4];   // (b0 * (H0 >> 4) >> 16) + (b1 * (H1 >> 4) >> 16) + 2 <= 1023: the pixel is bits 2 .. 9
#pragma unroll
        for (int k = 0; k < 4; k++) sum[k] = __umulhi(H0[k], b0) + __umulhi(H1[k], b1) + 2u;
        const unsigned t01 = (sum[0] | (sum[1] << 16)) >> 2, t23 = (sum[2] | (sum[3] << 16)) >> 2;
        return __builtin_amdgcn_perm(t23, t01, 0x06040200u);
    };
    if constexpr (FL == 0) {
        asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(R01), "+s"(R23), "+s"(R45), "+s"(R67));
        const unsigned r01s[kPyrPasses] = {R01[0], R01[4], R23[0], R23[4], R45[0], R45[4], R67[0], R67[4]};
        const unsigned b0s[kPyrPasses] = {R01[1], R01[5], R23[1], R23[5], R45[1], R45[5], R67[1], R67[5]};
        const unsigned b1s[kPyrPasses] = {R01[2], R01[6], R23[2], R23[6], R45[2], R45[6], R67[2], R67[6]};
        int have = -1;
        unsigned Hc[4] = {0, 0, 0, 0};
#pragma unroll
        for (int p = 0; p < kPyrPasses; p++) {
            const int y = yw + p;
            if (y >= gh) break;
            const int r0 = (int) (r01s[p] & 0xffffu) - sya, r1 = (int) (r01s[p] >> 16) - sya;
            unsigned H0[4], H1[4];
            if (r0 == have) {   // the lower row of the previous output row: five times out of six at scale 1.2
#pragma unroll
                for (int k = 0; k < 4; k++) H0[k] = Hc[k];
            } else {
                hrow(r0 * kPitch, H0);
            }
            if (r1 == r0) {
#pragma unroll
                for (int k = 0; k < 4; k++) H1[k] = H0[k];
            } else {
                hrow(r1 * kPitch, H1);
            }
            have = r1;
#pragma unroll
            for (int k = 0; k < 4; k++) Hc[k] = H1[k];
            const unsigned out = vout(H0, H1, b0s[p], b1s[p]);
            if (colIn) *(unsigned *) (dstf + (unsigned) y * (unsigned) gpitch + xb) = out;
        }
    } else {
        // folded: the lanes of a pass are on 2^FL different rows, so the records are per lane and nothing is shared between passes
        u32x4 rec[kPyrPasses];
#pragma unroll
        for (int p = 0; p < kPyrPasses; p++) rec[p] = *(const u32x4 *) (rows + min(yw + (p << FL) + rj, gh - 1));
#pragma unroll
        for (int p = 0; p < kPyrPasses; p++) {
            if (yw + (p << FL) >= gh) break;
            const int y = yw + (p << FL) + rj;
            const int r0 = (int) (rec[p][0] & 0xffffu) - sya, r1 = (int) (rec[p][0] >> 16) - sya;
            unsigned H0[4], H1[4];
            hrow(r0 * kPitch, H0);
            hrow(r1 * kPitch, H1);
            const unsigned out = vout(H0, H1, rec[p][1], rec[p][2]);
            if (colIn && y < gh) *(unsigned *) (dstf + (unsigned) y * (unsigned) gpitch + xb) = out;
        }
    }
}

__global__ __launch_bounds__(kPyrThreads) void k_pyr_resize_tiled(FrameSet fs, const LevelGeom *__restrict__ geom, int level, const PyrColRec *__restrict__ cols,
                                                                  const PyrRowRec *__restrict__ rows, const PyrTileRec *__restrict__ tiles) {
    __shared__ __attribute__((aligned(16))) uint8_t tile[kPyrLdsBytes];
    const LevelGeom g = geom[level];
    const int f = blockIdx.y;
    const PyrTileRec t = tiles[g.pyrTile + blockIdx.x];
    int sp;
    const uint8_t *src = level_ptr(fs, geom[level - 1], level - 1, f, &sp);
    const int sh = geom[level - 1].h;
    uint8_t *dstf = fs.pyr + (long long) f * fs.pyr_stride + g.off;
    cols += g.pyrCol;
    rows += g.pyrRow;
    switch (t.foldLog2) {   // (uniform)
    case 0: pyr_tile<0>(tile, dstf, g.w, g.h, g.pitch, src, sp, sh, t.x0, t.y0, t.sxa, t.nc, cols, rows); break;
    case 1: pyr_tile<1>(tile, dstf, g.w, g.h, g.pitch, src, sp, sh, t.x0, t.y0, t.sxa, t.nc, cols, rows); break;
    case 2: pyr_tile<2>(tile, dstf, g.w, g.h, g.pitch, src, sp, sh, t.x0, t.y0, t.sxa, t.nc, cols, rows); break;
    default: pyr_tile<3>(tile, dstf, g.w, g.h, g.pitch, src, sp, sh, t.x0, t.y0, t.sxa, t.nc, cols, rows); break;
    }
}

// Strip variant (a few frames at a time: one Tracking frame, a stereo pair): the WHOLE chain of one frame in ONE launch.  Seven dependent
// launches of a 752x480 frame cost ~6 us each whatever they compute; here a workgroup owns a horizontal strip of the LAST level and
// produces, level by level in LDS, every row of the levels above it that the strip depends on (the host plans the row ranges from the
// yofs tables: PyrStripPlan) -- neighbouring strips recompute each other's halo rows instead of waiting for each other, so no workgroup
// ever reads what another one wrote.  Rows a strip owns (a partition of every level) also go to the pyramid slab.  Same integers as
// k_pyr_resize_tiled: same coefficient tables, same row sums, same rounding.  Level l - 1 and level l ping-pong between two LDS regions.
// The kernel is a chain of short dependent phases, so nothing inside the level loop waits for global memory: the row coefficients of every row the
// strip produces (source rows, beta pair) go to LDS together with level 0; a thread's column coefficients (source column, alpha pair of the four
// columns it produces on a level -- the same for every row) are read from the global tables one level AHEAD, while the level in hand is computed
// (until round 6 a table of all levels' columns sat in LDS: 8 bytes per column, 55 KB of the 160 for 1920x1080, which therefore had no one-launch
// plan); sizes and row ranges are scalar loads of two small tables.
constexpr int kPyrStripThreads = kPyrStripMaxThreads;

__global__ __launch_bounds__(kPyrStripThreads) void k_pyr_strips(FrameSet fs, int nlevels, const PyrStripPlan *__restrict__ plans,
                                                                 const PyrStripLevel *__restrict__ levels, int offA, int offB,
                                                                 const int *__restrict__ xofs, const short *__restrict__ xalpha,
                                                                 const int *__restrict__ yofs, const short *__restrict__ ybeta) {
    extern __shared__ __attribute__((aligned(16))) uint8_t pyrLds[];
    uint2 *rowTab = (uint2 *) pyrLds;                 // (r0 | r1 << 16, beta0 | beta1 << 16) per produced row, levels 1 .. in order
    const int tid = threadIdx.x, f = blockIdx.y;
    const PyrStripPlan *__restrict__ pl = plans + blockIdx.x;
    // this thread's four columns of level l: (source column, alpha0 | alpha1 << 16) each, straight from the coefficient tables
    auto load_cols = [&](int l, uint2 (&cv)[4], bool *active) {
        const int w = levels[l].w, nq = (w + 3) >> 2, nrp = kPyrStripThreads / nq;
        const int rp = tid / nq, cq = tid - rp * nq;
        *active = rp < nrp;
        const int xt = levels[l].xtab;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int gx = xt + min(4 * cq + k, w - 1);
            cv[k] = make_uint2((unsigned) xofs[gx], *(const unsigned *) (xalpha + 2 * gx));
        }
    };
    // Everything the strip needs from global memory is requested before the first LDS write waits for any of it: this thread's row
    // coefficients (one row of one level), the first 4096 column coefficients, the first eight dwords per thread of level 0.
    bool rowHas = false;
    int rowSy = 0, rowSh = 1, rowSca = 0;
    unsigned rowBeta = 0;
    {
        int i = tid, y = 0, ytab = 0;
        for (int l = 1; l < nlevels; l++) {   // uniform walk over the levels (scalar loads), selects per thread
            const int ca = (int) (pl->lv[l].x & 0xFFFFu), n = (int) (pl->lv[l].x >> 16) - ca;
            const bool hit = i >= 0 && i < n;
            y = hit ? ca + i : y;
            ytab = hit ? levels[l].ytab : ytab;
            rowSh = hit ? levels[l - 1].h : rowSh;
            rowSca = hit ? (int) (pl->lv[l - 1].x & 0xFFFFu) : rowSca;
            rowHas |= hit;
            i -= n;
        }
        if (rowHas) {
            rowSy = yofs[ytab + y];
            rowBeta = *(const unsigned *) (ybeta + 2 * (ytab + y));
        }
    }
    uint2 colNext[4];                                 // level 1's, requested with the row coefficients, ahead of the image
    bool actNext;
    load_cols(1, colNext, &actNext);
    {   // level 0 rows [ca, cb) -> region A, whole rows, aligned dwords of the row
        const uint8_t *src = fs.img0 + (long long) f * fs.img0_stride;
        const int sp = fs.img0_pitch;
        const int w0 = levels[0].w, h0 = levels[0].h;
        const int ca = (int) (pl->lv[0].x & 0xFFFFu), nr = (int) (pl->lv[0].x >> 16) - ca;
        const int nd = (w0 + 3) >> 2, P = pyr_strip_lds_pitch(w0) >> 2;
        const unsigned total = (unsigned) h0 * (unsigned) sp;
        unsigned *A = (unsigned *) (pyrLds + offA);
        int r = tid / nd, c = tid - r * nd;
        const int sr = kPyrStripThreads / nd, sc = kPyrStripThreads - sr * nd;
        constexpr int kU = 8;
        bool first = true;
        for (int i0 = 0; i0 < nd * nr; i0 += kPyrStripThreads * kU) {
            unsigned v[kU];
            int dst[kU];
#pragma unroll
            for (int u = 0; u < kU; u++) {
                const bool in = r < nr;
                const int rr = in ? r : nr - 1, cc = in ? c : nd - 1;
                const unsigned off = (unsigned) (ca + rr) * (unsigned) sp + (unsigned) (4 * cc);
                if (off + 4 <= total) v[u] = *(const unsigned *) (src + off);
                else {
                    v[u] = 0;
                    for (int b = 0; b < 4; b++) if (off + b < total) v[u] |= (unsigned) src[off + b] << (8 * b);
                }
                dst[u] = rr * P + cc;
                r += sr; c += sc;
                if (c >= nd) { c -= nd; r++; }
            }
            if (first) {   // the tables, requested before the image
                first = false;
                if (rowHas) {
                    const unsigned r0 = (unsigned) (min(max(rowSy, 0), rowSh - 1) - rowSca), r1 = (unsigned) (min(max(rowSy + 1, 0), rowSh - 1) - rowSca);
                    rowTab[tid] = make_uint2(r0 | (r1 << 16), rowBeta);
                }
            }
#pragma unroll
            for (int u = 0; u < kU; u++) A[dst[u]] = v[u];
        }
    }
    __syncthreads();
    int rowOff = 0;
    for (int l = 1; l < nlevels; l++) {
        // sizes and row ranges: scalar loads of two small tables that the staging phase has already pulled into the scalar cache
        const int w = levels[l].w, pitch = levels[l].pitch;
        const unsigned goff = (unsigned) levels[l].off;
        const int sw = levels[l - 1].w;
        const uint8_t *sb = pyrLds + ((l - 1) & 1 ? offB : offA);
        uint8_t *db = pyrLds + (l & 1 ? offB : offA);
        const int Ps = pyr_strip_lds_pitch(sw), Pd = pyr_strip_lds_pitch(w);
        const uint2 pv = pl->lv[l];
        const int ca = (int) (pv.x & 0xFFFFu), cb = (int) (pv.x >> 16), wa = (int) (pv.y & 0xFFFFu), wb = (int) (pv.y >> 16);
        const int nq = (w + 3) >> 2, nrp = kPyrStripThreads / nq;
        const int rp = tid / nq, cq = tid - rp * nq;
        uint2 colCur[4];
#pragma unroll
        for (int k = 0; k < 4; k++) colCur[k] = colNext[k];
        if (l + 1 < nlevels) load_cols(l + 1, colNext, &actNext);   // in flight while this level is computed
        if (rp < nrp) {   // this thread: column quad cq of rows ca + rp, ca + rp + nrp, ...
            const int xb = 4 * cq;
            unsigned sel[4], ap[4];
            const int lx0 = (int) colCur[0].x;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const uint2 cc = colCur[k];
                const int sx1 = min((int) cc.x + 1, sw - 1);
                sel[k] = (unsigned) ((int) cc.x - lx0) | 0x0C00u | ((unsigned) (sx1 - lx0) << 16) | 0x0C000000u;
                ap[k] = cc.y;
            }
            const int bd4 = lx0 & ~3;
            const unsigned osh = (unsigned) lx0 & 3u;
            uint8_t *dstf = fs.pyr + (long long) f * fs.pyr_stride + goff;
            auto hrow = [&](int r, unsigned (&H)[4]) {
                const unsigned *p = (const unsigned *) (sb + r * Ps + bd4);
                const unsigned d0 = p[0], d1 = p[1], d2 = p[2];
                const unsigned e0 = __builtin_amdgcn_alignbyte(d1, d0, osh), e1 = __builtin_amdgcn_alignbyte(d2, d1, osh);
#pragma unroll
                for (int k = 0; k < 4; k++)
                    H[k] = __builtin_amdgcn_udot2(__builtin_bit_cast(v2u, __builtin_amdgcn_perm(e1, e0, sel[k])), __builtin_bit_cast(v2u, ap[k]), 0u, false) & ~15u;
            };
            // (beta * (H >> 4)) >> 16 == v_mul_hi_u32(H & ~15, beta << 12), four pixels packed by two shifts and a byte permute: as k_pyr_resize_tiled
            auto vout = [&](const unsigned (&H0)[4], const unsigned (&H1)[4], unsigned b01) -> unsigned {
                const unsigned b0 = (b01 & 0xFFFFu) << 12, b1 = (b01 >> 16) << 12;
                unsigned sum[4];
#pragma unroll
                for (int k = 0; k < 4; k++) sum[k] = __umulhi(H0[k], b0) + __umulhi(H1[k], b1) + 2u;
                const unsigned t01 = (sum[0] | (sum[1] << 16)) >> 2, t23 = (sum[2] | (sum[3] << 16)) >> 2;
                return __builtin_amdgcn_perm(t23, t01, 0x06040200u);
            };
            for (int y = ca + rp; y < cb; y += 2 * nrp) {   // two rows per turn: their LDS reads travel together
                const int y2 = y + nrp;
                const bool two = y2 < cb;
                const uint2 rcA = rowTab[rowOff + y - ca], rcB = rowTab[rowOff + (two ? y2 : y) - ca];
                unsigned HA0[4], HA1[4], HB0[4], HB1[4];
                hrow((int) (rcA.x & 0xFFFFu), HA0);
                hrow((int) (rcA.x >> 16), HA1);
                hrow((int) (rcB.x & 0xFFFFu), HB0);
                hrow((int) (rcB.x >> 16), HB1);
                const unsigned outA = vout(HA0, HA1, rcA.y), outB = vout(HB0, HB1, rcB.y);
                *(unsigned *) (db + (y - ca) * Pd + xb) = outA;
                if (y >= wa && y < wb) *(unsigned *) (dstf + (unsigned) y * (unsigned) pitch + xb) = outA;
                if (two) {
                    *(unsigned *) (db + (y2 - ca) * Pd + xb) = outB;
                    if (y2 >= wa && y2 < wb) *(unsigned *) (dstf + (unsigned) y2 * (unsigned) pitch + xb) = outB;
                }
            }
        }
        rowOff += cb - ca;
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------------------------
// K2  FAST-9/16 per 30-px cell.  One wave per (cell, frame): the (wCell+6)x(hCell+6) window is staged in LDS, corners are
// found at minTh (byte-sliced test, four pixels per lane), their score (max arc margin - 1 == cv::FAST's cornerScore) is
// written to an LDS score map, then the cell-local 3x3 NMS is evaluated for BOTH thresholds and the minTh result is used
// only when the iniTh result is empty -- exactly the reference's "FAST(ini); if empty FAST(min)".
// The per-pixel helpers below (fast9_test / fast9_arc_score) serve the listed corners and the dense fallback.
// ------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool has_arc9(unsigned m16) {
    unsigned m = m16 | (m16 << 16);
    unsigned r = m & (m >> 1);
    r &= r >> 2;
    r &= r >> 4;
    r &= m >> 8;
    return (r & 0xFFFFu) != 0;
}

// 1 = bright corner, 2 = dark corner, 0 = none, at threshold t (9 contiguous ring pixels > v+t or < v-t).
// Ring pixels k and k+8 share one register as two 16-bit lanes, so one packed subtract (v_pk_sub_i16) tests two ring positions
// per polarity; the sign bits (bit 15 / 31) are shifted into place as they arrive: after 8 steps ring k sits at bit 8+k of the low
// half and ring 8+k at bit 24+k.  No compare -> SGPR -> select round trip (which also costs hazard nops on gfx9).
typedef short v2s __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pk_sub16(unsigned a, unsigned b) {
    const v2s r = __builtin_bit_cast(v2s, a) - __builtin_bit_cast(v2s, b);
    return __builtin_bit_cast(unsigned, r);
}
typedef unsigned short v2u __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned rot16pk(unsigned a, int sh) {   // rotate both 16-bit halves right by sh
    const v2u x = __builtin_bit_cast(v2u, a);
    const v2u r = (x >> (v2u) (unsigned short) sh) | (x << (v2u) (unsigned short) (16 - sh));
    return __builtin_bit_cast(unsigned, r);
}
__device__ __forceinline__ int fast9_test(const uint8_t *c, int tp, int t) {
    const int v = c[0];
    const int t2 = 2 * tp, t3 = 3 * tp;
    const unsigned hi = (unsigned) (v + t) * 0x10001u;                 // (v+t, v+t)
    const unsigned lo = ((unsigned) (v - t) & 0xFFFFu) * 0x10001u;     // (v-t, v-t) as two's-complement halves
    unsigned accB = 0, accD = 0;
#define RING_PAIR(a, b)                                                                \
    {                                                                                  \
        const unsigned pk = (unsigned) c[a] | ((unsigned) c[b] << 16);                 \
        accB = (accB >> 1) | (pk_sub16(hi, pk) & 0x80008000u);   /* ring > v + t */    \
        accD = (accD >> 1) | (pk_sub16(pk, lo) & 0x80008000u);   /* ring < v - t */    \
    }
    RING_PAIR(t3, -t3)            // 0, 8
    RING_PAIR(t3 + 1, -t3 - 1)    // 1, 9
    RING_PAIR(t2 + 2, -t2 - 2)    // 2, 10
    RING_PAIR(tp + 3, -tp - 3)    // 3, 11
    RING_PAIR(3, -3)              // 4, 12
    RING_PAIR(-tp + 3, tp - 3)    // 5, 13
    RING_PAIR(-t2 + 2, t2 - 2)    // 6, 14
    RING_PAIR(-t3 + 1, t3 - 1)    // 7, 15
#undef RING_PAIR
    // both 16-bit ring masks in one register (bright | dark << 16): one v_perm_b32 gathers the four mask bytes, and the
    // "9 contiguous of the circular 16" test runs on both halves at once with packed 16-bit rotates (runs >= 2, 4, 8, then 9)
    const unsigned m = __builtin_amdgcn_perm(accD, accB, 0x07050301u);
    unsigned r = m & rot16pk(m, 1);
    r &= rot16pk(r, 2);
    r &= rot16pk(r, 4);
    r &= rot16pk(m, 8);
    return (r & 0xFFFFu) ? 1 : (r ? 2 : 0);
}

// cv::FAST cornerScore<16> for a pixel known to be a corner of polarity `pol`: max over the 16 arcs of 9 of the minimum
// margin, minus 1 (== the largest threshold for which the pixel is still a corner).  Signed margins e[k] = +-(ring_k - centre)
// live as 16-bit pairs (e[k] | e[k+8] << 16), so every packed min serves two arcs; "k+8" neighbours are the swapped halves.
__device__ __forceinline__ unsigned pk_min16(unsigned a, unsigned b) {
    return __builtin_bit_cast(unsigned, __builtin_elementwise_min(__builtin_bit_cast(v2s, a), __builtin_bit_cast(v2s, b)));
}
__device__ __forceinline__ unsigned pk_max16(unsigned a, unsigned b) {
    return __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(v2s, a), __builtin_bit_cast(v2s, b)));
}
__device__ __forceinline__ unsigned pk_mul16(unsigned a, unsigned b) {
    return __builtin_bit_cast(unsigned, __builtin_bit_cast(v2s, a) * __builtin_bit_cast(v2s, b));
}
__device__ __forceinline__ unsigned swap16(unsigned a) {   // a half swap feeding a packed min/max folds into its op_sel bits
    const v2s x = __builtin_bit_cast(v2s, a);
    return __builtin_bit_cast(unsigned, __builtin_shufflevector(x, x, 1, 0));
}
__device__ __forceinline__ int fast9_arc_score(const uint8_t *c, int tp, int pol) {
    const int t2 = 2 * tp, t3 = 3 * tp;
    const unsigned vv = (unsigned) c[0] * 0x10001u;
    const unsigned sg = pol == 2 ? 0xFFFFFFFFu : 0x00010001u;   // -1 / +1 in both halves
    unsigned P[8];
#define MARGIN_PAIR(k, a, b) P[k] = pk_mul16(pk_sub16((unsigned) c[a] | ((unsigned) c[b] << 16), vv), sg);
    MARGIN_PAIR(0, t3, -t3) MARGIN_PAIR(1, t3 + 1, -t3 - 1) MARGIN_PAIR(2, t2 + 2, -t2 - 2) MARGIN_PAIR(3, tp + 3, -tp - 3)
    MARGIN_PAIR(4, 3, -3) MARGIN_PAIR(5, -tp + 3, tp - 3) MARGIN_PAIR(6, -t2 + 2, t2 - 2) MARGIN_PAIR(7, -t3 + 1, t3 - 1)
#undef MARGIN_PAIR
    unsigned M2[8], M4[8], M8[8];
#pragma unroll
    for (int k = 0; k < 7; k++) M2[k] = pk_min16(P[k], P[k + 1]);                 // (m2[k], m2[k+8])
    M2[7] = pk_min16(P[7], swap16(P[0]));
#pragma unroll
    for (int k = 0; k < 6; k++) M4[k] = pk_min16(M2[k], M2[k + 2]);
    M4[6] = pk_min16(M2[6], swap16(M2[0]));
    M4[7] = pk_min16(M2[7], swap16(M2[1]));
#pragma unroll
    for (int k = 0; k < 4; k++) M8[k] = pk_min16(M4[k], M4[k + 4]);
#pragma unroll
    for (int k = 4; k < 8; k++) M8[k] = pk_min16(M4[k], swap16(M4[k - 4]));
    unsigned best = 0;   // margins of a corner's winning arc are positive; max(0, ...) as the scalar form's `a = 0` start
#pragma unroll
    for (int k = 0; k < 8; k++) best = pk_max16(best, pk_min16(M8[k], swap16(P[k])));   // min(m8[k], e[k+8]) | min(m8[k+8], e[k])
    const int lo = (short) (best & 0xFFFFu), hi = (short) (best >> 16);
    return max(lo, hi) - 1;
}

// LDS pitch of a cell's score map = the window pitch (wCell + 2 columns used, the window pitch is >= wCell + 7)
struct FastGroupBases { int v[kMaxLevels]; };   // first 2x2 cell group of every level inside a frame (LevelGeom::groupBase), by value
constexpr int kCornerCap = 512;  // corners listed per cell before the dense fallback takes over

// ------------------------------------------------------------------------------------------------------------------
// K2q  The same cell loop with FOUR pixels per lane in pass 1 (byte-sliced FAST-9 test).
// Every wave stages its own cell window into LDS, shifted so that tested pixel x of the cell sits at column x + 4: the four centres of
// quad q are then exactly dword q + 1 of their row, and the ring pixel (dx, dy) of all four is one v_alignbyte_b32 of two neighbouring
// dwords with a compile-time shift.  One v_lerp_u8 compares four bytes at once (bit 7 of (a + b + r) >> 1 is the carry of the 9-bit sum):
//   bright_k  = bit7 lerp(ring_k, 255 - min(c + t, 255), 0)   <=> ring_k >  c + t
//   !dark_k   = bit7 lerp(ring_k, 255 - max(c - t, 0),   1)   <=> ring_k >= c - t
// and "9 contiguous of the circular 16" is evaluated bit-sliced over the 16 ring registers (3-input AND / OR trees: an arc of 9 is three
// arcs of 3), so no per-pixel mask is ever assembled.  The low 7 bits of every byte are don't-care throughout.  Results (polarity per
// pixel) are identical to fast9_test; the score / NMS passes work per listed corner.
// ------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned fast9_quad(const unsigned *w, int pd, int t) {
#define QROW(r) const unsigned a##r = w[(r) * pd], b##r = w[(r) * pd + 1], c##r = w[(r) * pd + 2];
    QROW(0) QROW(1) QROW(2) QROW(3) QROW(4) QROW(5) QROW(6)
#undef QROW
#define AB(hi, lo, sh) __builtin_amdgcn_alignbyte(hi, lo, sh)
    unsigned R[16];
    R[8] = b0;            R[9] = AB(b0, a0, 3);   R[7] = AB(c0, b0, 1);   // dy = -3: dx 0, -1, +1
    R[10] = AB(b1, a1, 2); R[6] = AB(c1, b1, 2);                          // dy = -2: dx -2, +2
    R[11] = AB(b2, a2, 1); R[5] = AB(c2, b2, 3);                          // dy = -1: dx -3, +3
    R[12] = AB(b3, a3, 1); R[4] = AB(c3, b3, 3);                          // dy =  0
    R[13] = AB(b4, a4, 1); R[3] = AB(c4, b4, 3);                          // dy = +1
    R[14] = AB(b5, a5, 2); R[2] = AB(c5, b5, 2);                          // dy = +2
    R[0] = b6;            R[15] = AB(b6, a6, 3);  R[1] = AB(c6, b6, 1);   // dy = +3
#undef AB
    // per-byte saturated thresholds on the complemented centre: 255 - min(c + t, 255) = max(nc - t, 0); 255 - max(c - t, 0) = min(nc + t, 255)
    const unsigned nc = ~b3;
    const unsigned ev = nc & 0x00FF00FFu, od = (nc >> 8) & 0x00FF00FFu;
    const unsigned tt = (unsigned) t * 0x10001u;
    const v2u tv = __builtin_bit_cast(v2u, tt), cap = __builtin_bit_cast(v2u, 0x00FF00FFu);
    const unsigned nhiE = __builtin_bit_cast(unsigned, __builtin_elementwise_sub_sat(__builtin_bit_cast(v2u, ev), tv));
    const unsigned nhiO = __builtin_bit_cast(unsigned, __builtin_elementwise_sub_sat(__builtin_bit_cast(v2u, od), tv));
    const unsigned nloE = __builtin_bit_cast(unsigned, __builtin_elementwise_min(__builtin_bit_cast(v2u, ev) + tv, cap));
    const unsigned nloO = __builtin_bit_cast(unsigned, __builtin_elementwise_min(__builtin_bit_cast(v2u, od) + tv, cap));
    const unsigned nhi = nhiE | (nhiO << 8), nlo = nloE | (nloO << 8);
    unsigned B[16], N[16];
#pragma unroll
    for (int k = 0; k < 16; k++) {
        B[k] = __builtin_amdgcn_lerp(R[k], nhi, 0u);
        N[k] = __builtin_amdgcn_lerp(R[k], nlo, 0x01010101u);
    }
    // "9 contiguous of the circular 16", bit-sliced: arcs of 3, then arcs of 9 = three arcs of 3, then the reduction over the 16 start
    // positions -- 80 three-input operations.  Every one is written as v_bitop3_b32: on gfx950 that instruction issues at the full rate
    // (2 cycles per wave64) while v_or3_b32 / v_and_or_b32, which the compiler would pick for `a | b | c`, issue at half rate like most
    // three-operand integer instructions (profiles/micro/r02_valu_issue_rates.txt).
#define AND3(a, b, c) __builtin_amdgcn_bitop3_b32(a, b, c, 0x80)
#define OR3(a, b, c) __builtin_amdgcn_bitop3_b32(a, b, c, 0xFE)
    unsigned A3[16], O3[16], A9[16], O9[16];
#pragma unroll
    for (int k = 0; k < 16; k++) {
        A3[k] = AND3(B[k], B[(k + 1) & 15], B[(k + 2) & 15]);
        O3[k] = OR3(N[k], N[(k + 1) & 15], N[(k + 2) & 15]);
    }
#pragma unroll
    for (int k = 0; k < 16; k++) {
        A9[k] = AND3(A3[k], A3[(k + 3) & 15], A3[(k + 6) & 15]);
        O9[k] = OR3(O3[k], O3[(k + 3) & 15], O3[(k + 6) & 15]);
    }
    unsigned bright = OR3(A9[0], A9[1], A9[2]), ndark = AND3(O9[0], O9[1], O9[2]);
#pragma unroll
    for (int k = 3; k < 15; k += 2) {
        bright = OR3(bright, A9[k], A9[k + 1]);
        ndark = AND3(ndark, O9[k], O9[k + 1]);
    }
    bright |= A9[15];
    ndark &= O9[15];
#undef AND3
#undef OR3
    const unsigned fb = bright & 0x80808080u, fd = ~(ndark | bright) & 0x80808080u;
    return (fb >> 7) | (fd >> 6);   // per byte: 1 bright corner, 2 dark corner, 0 none
}

// a / x for 0 <= a <= 64, 1 <= x <= 64 as (a * kRcp16[x]) >> 16 (exact in that range): lane -> (row, column) splits without the
// 30-instruction integer division sequence; x is wave-uniform, so the table read is one scalar load.
__constant__ unsigned kRcp16[65] = {0, 65537, 32769, 21846, 16385, 13108, 10923, 9363, 8193, 7282, 6554, 5958, 5462, 5042, 4682, 4370, 4097, 3856, 3641, 3450, 3277, 3121, 2979, 2850, 2731, 2622, 2521, 2428, 2341, 2260, 2185, 2115, 2049, 1986, 1928, 1873, 1821, 1772, 1725, 1681, 1639, 1599, 1561, 1525, 1490, 1457, 1425, 1395, 1366, 1338, 1311, 1286, 1261, 1237, 1214, 1192, 1171, 1150, 1130, 1111, 1093, 1075, 1058, 1041, 1025};
__device__ __forceinline__ int div_small(int a, unsigned m) { return (int) (__umul24((unsigned) a, m) >> 16); }   // both < 2^24: full-rate multiply

// 3x3 NMS of a listed corner on the score map: survivor at minTh (strictly above all 8 neighbours) and at iniTh.  A corner of
// FAST(iniTh) has score >= iniTh and competes with the neighbours of score >= iniTh only -- but a neighbour below iniTh <= s cannot
// beat s, so the iniTh survivor test is "minTh survivor and s >= iniTh".
__device__ __forceinline__ int nms_flags(const uint8_t *sp, int iniTh, int kSP) {
    const int s = sp[0];
    int nmax = max(max((int) sp[-kSP - 1], (int) sp[-kSP]), max((int) sp[-kSP + 1], (int) sp[-1]));
    nmax = max(nmax, max(max((int) sp[1], (int) sp[kSP - 1]), max((int) sp[kSP], (int) sp[kSP + 1])));
    const int kMin = s > nmax;
    return (kMin && s >= iniTh ? 1 : 0) | (kMin << 1);
}

// One workgroup = a 2x2 group of cells of one level, one WAVE per cell, no block-level synchronisation: every wave stages its own
// window.  Pass 1 lists the quads that hold a corner (raster order); the quad list is expanded into the corner list (raster order);
// then score per corner -> private score map, 3x3 NMS for both thresholds, cell-empty fallback and ordered output.  The common case
// (<= 64 corners in the cell) keeps everything after the expansion in registers.  kP = compile-time window pitch (0: run-time).
// one cell = one wave's unit of work; returns when the cell is done (all paths), so that a wave can take several cells in a row
// Everything after the window is staged: pass 1 (quads), expansion, score, NMS, output -- see fast_cell.  The window holds image pixel
// (cell-tested pixel x, row y) at byte (y + 3) * P + x + 4.
template <int kP, bool kIniFirst>
__device__ __forceinline__ void fast_cell_process(uint8_t *win, uint8_t *smap, unsigned short *clist, const int P, const int dw, const int dh,
                                                  const int iniTh, const int minTh, unsigned short *cnt_out,
                                                  unsigned *__restrict__ out, const int grp, const int lane, unsigned *__restrict__ stats, PhaseClk &pc) {
    const int kSP = P;   // score-map pitch
    unsigned *qlist = (unsigned *) smap;
    const unsigned long long lane_lt = (1ull << lane) - 1ull;
    // Threshold plan of the cell (identical results, different cost):
    //   kIniFirst = false  ONE pass at minTh; the 3x3 NMS is evaluated for both thresholds at once (a corner of FAST(iniTh) has score
    //                      >= iniTh and only neighbours of score >= iniTh compete in FAST(iniTh) -- but a weaker neighbour cannot beat it, so
    //                      "survives at minTh and score >= iniTh" is the FAST(iniTh) survivor test).  Every corner of FAST(minTh) is scored.
    //   kIniFirst = true   the reference's own order: a pass at iniTh, and only a cell that keeps nothing runs the pass at minTh.  Corner-
    //                      dense content (hundreds of FAST(minTh) corners per cell, of which a handful reach iniTh) scores a fraction of the
    //                      corners; the price is a second pass 1 in the cells that are empty at iniTh.  The host picks the plan per batch
    //                      from the statistics the kernel leaves in `stats` (sampled workgroups).
    const int nPass = (kIniFirst && iniTh != minTh) ? 2 : 1;
    // every 16th cell group of the frame reports (the group index runs over all levels, rows and columns, so the sample is spread over the pyramid)
    const bool sampled = stats != nullptr && (grp & 15) == 0;
    unsigned *st = sampled ? stats + 8 * ((grp >> 4) & 63) : nullptr;   // kFastStatWords: 64 records of 8 words
#pragma unroll 1
    for (int pass = 0; pass < nPass; pass++) {
        const int th = nPass == 2 && pass == 0 ? iniTh : minTh;   // threshold of this pass' corner test
        const bool lastPass = pass == nPass - 1;
        // ---- pass 1: four pixels per lane, quads in raster order; quads that hold a corner are listed as  pol bytes | y << 2 | q << 10 ----
        int nQ = 0;
        const int nquadsAll = ((dw + 3) >> 2) * dh;
        {
            const int nq = (dw + 3) >> 2, nquads = nq * dh;
            const unsigned lastMask = 0x03030303u >> (8 * (4 * nq - dw));
            const unsigned mnq = kRcp16[nq];
            int y = div_small(lane, mnq), q = lane - __mul24(y, nq);
            const int qy = div_small(64, mnq), qx = 64 - qy * nq;
            for (int base = 0; base < nquads; base += 64) {
                unsigned pb = 0;
                if (base + lane < nquads) {
                    pb = fast9_quad((const unsigned *) (win + __mul24(y, P)) + q, P >> 2, th);
                    pb &= (q == nq - 1) ? lastMask : 0x03030303u;
                }
                const unsigned long long m = __ballot(pb != 0);
                if (m) {
                    if (pb) qlist[nQ + (int) __builtin_amdgcn_mbcnt_hi((unsigned) (m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned) m, 0u))] =
                                pb | ((unsigned) y << 2) | ((unsigned) q << 10);
                    nQ += __popcll(m);
                }
                y += qy; q += qx;
                if (q >= nq) { q -= nq; y++; }
            }
        }
        if (sampled && lane == 0) { atomicAdd(&st[4], (unsigned) nQ); atomicAdd(&st[5], (unsigned) nquadsAll); atomicAdd(&st[7], 1u); }   // corner-bearing quads / quads / pass-1 runs
        wave_lds_sync();
        pc.mark(1);
        pc.bump();
        // ---- expansion: quad list -> corner list (y << 8 | x << 2 | polarity), still raster order ----
        int ncorn = 0;
        bool overflow = false;
        for (int qb = 0; qb < nQ; qb += 64) {
            const unsigned rec = qb + lane < nQ ? qlist[qb + lane] : 0u;
            unsigned rank = 0;
            int add = 0;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const unsigned long long m = __ballot(((rec >> (8 * j)) & 3u) != 0);
                rank = __builtin_amdgcn_mbcnt_hi((unsigned) (m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned) m, rank));
                add += __popcll(m);
            }
            if (ncorn + add > kCornerCap) { overflow = true; break; }
            int pos = ncorn + (int) rank;
            const unsigned yx = ((rec & 0xFCu) << 6) | ((rec & 0x3C00u) >> 6);   // y << 8 | 4q << 2
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const unsigned pj = (rec >> (8 * j)) & 3u;
                if (pj) clist[pos++] = (unsigned short) (yx | (j << 2) | pj);
            }
            ncorn += add;
        }
        wave_lds_sync();
        pc.mark(2);
        // the quad list is consumed: its bytes become the (zeroed) score map
        for (int idx = lane; idx < ((dh + 2) * kSP + 15) / 16; idx += 64) ((uint4 *) smap)[idx] = make_uint4(0, 0, 0, 0);
        wave_lds_sync();
        pc.mark(3);
        if (sampled && lane == 0 && !kIniFirst && ncorn > 64) atomicAdd(&st[2], (unsigned) ((min(ncorn, kCornerCap) - 1) >> 6));   // score rounds beyond the first
        int total = 0;          // keypoints this pass keeps
        bool usedMin = false;   // the single-pass plan fell back to the minTh survivors
        if (!overflow && ncorn <= 64) {
            // ---- common case: one corner per lane, in registers ----
            const bool have = lane < ncorn;
            const int e = have ? clist[lane] : 0;
            const int y = e >> 8, x = (e >> 2) & 63;
            uint8_t *sp = &smap[(y + 1) * kSP + x + 1];
            int sc = 0;
            if (have) {
                sc = fast9_arc_score(&win[__mul24(y + 3, P) + x + 4], P, e & 3);
                sp[0] = (uint8_t) sc;
            }
            wave_lds_sync();
            pc.mark(4);
            const int fl = have ? nms_flags(sp, iniTh, kSP) : 0;
            bool keep;
            if (kIniFirst) keep = (fl & 2) != 0;                      // survivors of FAST(th)
            else {
                const bool anyIni = __ballot(fl & 1) != 0;
                keep = (fl & (anyIni ? 1 : 2)) != 0;                  // the iniTh survivors, or the minTh survivors when the cell is empty at iniTh
                usedMin = !anyIni;
            }
            const unsigned long long m = __ballot(keep);
            total = __popcll(m);
            if (total > 0 || lastPass) {
                if (keep) out[__popcll(m & lane_lt)] = (unsigned) (x + 3) | ((unsigned) (y + 3) << 8) | ((unsigned) sc << 16);
            }
        } else if (!overflow) {
            // score of every listed corner -> private score map
            for (int qb = 0; qb < ncorn; qb += 64) {
                const int qi = qb + lane;
                if (qi < ncorn) {
                    const int e = clist[qi];
                    const int y = e >> 8, x = (e >> 2) & 63;
                    smap[(y + 1) * kSP + x + 1] = (uint8_t) fast9_arc_score(&win[(y + 3) * P + x + 4], P, e & 3);
                }
            }
            wave_lds_sync();
            pc.mark(4);
            // 3x3 NMS at both thresholds over the corner list
            int nIni = 0;
            for (int qb = 0; qb < ncorn; qb += 64) {
                const int qi = qb + lane;
                int fl = 0;
                if (qi < ncorn) {
                    const int e = clist[qi];
                    const int y = e >> 8, x = (e >> 2) & 63;
                    fl = nms_flags(&smap[(y + 1) * kSP + x + 1], iniTh, kSP);
                    clist[qi] = (unsigned short) ((e & ~3) | fl);   // polarity no longer needed: keep the flags
                }
                nIni += __popcll(__ballot(fl & 1));
            }
            wave_lds_sync();
            // output: the list is in raster order
            const int want = kIniFirst ? 2 : (nIni > 0 ? 1 : 2);
            usedMin = !kIniFirst && nIni == 0;
            for (int qb = 0; qb < ncorn; qb += 64) {
                const int qi = qb + lane;
                const int e = qi < ncorn ? clist[qi] : 0;
                const bool keep = (e & want) != 0;
                const unsigned long long m = __ballot(keep);
                if (keep) {
                    const int y = e >> 8, x = (e >> 2) & 63;
                    const unsigned sv = smap[(y + 1) * kSP + x + 1];
                    out[total + __popcll(m & lane_lt)] = (unsigned) (x + 3) | ((unsigned) (y + 3) << 8) | (sv << 16);   // (a pass that keeps nothing writes nothing)
                }
                total += __popcll(m);
            }
        } else {
            // ---- dense fallback (corner list overflow, never on natural images: > 512 corners in one cell): score and NMS every pixel ----
            const int npix = dw * dh;
            const unsigned mdw = kRcp16[dw];
            const int py = div_small(64, mdw), px = 64 - py * dw;
            {
                int y = div_small(lane, mdw), x = lane - y * dw;
                for (int base = 0; base < npix; base += 64) {
                    if (base + lane < npix) {
                        const uint8_t *cp = &win[(y + 3) * P + x + 4];
                        const int pol = fast9_test(cp, P, th);
                        if (pol) smap[(y + 1) * kSP + x + 1] = (uint8_t) fast9_arc_score(cp, P, pol);
                    }
                    y += py; x += px;
                    if (x >= dw) { x -= dw; y++; }
                }
            }
            wave_lds_sync();
            // single-pass plan: sub-pass 0 keeps the iniTh survivors, sub-pass 1 (only if that is empty) the minTh survivors
            for (int sub = kIniFirst ? 1 : 0; sub < 2; sub++) {
                total = 0;
                int y = div_small(lane, mdw), x = lane - y * dw;
                for (int base = 0; base < npix; base += 64) {
                    bool keep = false;
                    unsigned sv = 0;
                    if (base + lane < npix) {
                        const uint8_t *sp = &smap[(y + 1) * kSP + x + 1];
                        sv = sp[0];
                        if (sv > 0) keep = (nms_flags(sp, iniTh, kSP) & (sub == 0 ? 1 : 2)) != 0;
                    }
                    const unsigned long long m = __ballot(keep);
                    if (keep) out[total + __popcll(m & lane_lt)] = (unsigned) (x + 3) | ((unsigned) (y + 3) << 8) | (sv << 16);
                    total += __popcll(m);
                    y += py; x += px;
                    if (x >= dw) { x -= dw; y++; }
                }
                if (total > 0) break;
                usedMin = !kIniFirst;
            }
        }
        pc.mark(5);
        if (total > 0 || lastPass) {
            if (lane == 0) *cnt_out = (unsigned short) total;
            if (sampled && lane == 0) {
                atomicAdd(&st[0], 1u);
                st[3] = kIniFirst ? 2u : 1u;   // which plan these numbers come from
                if (usedMin || (nPass == 2 && pass == 1)) atomicAdd(&st[1], 1u);   // the cell's keypoints are FAST(minTh)'s
            }
            return;
        }
        wave_lds_sync();   // the next pass reuses the quad list / score map / corner list bytes
    }
}



template <int kP, bool kIniFirst>
__device__ __forceinline__ void fast_cell(const FrameSet &fs, const LevelGeom *__restrict__ geom, int nlevels, int iniTh, int minTh,
                                          unsigned short *__restrict__ cellCnt, unsigned *__restrict__ slots, int totalCells, long long totalSlots,
                                          int winPitch, int winRows, int smapRows, int quadCap, uint8_t *fdyn, int grp, int f, int wv, int lane,
                                          const FastGroupBases &gb, unsigned *__restrict__ stats) {
    // level of this group: the per-level first-group table rides in the kernel arguments (SGPRs after the initial argument load), so the
    // search is scalar compares instead of a chain of dependent scalar loads from geom[] (up to one memory round trip per level)
    int l = 0;
#pragma unroll
    for (int k = 1; k < kMaxLevels; k++)
        if (k < nlevels && grp >= gb.v[k]) l = k;
    const LevelGeom g = geom[l];
    const int gl = grp - g.groupBase;
    const int gCols = (g.nCols + 1) >> 1;
    const int gi = gl / gCols, gj = gl - gi * gCols;
    const int ci = 2 * gi + (wv >> 1), cj = 2 * gj + (wv & 1);
    if (ci >= g.nRows || cj >= g.nCols) return;
    const int c = ci * g.nCols + cj;
    unsigned short *cnt_out = cellCnt + (long long) f * totalCells + g.cellBase + c;
    const int iniX = kBorder + cj * g.wCell, iniY = kBorder + ci * g.hCell;
    const int maxX = min(iniX + g.wCell + 6, g.maxBorderX), maxY = min(iniY + g.hCell + 6, g.maxBorderY);
    const bool skip = (iniX >= g.maxBorderX - 6) || (iniY >= g.maxBorderY - 3);  // :751,:759 (asymmetric on purpose)
    const int dw = maxX - iniX - 6, dh = maxY - iniY - 6;
    if (skip || dw <= 0 || dh <= 0) {
        if (lane == 0) *cnt_out = 0;
        return;
    }
    const int P = kP ? kP : winPitch;
    const int winBytes = (winRows * P + 16 + 15) & ~15;   // 16 bytes of slack: the last quad of the last row reads past its row
    // the quad list lives only between pass 1 and the expansion, the score map only after the expansion: they share their bytes (one
    // more workgroup fits a CU: 8 x 4 waves instead of 6 x 4 at the usual 30-px cells)
    const int smapBytes = (max(smapRows * P, quadCap * 4) + 15) & ~15;
    const int perWave = winBytes + smapBytes + kCornerCap * 2;    // all multiples of 16
    uint8_t *win = fdyn + wv * perWave;
    uint8_t *smap = win + winBytes;
    unsigned short *clist = (unsigned short *) (smap + smapBytes);
    {   // stage the window: LDS column 1 + b of row r = image pixel (iniX + b, iniY + r), i.e. tested pixel x of the cell at column x + 4.
        // All global loads of a chunk are issued before the first use; lanes past the end repeat the last dword.
        int pitch;
        const uint8_t *img = level_ptr(fs, g, l, f, &pitch);
        const int ww = maxX - iniX, wh = maxY - iniY;
        const int nd = (ww + 1 + 3) >> 2, total = nd * wh;
        const unsigned g0 = (unsigned) (iniX - 1);
        const unsigned sh = g0 & 3u;
        const unsigned *src = (const unsigned *) (img + (unsigned) iniY * (unsigned) pitch + (g0 & ~3u));
        const unsigned pitch4 = (unsigned) pitch >> 2;
        const unsigned mnd = kRcp16[nd];
        int r = div_small(lane, mnd), d = lane - __mul24(r, nd);
        const int sr = div_small(64, mnd), sd = 64 - sr * nd;
        constexpr int kU = 6;
        for (int i0 = 0; i0 < total; i0 += 64 * kU) {
            unsigned lo[kU], hi[kU];
            int dst[kU];
#pragma unroll
            for (int u = 0; u < kU; u++) {
                const bool in = r < wh;
                const int rr = in ? r : wh - 1, dd = in ? d : nd - 1;
                const unsigned *sp = src + __umul24((unsigned) rr, pitch4) + (unsigned) dd;   // rows, pitch < 2^24
                lo[u] = sp[0];
                hi[u] = sp[1];
                dst[u] = __mul24(rr, P >> 2) + dd;
                r += sr; d += sd;
                if (d >= nd) { d -= nd; r++; }
            }
#pragma unroll
            for (int u = 0; u < kU; u++) ((unsigned *) win)[dst[u]] = __builtin_amdgcn_alignbyte(hi[u], lo[u], sh);
        }
    }
    wave_lds_sync();
    PhaseClk pc;   // (the phase clock reports k_fast_tab; here it is only the argument)
    pc.start();
    fast_cell_process<kP, kIniFirst>(win, smap, clist, P, dw, dh, iniTh, minTh, cnt_out,
                                     slots + (long long) f * totalSlots + g.slotBase + (long long) c * g.slotCap, grp, lane, stats, pc);
}


// One workgroup = kFastRep consecutive 2x2 groups of cells of one frame, one WAVE per cell position, no block-level synchronisation: a wave
// works through its kFastRep cells one after the other.  Measured (isolated, 256 frames): 1 cell per wave 625-645 us, 2 cells 644 us,
// 4 cells 742 us -- longer-lived waves lose more to imbalance inside a workgroup (its LDS is held until the slowest wave ends) and to
// the grid's tail than they gain from fewer workgroup launches, so one cell per wave stays.
constexpr int kFastRep = 1;
template <int kP, bool kIniFirst>
__global__ __launch_bounds__(kFastBlock) void k_fast_quads(FrameSet fs, const LevelGeom *__restrict__ geom, int nlevels,
                                                          int iniTh, int minTh, unsigned short *__restrict__ cellCnt,
                                                          unsigned *__restrict__ slots, int totalCells, long long totalSlots,
                                                          int totalGroups, int groupsPerXcd, int winPitch, int winRows,
                                                          int smapRows, int quadCap, FastGroupBases gb, unsigned *__restrict__ stats) {
    extern __shared__ __attribute__((aligned(16))) uint8_t fdyn[];
    const int tid = threadIdx.x, lane = tid & 63;
    // XCD-aware mapping (performance only): workgroup b runs on XCD b % 8; give every XCD a contiguous run of groups so
    // that neighbouring windows, which share cache lines, meet in the same L2.  groupsPerXcd counts workgroups (kFastRep groups each).
    if ((int) (blockIdx.x >> 3) >= groupsPerXcd) return;
    const int sg = (blockIdx.x & 7) * groupsPerXcd + (blockIdx.x >> 3);
    const int f = blockIdx.y;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);   // everything derived from the wave index stays on the scalar unit
#pragma unroll 1
    for (int r = 0; r < kFastRep; r++) {
        const int grp = sg * kFastRep + r;
        if (grp >= totalGroups) return;
        fast_cell<kP, kIniFirst>(fs, geom, nlevels, iniTh, minTh, cellCnt, slots, totalCells, totalSlots, winPitch, winRows, smapRows, quadCap, fdyn, grp, f, wv, lane, gb, stats);
        wave_lds_sync();   // the next cell reuses this wave's LDS region
    }
}

// ------------------------------------------------------------------------------------------------------------------
// K2t  The same cell loop with the per-cell bookkeeping taken off the instruction stream.  Counters of k_fast_quads showed that a wave
// spends about as many issue slots outside the corner test as inside it (per cell ~ 1050 vector + 410 scalar + 110 LDS instructions, of
// which pass 1 is 4 x 165 vector), and that the kernel's time is the sum of those slots over the SIMD's waves -- every instruction of any
// kind counts, a persistent variant that hid the window latency behind the previous cell (LDS-DMA double buffer, 6 waves per SIMD) was
// slower.  So:
//   * the level search, geometry load, group -> cell division and border arithmetic come precomputed: one FastCellRec per (group, wave)
//     position, built on the host with the geometry, fetched by ONE scalar 32-byte load;
//   * the window goes global memory -> LDS by LDS-DMA in 16-byte pieces (global_load_lds_dwordx4: lane i of a piece delivers LDS bytes
//     16 i .. 16 i + 15, so a 48-byte window pitch makes lane i = row i / 3, chunk i % 3 -- two pieces for a 40-row window instead of
//     twelve loads, six v_alignbyte and six ds_write per lane); gfx950's LDS-DMA takes byte-unaligned global addresses
//     (tools/micro/glds_probe.hip), so the window still starts at image column iniX - 1 and nothing downstream changes.
// Cells wider than 41 px (window > 48 bytes) keep k_fast_quads.
// ------------------------------------------------------------------------------------------------------------------
typedef __attribute__((address_space(3))) void lds_void_t;
constexpr int kTabPitch = 48;
#ifndef YGZF_FAST_TAB_WAVES
#define YGZF_FAST_TAB_WAVES 4
#endif
// cells (waves) per workgroup of k_fast_tab: a divisor of 4 (the cell table is built in groups of four).  Measured per 256 frames on one box,
// isolated / in the three-stream bench: 4 waves 432 us / 235.8 k frames/s, 2 waves 418 us / 231.2 k, 1 wave 455 us / 225.6 k -- smaller
// workgroups free their LDS sooner, but the pipeline as a whole runs best with four.
constexpr int kFastTabWaves = YGZF_FAST_TAB_WAVES;

// one cell of the table: record load, window by LDS-DMA, fast_cell_process.  `win` = this wave's LDS region.
template <bool kIniFirst>
__device__ __forceinline__ void fast_tab_cell(const FrameSet &fs, const FastCellRec *__restrict__ cells, int rec, int f, int iniTh, int minTh,
                                              unsigned short *__restrict__ cellCnt, unsigned *__restrict__ slots, int totalCells, long long totalSlots,
                                              uint8_t *win, int winBytes, int smapBytes, int lane, unsigned *__restrict__ stats, PhaseClk &pc) {
    const int grp = rec >> 2;
    const FastCellRec R = cells[rec];
    const unsigned fl = R.flags >> 8;
    if (!(fl & kFastCellExists)) return;
    unsigned short *cnt_out = cellCnt + (long long) f * totalCells + R.cell;
    if (!(fl & kFastCellRun)) {
        if (lane == 0) *cnt_out = 0;
        return;
    }
    const int wh = (int) (R.flags & 0xFFu), dw = (int) ((R.geo >> 16) & 0xFFu), dh = (int) (R.geo >> 24);
    const bool lvl0 = (R.flags >> 16) == 0;
    const unsigned pitch = lvl0 ? (unsigned) fs.img0_pitch : (R.geo & 0xFFFFu);
    const uint8_t *src = (lvl0 ? fs.img0 + (long long) f * fs.img0_stride : fs.pyr + (long long) f * fs.pyr_stride + R.off) +
                         (R.xy >> 16) * pitch + (R.xy & 0xFFFFu);
    constexpr int P = kTabPitch;
    uint8_t *smap = win + winBytes;
    unsigned short *clist = (unsigned short *) (smap + smapBytes);
    {   // window rows 0 .. wh - 1, 48 bytes each from column iniX - 1.  Reads at most 47 bytes past that column: inside the row's 16-px border
        // margin or, for windows clipped at the right border, inside the next row -- never past the image (the last window row lies at
        // least 16 rows above the bottom)
        const int total = wh * 3;
        for (int i0 = 0; i0 < total; i0 += 64) {
            const int i = i0 + lane;
            const int r = (int) (((unsigned) i * 21846u) >> 16), c16 = i - 3 * r;   // i / 3, exact for i < 32768
            if (i < total)
                __builtin_amdgcn_global_load_lds((const unsigned *) (src + __umul24((unsigned) r, pitch) + 16 * c16), (lds_void_t *) (win + 16 * i0), 16, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    wave_lds_sync();
    pc.mark(0);
    fast_cell_process<P, kIniFirst>(win, smap, clist, P, dw, dh, iniTh, minTh, cnt_out, slots + (long long) f * totalSlots + R.slot, grp, lane, stats, pc);
}

template <bool kIniFirst>
__global__ __launch_bounds__(64 * kFastTabWaves) void k_fast_tab(FrameSet fs, const FastCellRec *__restrict__ cells, int iniTh, int minTh,
                                                         unsigned short *__restrict__ cellCnt, unsigned *__restrict__ slots, int totalCells,
                                                         long long totalSlots, int totalGroups, int groupsPerXcd, int winRows, int smapRows,
                                                         int quadCap, unsigned *__restrict__ stats) {
    extern __shared__ __attribute__((aligned(16))) uint8_t fdyn[];
    const int tid = threadIdx.x, lane = tid & 63;
    PhaseClk pc;
    pc.start();
    if ((int) (blockIdx.x >> 3) >= groupsPerXcd) return;
    const int blk = (blockIdx.x & 7) * groupsPerXcd + (blockIdx.x >> 3);   // XCD-aware: every XCD gets a contiguous run of cells
    const int f = blockIdx.y;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rec = blk * kFastTabWaves + wv;                              // (groupsPerXcd / totalGroups count workgroups of kFastTabWaves cells here)
    if (rec >= totalGroups * kFastTabWaves) return;
    const int winBytes = (winRows * kTabPitch + 16 + 15) & ~15;
    const int smapBytes = (max(smapRows * kTabPitch, quadCap * 4) + 15) & ~15;
    const int perWave = winBytes + smapBytes + kCornerCap * 2;
    fast_tab_cell<kIniFirst>(fs, cells, rec, f, iniTh, minTh, cellCnt, slots, totalCells, totalSlots, fdyn + wv * perWave, winBytes, smapBytes, lane, stats, pc);
    pc.flush(0, lane, (unsigned) rec + (unsigned) f * (unsigned) (totalGroups * kFastTabWaves));
}

// ------------------------------------------------------------------------------------------------------------------
// K2p  The same cells, PERSISTENT waves.  Counters of k_fast_tab on a 256-frame launch (profiles/r06_spi_*.txt): 21 of a CU's 32 wave slots are
// occupied on average although LDS and registers allow all 32 -- the dispatcher places the four waves of a workgroup on the four SIMDs in order
// and stalls on the first full one (SPI_RA_WAVE_SIMD_FULL 71 % of the cycles, LDS_CU_FULL 61 %) while slots elsewhere stand empty, because the
// waves of 270 000 ten-microsecond workgroups end in no particular order.  Here as many workgroups as the chip holds are launched ONCE and every
// wave draws cells from a counter until none is left: no wave is ever launched again, every slot stays occupied to the end, and a wave that
// met an expensive cell simply draws fewer.  Items = (frame, cell record) in frame-major order, cut into eight contiguous chunks, one per XCD
// (workgroup b runs on XCD b % 8: a frame's windows meet in one L2), each with its own counter; a wave whose chunk is exhausted helps the next
// XCD's share.  The draw for the NEXT cell is issued before the current one's window is requested, so its round trip hides behind that window's.
// ------------------------------------------------------------------------------------------------------------------
template <bool kIniFirst>
__global__ __launch_bounds__(256) void k_fast_tab_persist(FrameSet fs, const FastCellRec *__restrict__ cells, int iniTh, int minTh,
                                                          unsigned short *__restrict__ cellCnt, unsigned *__restrict__ slots, int totalCells,
                                                          long long totalSlots, unsigned recsPerFrame, unsigned long long recMagic, unsigned totalItems,
                                                          unsigned itemsPerXcd, int winRows, int smapRows, int quadCap, unsigned *__restrict__ stats,
                                                          unsigned *__restrict__ counters) {
    extern __shared__ __attribute__((aligned(16))) uint8_t fdyn[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int winBytes = (winRows * kTabPitch + 16 + 15) & ~15;
    const int smapBytes = (max(smapRows * kTabPitch, quadCap * 4) + 15) & ~15;
    const int perWave = winBytes + smapBytes + kCornerCap * 2;
    uint8_t *win = fdyn + wv * perWave;
    PhaseClk pc;
    pc.start();
    // 64 counters, 256 bytes apart: XCD x (= blockIdx.x % 8) owns counters 8 x .. 8 x + 7, each over an eighth of the XCD's items.  The draws are
    // WORKGROUP-scope atomics: they execute in the XCD's own L2 (an agent-scope one is carried out at the memory side, ~14 ns apiece on one
    // line for the whole device: 270 000 draws of a launch took 3.8 ms that way).  That is enough: a counter is only ever drawn from by waves of
    // its own XCD -- and should the hardware ever place a workgroup elsewhere, two L2s would each count from zero and some cells would be
    // computed twice, to the same bytes (the work is idempotent), never skipped.
    const unsigned xcd = blockIdx.x & 7u;
    unsigned sub = (blockIdx.x >> 3) & 7u;
    int chunksTried = 0;
    const unsigned xLo = xcd * itemsPerXcd, xHi = min(xLo + itemsPerXcd, totalItems);
    const unsigned perSub = (itemsPerXcd + 7u) >> 3;
    auto draw = [&](unsigned sb) { return __hip_atomic_fetch_add(&counters[(xcd * 8u + sb) * 64u], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); };
    unsigned pend = 0;                                  // lane 0: the draw in flight
    if (lane == 0) pend = draw(sub);
#pragma unroll 1
    while (true) {
        const unsigned idx = (unsigned) __builtin_amdgcn_readfirstlane((int) pend);
        const unsigned lo = min(xLo + sub * perSub, xHi), hi = min(lo + perSub, xHi);
        if (idx >= hi - lo) {                           // this share is done: help the XCD's next one
            if (++chunksTried == 8) break;
            sub = (sub + 1u) & 7u;
            if (lane == 0) pend = draw(sub);
            continue;
        }
        const unsigned item = lo + idx;
        if (lane == 0) pend = draw(sub);                // the next draw: in flight while this cell's record and window are fetched
        unsigned f = (unsigned) (((unsigned long long) item * recMagic) >> 40);   // item / recsPerFrame (magic multiply, corrected below)
        if (f * recsPerFrame > item) f--;
        if ((f + 1u) * recsPerFrame <= item) f++;
        const unsigned rec = item - f * recsPerFrame;
        fast_tab_cell<kIniFirst>(fs, cells, (int) rec, (int) f, iniTh, minTh, cellCnt, slots, totalCells, totalSlots, win, winBytes, smapBytes, lane, stats, pc);
        pc.flush(0, lane, item);
        pc.start();
        wave_lds_sync();                                // the next cell reuses this wave's LDS region
    }
}

// ------------------------------------------------------------------------------------------------------------------
// K4  DistributeOctTree on the device.  One workgroup per (level, frame).
//
// The reference subdivides std::list nodes holding copies of their keypoints.  Here every candidate gets a PATH KEY:
// its root index followed by 2 bits per subdivision (bit0: x >= midX, bit1: y >= midY, with DivideNode's ceil-halving,
// :479-531).  After a stable radix sort by key every node of the tree at every depth is a contiguous range, so the
// breadth-first passes (:578-700) only touch (lo, cnt, depth) triples; child ranges come from binary searches.
// List order (push_front of n1..n4, erase of the parent), the "expand the biggest nodes first" phase with its
// (size, creation order) sort and the early break at N nodes, and the final per-node arg-max response (first maximum in
// original candidate order) are reproduced exactly; tools/octree_proto.py is the executable specification.
// ------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned path_key(int x, int y, const LevelGeom &g) {
    int root = (int) ((float) x / g.hX);
    root = min(root, g.nIni - 1);
    int xl = (int) (g.hX * (float) root), xr = (int) (g.hX * (float) (root + 1));
    int yl = 0, yr = g.regH;
    unsigned k = (unsigned) root;
    for (int d = 0; d < g.depth; d++) {
        const int mx = xl + ((xr - xl + 1) >> 1);
        const int my = yl + ((yr - yl + 1) >> 1);
        const unsigned bx = x >= mx, by = y >= my;
        if (bx) xl = mx; else xr = mx;
        if (by) yl = my; else yr = my;
        k = (k << 2) | (by << 1) | bx;
    }
    return k;
}

// the three child boundaries of a node at once: a_c = first index in [lo, lo+cnt) whose digit is >= c (c = 1, 2, 3).  The range is sorted by
// that digit, so each boundary is a lower bound over the WHOLE range -- the same values as three nested searches, but the three walks
// advance together: 11 dependent LDS round trips for a 2000-candidate node instead of 33
__device__ __forceinline__ void digit_bounds3(const unsigned *keys, int lo, int cnt, int shift, int *a1, int *a2, int *a3) {
    int a[3] = {lo, lo, lo}, b[3] = {lo + cnt, lo + cnt, lo + cnt};
    while (a[0] < b[0] || a[1] < b[1] || a[2] < b[2]) {
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const bool go = a[c] < b[c];
            const int m = go ? (a[c] + b[c]) >> 1 : lo;
            const bool less = ((keys[m] >> shift) & 3u) < (unsigned) (c + 1);
            a[c] = go && less ? m + 1 : a[c];
            b[c] = go && !less ? m : b[c];
        }
    }
    *a1 = a[0]; *a2 = a[1]; *a3 = a[2];
}

// Candidates that follow each other in cell-major order fall into the same histogram bin / the same final node in runs, and an LDS atomic of 64 lanes
// on one address is 64 atomics in a row (the selection pass of a 61 000-candidate level: 39 us of same-address conflicts).  Within a row of 16 lanes
// (DPP, no LDS traffic) the lanes of a run of equal ids are combined and only the LAST lane of each run goes to LDS.
// id >= 0; lanes without an item pass a negative id of their own.
__device__ __forceinline__ bool row_run_last(int id) {   // is this the last lane of its run of equal ids inside its row?
    const int nxt = __builtin_amdgcn_update_dpp(-1, id, 0x101, 0xF, 0xF, false);   // row_shl:1 -- the row's last lane keeps -1
    return nxt != id;
}
template <int kCtrl>
__device__ __forceinline__ unsigned row_run_max_step(int id, unsigned v) {
    const int oid = __builtin_amdgcn_update_dpp(-1, id, kCtrl, 0xF, 0xF, false);
    const unsigned ov = (unsigned) __builtin_amdgcn_update_dpp(0, (int) v, kCtrl, 0xF, 0xF, false);
    return oid == id ? max(v, ov) : v;
}
__device__ __forceinline__ unsigned row_run_max(int id, unsigned v) {   // max of v over the lanes of this lane's run up to this lane (max is idempotent:
    v = row_run_max_step<0x111>(id, v);                                 // equal ids are enough, the run need not be checked for gaps); row_shr:1, 2, 4, 8
    v = row_run_max_step<0x112>(id, v);
    v = row_run_max_step<0x114>(id, v);
    return row_run_max_step<0x118>(id, v);
}
__device__ __forceinline__ int row_run_length(int id, int lane) {   // lanes of this lane's run up to and including this lane
    const int prv = __builtin_amdgcn_update_dpp(-1, id, 0x111, 0xF, 0xF, false);   // row_shr:1 -- the row's first lane keeps -1
    const unsigned long long starts = __ballot(prv != id);                          // first lanes of runs (every row's lane 0 among them)
    const unsigned long long upTo = starts & (~0ull >> (63 - lane));
    return lane - (63 - __builtin_clzll(upTo)) + 1;
}

struct OctShared {  // carved out of dynamic LDS
    int *cellPref;            // nCells + 1
    int *nlo[2], *ncnt[2], *ndep[2];  // node list, double buffered (cap each)
    int *kArr, *eArr, *sArr;  // per-node scratch (cap each)
    int *b1, *b2, *b3;        // child boundaries (cap each)
    int *Epos, *Ecnt;         // expandable nodes in creation order (cap each)
    unsigned *sk[2], *sv[2];  // sort buffers for E (cap each)
    int *flag;                // processed flag per list position (cap)
    unsigned *cand;           // LDS-resident candidate sort buffers (4 x ldsCand), optional
};

// kGlobalNodes: the 19 per-list-position arrays live in a global arena (nodeArena, 19 * cap ints per (frame, level)) instead of LDS --
// configurations whose per-level feature budget is too large for the LDS plan (e.g. one level with > 2000 features).
constexpr unsigned kNoKeypoint = 0xFFFFFFFFu;   // procRec.y of a processing position without a keypoint
//
// kHist (the plan of configurations whose levels hold tens of thousands of candidates -- 1920x1080 / 4000, 3840x2160 / 8000): NO SORT.  What the
// tree passes ask of the sorted key array is (a) the child boundaries of a node = counts of candidates per key prefix and (b) the best
// response per final node.  Both come from a histogram over the key prefixes of depth dm (nIni << 2 dm <= histBins bins, LDS atomics while the
// keys are computed) and its exclusive prefix sum PS: the node (lo, cnt, dep) starts at the aligned bin f0 with PS[f0] == lo, its child
// boundaries are PS[f0 + c * 4^(dm - dep - 1)] -- the very values digit_bounds3 finds in the sorted array, so the tree passes run unchanged; the
// best response per node is one more pass over the candidates (bin -> final node table, LDS atomic max).  A level of 60 000 candidates spent
// 1.4-1.7 ms in four scattered radix passes through global memory (4.6 GB written per 64-frame launch against 0.9 GB of keys) and 0.2-0.5 ms
// in tree passes whose every boundary search was 11 dependent global reads; here the candidates are read twice, linearly, and the tree passes
// stay in LDS.  A tree that wants to split a node BELOW depth dm (a few very crowded spots in an otherwise empty level) raises s_overflow and
// the workgroup starts over on the sorting path -- same result, the old speed.
template <bool kGlobalNodes, bool kHist>
__global__ __launch_bounds__(kOctBlock) __attribute__((amdgpu_waves_per_eu(kHist ? 4 : 8, 8))) void k_octree(const LevelGeom *__restrict__ geom, int nlevels, int levelBase,
                                                      const unsigned short *__restrict__ cellCnt,
                                                      const unsigned *__restrict__ slots, int totalCells,
                                                      long long totalSlots, unsigned *__restrict__ candKey0,
                                                      unsigned *__restrict__ candVal0, unsigned *__restrict__ candKey1,
                                                      unsigned *__restrict__ candVal1, unsigned *__restrict__ candXY,
                                                      long long candStride, unsigned *__restrict__ lvlKpXY,
                                                      unsigned char *__restrict__ lvlKpScore, int *__restrict__ lvlKpCnt,
                                                      int *__restrict__ lvlCandCnt, uint2 *__restrict__ procRec,
                                                      int kpStride, int cap, int ldsCand, long long *dbg, int *__restrict__ nodeArena,
                                                      int regionInts, int histBins, int *gHist, int *gDone, int doneTarget, int spinBudget) {
    static_assert(!(kGlobalNodes && kHist), "the histogram plan keeps the node arrays in LDS");
    extern __shared__ __attribute__((aligned(16))) int dyn[];
    __shared__ int histT[kRadixHist];
    __shared__ int s_tmp[20];
    __shared__ unsigned long long s_tmp64[17];
    __shared__ int s_n, s_nE, s_cut, s_flagA, s_overflow;
    __shared__ int s_head[4];
    const int tid = threadIdx.x;
    const int l = blockIdx.x + levelBase, f = blockIdx.y;
    // kHist, launches of a few frames (gridDim.z > 1): a level of one 3840x2160 frame is 60 000 candidates whose keys took ONE compute unit 95 us while
    // 250 others had nothing to do.  The workgroups z = 1 .. gridDim.z - 1 of a (level, frame) are helpers: each computes the keys of its share of the
    // candidates (keys and positions in the global candidate arrays as always, its bin counts into a slice of gHist), releases, bumps gDone and
    // leaves; workgroup 0 does its own share, waits for the others, adds their counts to its own and goes on alone.  gDone only ever grows (the host
    // passes the value this launch brings it to): should the helpers not arrive within the spin budget -- every compute unit held by waiting
    // workgroups 0 of many contexts at once -- workgroup 0 stops waiting, computes the whole level itself and ignores the slices; helpers that come
    // late write the same keys to the same places and bump a counter nobody reads again before the next launch.  The host launches helpers only while ALL workgroups of the launch fit the device together (ygzf_api.hip), so nobody waits for
    // a workgroup that cannot start.
    const int parts = kHist && gHist ? (int) gridDim.z : 1, part = kHist ? (int) blockIdx.z : 0;
#define OSTAMP(k) do { if (dbg && tid == 0 && f == 0 && part == 0) dbg[l * 8 + (k)] = wall_clock64(); } while (0)
    OSTAMP(0);
    int bfsEv = 0;
// (the tree passes' own stamps exist in the -DYGZF_PHASE_CLOCK build only: eight conditional stamps inside the pass loops cost the product build 10 % more vector
// instructions in this kernel -- 23 k per frame -- although the branch is never taken)
#ifndef YGZF_PHASE_CLOCK
#define OBFS(nn) do { } while (0)
#else
#define OBFS(nn) do { if (dbg && tid == 0 && f == 0 && part == 0 && bfsEv < 8) { dbg[16 * 8 + 16 + l * 16 + 2 * bfsEv] = (nn); dbg[16 * 8 + 16 + l * 16 + 2 * bfsEv + 1] = wall_clock64(); bfsEv++; } } while (0)
#endif
    const LevelGeom g = geom[l];
    const int nCells = g.nCols * g.nRows;
    int *lvlCnt = lvlKpCnt + f * nlevels + l;
    int *gd = kHist && parts > 1 ? gDone + (f * nlevels + l) : nullptr;
    auto helper_leaves = [&]() { if (tid == 0) __hip_atomic_fetch_add(gd, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT); };   // (every helper, whatever way it leaves)
    if (nCells <= 0 || g.nCols <= 0) {
        if (part != 0) { helper_leaves(); return; }
        if (tid == 0) { *lvlCnt = 0; lvlCandCnt[f * nlevels + l] = 0; }
        for (int i = tid; i < g.kpCap; i += kOctBlock) procRec[(long long) f * kpStride + g.kpBase + i] = make_uint2(0u, kNoKeypoint);
        return;
    }
    OctShared S;
    int *PS = nullptr;        // kHist: histBins + 1 ints behind the region the cell table, the key tables and (later) the node arrays share
    {
        int *p = dyn;
        S.cellPref = p;
        if (kHist) PS = dyn + regionInts;        // (the node arrays start at dyn as well: the cell prefix table is dead once the keys exist)
        else p += (nCells + 1 + 3) & ~3;         // (16-byte aligned node arrays: cap is a multiple of 4)
        int *candLds = p;
        if (kGlobalNodes) p = nodeArena + ((long long) blockIdx.y * nlevels + l) * (19LL * cap);
        for (int b = 0; b < 2; b++) { S.nlo[b] = p; p += cap; S.ncnt[b] = p; p += cap; S.ndep[b] = p; p += cap; }
        S.kArr = p; p += cap; S.eArr = p; p += cap; S.sArr = p; p += cap;
        S.b1 = p; p += cap; S.b2 = p; p += cap; S.b3 = p; p += cap;
        S.Epos = p; p += cap; S.Ecnt = p; p += cap;
        for (int b = 0; b < 2; b++) { S.sk[b] = (unsigned *) p; p += cap; S.sv[b] = (unsigned *) p; p += cap; }
        S.flag = p; p += cap;
        S.cand = (unsigned *) (kGlobalNodes ? candLds : p);   // 4 * ldsCand words: key/val double buffers when the level's candidates fit
    }
    const unsigned short *cc = cellCnt + (long long) f * totalCells + g.cellBase;
    const unsigned *sl = slots + (long long) f * totalSlots + g.slotBase;
    unsigned *key0 = candKey0 + (long long) f * candStride + g.candBase;   // global scratch (used when M > ldsCand)
    unsigned *val0 = candVal0 + (long long) f * candStride + g.candBase;
    unsigned *key1 = candKey1 + (long long) f * candStride + g.candBase;
    unsigned *val1 = candVal1 + (long long) f * candStride + g.candBase;
    unsigned *xy = candXY + (long long) f * candStride + g.candBase;

    // kHist: depth dm of the histogram's key prefixes -- as deep as the bin budget allows (a level whose tree is fully resolved at dm cannot overflow)
    int dm = 0;
    if (kHist) {
        dm = g.depth;
        while (dm > 0 && ((long long) g.nIni << (2 * dm)) > (long long) histBins) dm--;
    }
    bool useHist = kHist && dm >= 1;
    const int nBins = g.nIni << (2 * dm);
    const int binShift = 2 * (g.depth - dm);
    if (part != 0 && !useHist) { helper_leaves(); return; }   // (no histogram for this level: workgroup 0 sorts, alone)
    bool alone = false;                  // workgroup 0 gave up waiting for its helpers
    int *gh = kHist && parts > 1 ? gHist + ((long long) f * nlevels + l) * (parts - 1) * histBins : nullptr;   // one slice of histBins counts per helper
restart:   // (kHist: a second time, on the sorting path, after the tree asked for a split below depth dm)
    if (kHist && tid == 0) s_overflow = 0;
    // ---- 1. candidate offsets per cell (cell-major order == the reference's vToDistributeKeys order) ----
    for (int i = tid; i < nCells; i += kOctBlock) S.cellPref[i] = cc[i];
    __syncthreads();
    const int M = block_scan_array(S.cellPref, nCells, s_tmp);
    if (tid == 0) { S.cellPref[nCells] = M; if (part == 0) lvlCandCnt[f * nlevels + l] = M; }
    __syncthreads();
    if (M == 0) {
        if (part != 0) { helper_leaves(); return; }
        if (tid == 0) *lvlCnt = 0;
        for (int i = tid; i < g.kpCap; i += kOctBlock) procRec[(long long) f * kpStride + g.kpBase + i] = make_uint2(0u, kNoKeypoint);
        return;
    }
    OSTAMP(1);
    // ---- 2. path keys: one thread per candidate (its cell by binary search in the prefix table) ----
    const bool inLds = M <= ldsCand;
    if (inLds) { key0 = S.cand; val0 = S.cand + ldsCand; key1 = S.cand + 2 * ldsCand; val1 = S.cand + 3 * ldsCand; }
    // A workgroup has ONE CU: 2800 candidates x ~300 instructions of cell decode (a division), root (a float division) and ten
    // subdivision steps were 7 us of a 752x480 level 0.  x and y subdivide independently, so the key is the OR of a per-column word
    // (root, x bits spread to the even positions) and a per-row word (y bits on the odd positions): both tables and the cells' origins
    // are built once per workgroup in the node arrays (idle until the tree passes) when they fit there.
    const int lenX = max(g.regW, g.nCols * g.wCell) + 9, lenY = max(g.regH, g.nRows * g.hCell) + 9;
    const bool useTab = (kHist ? lenX + lenY + nCells <= regionInts - (nCells + 1) : lenX + lenY + nCells <= 19 * cap) && g.depth <= 15;
    unsigned *xTab = (unsigned *) (kHist ? dyn + nCells + 1 : S.nlo[0]), *yTab = xTab + lenX, *orgTab = yTab + lenY;

    if (useTab) {
        auto spread = [](unsigned v) {
            v = (v | (v << 8)) & 0x00FF00FFu;
            v = (v | (v << 4)) & 0x0F0F0F0Fu;
            v = (v | (v << 2)) & 0x33333333u;
            return (v | (v << 1)) & 0x55555555u;
        };
        for (int t = tid; t < lenX + lenY + nCells; t += kOctBlock) {
            if (t < lenX) {
                const int x = t;
                int root = (int) ((float) x / g.hX);
                root = min(root, g.nIni - 1);
                int xl = (int) (g.hX * (float) root), xr = (int) (g.hX * (float) (root + 1));
                unsigned kx = 0;
                for (int d = 0; d < g.depth; d++) {
                    const int mx = xl + ((xr - xl + 1) >> 1);
                    const unsigned bx = x >= mx;
                    if (bx) xl = mx; else xr = mx;
                    kx = (kx << 1) | bx;
                }
                xTab[t] = ((unsigned) root << (2 * g.depth)) | spread(kx);
            } else if (t < lenX + lenY) {
                const int y = t - lenX;
                int yl = 0, yr = g.regH;
                unsigned ky = 0;
                for (int d = 0; d < g.depth; d++) {
                    const int my = yl + ((yr - yl + 1) >> 1);
                    const unsigned by = y >= my;
                    if (by) yl = my; else yr = my;
                    ky = (ky << 1) | by;
                }
                yTab[y] = spread(ky) << 1;
            } else {
                const int c = t - lenX - lenY;
                const int ci = c / g.nCols, cj = c - ci * g.nCols;
                orgTab[c] = (unsigned) (cj * g.wCell) | ((unsigned) (ci * g.hCell) << 16);
            }
        }
        __syncthreads();
    }
    if (kHist && useHist) {
        for (int b = tid; b <= nBins; b += kOctBlock) PS[b] = 0;
        __syncthreads();
    }
    {
        // four candidates per thread at a time: their binary searches advance together (the LDS reads of a step travel together) and their
        // slot reads are all in flight before the first key is assembled
        int steps = 0;
        while ((1 << steps) < nCells) steps++;
        constexpr int kKU = 4;
        const int shares = kHist && useHist && !alone ? parts : 1;   // (the sorting path after a restart, or no helpers in sight: workgroup 0 takes everything)
        for (int i0 = tid + part * kKU * kOctBlock; i0 < M; i0 += shares * kKU * kOctBlock) {
            int ia[kKU], a[kKU], b[kKU];
#pragma unroll
            for (int u = 0; u < kKU; u++) { ia[u] = min(i0 + u * kOctBlock, M - 1); a[u] = 0; b[u] = nCells; }
            for (int st = 0; st < steps; st++) {             // a[u] = last cell c with cellPref[c] <= ia[u]
#pragma unroll
                for (int u = 0; u < kKU; u++) {
                    const int m = (a[u] + b[u]) >> 1;
                    const bool go = b[u] - a[u] > 1, le = S.cellPref[m] <= ia[u];
                    a[u] = go && le ? m : a[u];
                    b[u] = go && !le ? m : b[u];
                }
            }
            unsigned e[kKU];
            int ox[kKU], oy[kKU];
#pragma unroll
            for (int u = 0; u < kKU; u++) {
                const int c = a[u], k = ia[u] - S.cellPref[c];
                e[u] = sl[(long long) c * g.slotCap + k];
                if (useTab) {
                    const unsigned o = orgTab[c];
                    ox[u] = (int) (o & 0xFFFFu);
                    oy[u] = (int) (o >> 16);
                } else {
                    const int ci = c / g.nCols, cj = c - ci * g.nCols;
                    ox[u] = cj * g.wCell;
                    oy[u] = ci * g.hCell;
                }
            }
            int binOf[kKU];
#pragma unroll
            for (int u = 0; u < kKU; u++) {
                const int i = i0 + u * kOctBlock;
                binOf[u] = -2 - (tid & 63);
                if (i < M) {
                    const int x = (int) (e[u] & 255u) + ox[u], y = (int) ((e[u] >> 8) & 255u) + oy[u];
                    const unsigned key = useTab ? (xTab[x] | yTab[y]) : path_key(x, y, g);
                    if (kHist && useHist) {
                        const unsigned bin = key >> binShift;
                        binOf[u] = (int) bin;
                        key0[i] = bin | ((e[u] >> 16) << 16);      // bin < 65536; the score rides along for the selection pass
                    } else {
                        key0[i] = key;
                        val0[i] = ((e[u] >> 16) << 24) | (0xFFFFFFu - (unsigned) i);  // max() picks best score, then smallest index
                    }
                    xy[i] = (unsigned) x | ((unsigned) y << 16);
                }
            }
            if (kHist && useHist) {   // (uniform per workgroup; every lane of the wave takes part in the row operations)
#pragma unroll
                for (int u = 0; u < kKU; u++) {
                    const int len = row_run_length(binOf[u], tid & 63);
                    if (binOf[u] >= 0 && row_run_last(binOf[u])) atomicAdd(&PS[binOf[u]], len);
                }
            }
        }
    }
    if (kHist && useHist && parts > 1 && !alone) {
        // (counts in a slice of its own per helper, plain stores: 60 000 device-scope atomics on one histogram ran at 9 per nanosecond for the whole
        // device -- every level of a 3840x2160 pair waited 95 us for them)
        __syncthreads();
        if (part != 0) {
            int *mine = gh + (long long) (part - 1) * histBins;
            for (int b = tid; b < nBins; b += kOctBlock) mine[b] = PS[b];
        }
        // Release / acquire by ONE thread per workgroup, the barriers carrying the other threads' accesses along: an agent-scope release is a write-back
        // of the XCD's whole L2 -- executed by every wave of every workgroup (`__threadfence()` in all threads, the portable idiom) it made every level
        // of a 3840x2160 pair wait 95 us for 6000 of them.
        __syncthreads();                       // every store of this workgroup has reached the L2 (the barrier waits for vmcnt(0)) ...
        if (part != 0) {
            helper_leaves();                   // ... and leaves it with this release
            return;
        }
        if (tid == 0) {
            int spins = 0;
            bool here;
            while (!(here = __hip_atomic_load(gd, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) >= doneTarget) && spins < spinBudget) { __builtin_amdgcn_s_sleep(8); spins++; }
            s_flagA = here ? 1 : 0;
        }
        __syncthreads();
        if (!s_flagA) {                        // (uniform) nobody came: the whole level again, alone
            alone = true;
            __syncthreads();
            goto restart;
        }
        for (int b = tid; b < nBins; b += kOctBlock) {
            int v = PS[b];
            for (int q = 0; q < parts - 1; q++) v += gh[(long long) q * histBins + b];
            PS[b] = v;
        }
    }
    __syncthreads();
    OSTAMP(2);
    // ---- 3. sort by path key ----
    unsigned *skeys = key0, *svals = val0;
    if (kHist && useHist) {
        block_scan_array(PS, nBins, s_tmp);        // exclusive: PS[b] = candidates in front of bin b in path-key order
        if (tid == 0) PS[nBins] = M;
        __syncthreads();
    } else
        block_radix_sort(key0, val0, key1, val1, M, g.keyBits, histT, s_tmp, &skeys, &svals);
    OSTAMP(3);
    // child boundaries of the node (lo, cnt, dep): a_c = first position in [lo, lo + cnt] whose child digit is >= c
    auto bounds = [&](int lo, int cnt, int dep, int *a1, int *a2, int *a3) {
        if (kHist && useHist) {
            if (dep + 1 > dm) {                    // finer than the histogram: start over on the sorting path (the dummy answer keeps every loop finite)
                s_overflow = 1;
                *a1 = *a2 = *a3 = lo + cnt;
                return;
            }
            const int sh = 2 * (dm - dep);         // a node of depth dep spans 2^sh bins, aligned
            int a = 0, b = nBins >> sh;            // its index among the nodes of its depth: the largest j with PS[j << sh] <= lo (the node is not empty)
            while (b - a > 1) {
                const int m = (a + b) >> 1;
                if (PS[m << sh] <= lo) a = m; else b = m;
            }
            const int f0 = a << sh, s1 = 1 << (sh - 2);
            *a1 = PS[f0 + s1]; *a2 = PS[f0 + 2 * s1]; *a3 = PS[f0 + 3 * s1];
        } else
            digit_bounds3(skeys, lo, cnt, 2 * (g.depth - (dep + 1)), a1, a2, a3);
    };
    // ---- 4. breadth-first subdivision on ranges ----
    const int D = g.depth;
    const int N = g.nFeat;
    // children of node (lo, cnt, dep) with boundaries a1..a3 into list `nxt`; exK / exE / exS = the exclusive sums over the nodes before it
    auto emit = [&](int nxt, int lo, int cnt, int dep, int a1, int a2, int a3, int exK, int exE, int exS, int sumK) {
        if (cnt == 1) {
            const int p = sumK + exS;
            (nxt ? S.nlo[1] : S.nlo[0])[p] = lo; (nxt ? S.ncnt[1] : S.ncnt[0])[p] = 1; (nxt ? S.ndep[1] : S.ndep[0])[p] = dep;
        } else {
            const int bb[5] = {lo, a1, a2, a3, lo + cnt};
            int k = 0;
            for (int c = 0; c < 4; c++) k += (bb[c + 1] - bb[c]) > 0;
            int p = sumK - (exK + k);  // children of later parents sit in front (push_front)
            int epos[4];
            for (int c = 3; c >= 0; c--) {   // list order n4,n3,n2,n1
                const int cc2 = bb[c + 1] - bb[c];
                epos[c] = p;
                if (cc2 > 0) { (nxt ? S.nlo[1] : S.nlo[0])[p] = bb[c]; (nxt ? S.ncnt[1] : S.ncnt[0])[p] = cc2; (nxt ? S.ndep[1] : S.ndep[0])[p] = dep + 1; p++; }
            }
            int es = exE;
            for (int c = 0; c < 4; c++) {    // creation order n1..n4
                const int cc2 = bb[c + 1] - bb[c];
                if (cc2 > 1) { S.Epos[es] = epos[c]; S.Ecnt[es] = cc2; es++; }
            }
        }
    };
    // -- the head of the subdivision on ONE wave, the block waiting at one barrier: the roots (:566-575, a lane each) and every full pass
    //    (:588-640) while the list has at most 64 nodes (2, 8, 32 for a 752x480 level) -- bounds, the three prefix sums as one wave scan, the
    //    children, all on the wave's lanes, with nothing but the wave's own LDS order between the passes
    if (wave_id() == 0) {
        const int lane = lane_id();
        int n = 0, cur = 0, nE = 0, state = 0;   // state: 0 = go on block-wide, 1 = finished, 2 = continue with the expand phase
        if (g.nIni <= 64) {
            int hi = M;                           // first index of a root > lane
            if (lane < g.nIni) {
                if (kHist && useHist) hi = PS[(lane + 1) << (2 * dm)];
                else {
                    int a = 0, b = M;
                    while (a < b) {
                        const int m = (a + b) >> 1;
                        if ((int) (skeys[m] >> (2 * D)) <= lane) a = m + 1; else b = m;
                    }
                    hi = a;
                }
            }
            int lo = __shfl_up(hi, 1);
            if (lane == 0) lo = 0;
            const bool some = lane < g.nIni && hi > lo;   // empty roots are erased
            const unsigned long long bal = __ballot(some);
            if (some) {
                const int p = __popcll(bal & ((1ull << lane) - 1ull));
                S.nlo[0][p] = lo; S.ncnt[0][p] = hi - lo; S.ndep[0][p] = 0;
            }
            n = __popcll(bal);
        } else {
            if (lane == 0) {
                int lo = 0;
                for (int r = 0; r < g.nIni; r++) {
                    int a = lo, b = M;
                    if (kHist && useHist) a = b = PS[(r + 1) << (2 * dm)];
                    while (a < b) {
                        const int m = (a + b) >> 1;
                        if ((int) (skeys[m] >> (2 * D)) <= r) a = m + 1; else b = m;
                    }
                    if (a > lo) { S.nlo[0][n] = lo; S.ncnt[0][n] = a - lo; S.ndep[0][n] = 0; n++; }
                    lo = a;
                }
            }
            n = __builtin_amdgcn_readfirstlane(n);
        }
        wave_lds_sync();
        while (n <= 64) {
            const int prevSize = n, nxt = cur ^ 1;
            const bool act = lane < n;
            int lo = 0, cnt = 0, dep = 0, a1 = 0, a2 = 0, a3 = 0, k = 0, e = 0;
            if (act) {
                cnt = (cur ? S.ncnt[1] : S.ncnt[0])[lane];
                lo = (cur ? S.nlo[1] : S.nlo[0])[lane];
                dep = (cur ? S.ndep[1] : S.ndep[0])[lane];
                if (cnt > 1) {
                    bounds(lo, cnt, dep, &a1, &a2, &a3);
                    const int c0 = a1 - lo, c1 = a2 - a1, c2 = a3 - a2, c3 = lo + cnt - a3;
                    k = (c0 > 0) + (c1 > 0) + (c2 > 0) + (c3 > 0);
                    e = (c0 > 1) + (c1 > 1) + (c2 > 1) + (c3 > 1);
                }
            }
            const unsigned long long mine = (unsigned long long) (unsigned) k | ((unsigned long long) (unsigned) e << 21) |
                                            ((unsigned long long) (act && cnt == 1 ? 1u : 0u) << 42);
            const unsigned long long incl = wave_incl_scan_u64(mine), ex = incl - mine;
            const unsigned long long tot = ((unsigned long long) (unsigned) __builtin_amdgcn_readlane((int) (incl >> 32), 63) << 32) |
                                           (unsigned) __builtin_amdgcn_readlane((int) incl, 63);
            const int sumK = (int) (tot & 0x1FFFFFu);
            if (act) emit(nxt, lo, cnt, dep, a1, a2, a3, (int) (ex & 0x1FFFFFu), (int) ((ex >> 21) & 0x1FFFFFu), (int) (ex >> 42), sumK);
            wave_lds_sync();
            cur = nxt;
            n = sumK + (int) (tot >> 42);
            nE = (int) ((tot >> 21) & 0x1FFFFFu);
            if (n >= N || n == prevSize) { state = 1; break; }
            if (n + 3 * nE > N) { state = 2; break; }
        }
        if (lane == 0) { s_head[0] = n; s_head[1] = cur; s_head[2] = nE; s_head[3] = state; }
    }
    __syncthreads();
    int n = s_head[0], cur = s_head[1];
    int nE = s_head[2];
    OBFS(n);
    bool finish = s_head[3] == 1, toExpand = s_head[3] == 2;
    while (!finish) {
        if (!toExpand) {
            const int prevSize = n;
            // -- full pass (:588-640): every node with more than one point is divided
            int totK, totE, totS;
            const int nxt = cur ^ 1;
            for (int i = tid; i < n; i += kOctBlock) {
                const int cnt = (cur ? S.ncnt[1] : S.ncnt[0])[i];
                int k = 0, e = 0;
                if (cnt > 1) {
                    const int lo = (cur ? S.nlo[1] : S.nlo[0])[i];
                    int a1, a2, a3;
                    bounds(lo, cnt, (cur ? S.ndep[1] : S.ndep[0])[i], &a1, &a2, &a3);
                    S.b1[i] = a1; S.b2[i] = a2; S.b3[i] = a3;
                    const int c0 = a1 - lo, c1 = a2 - a1, c2 = a3 - a2, c3 = lo + cnt - a3;
                    k = (c0 > 0) + (c1 > 0) + (c2 > 0) + (c3 > 0);
                    e = (c0 > 1) + (c1 > 1) + (c2 > 1) + (c3 > 1);
                }
                S.kArr[i] = k; S.eArr[i] = e; S.sArr[i] = (cnt == 1);
            }
            __syncthreads();
            block_scan_array3(S.kArr, S.eArr, S.sArr, n, s_tmp64, &totK, &totE, &totS);
            for (int i = tid; i < n; i += kOctBlock) {
                const int cnt = (cur ? S.ncnt[1] : S.ncnt[0])[i], lo = (cur ? S.nlo[1] : S.nlo[0])[i], dep = (cur ? S.ndep[1] : S.ndep[0])[i];
                emit(nxt, lo, cnt, dep, cnt > 1 ? S.b1[i] : 0, cnt > 1 ? S.b2[i] : 0, cnt > 1 ? S.b3[i] : 0, S.kArr[i], S.eArr[i], S.sArr[i], totK);
            }
            __syncthreads();
            cur = nxt;
            n = totK + totS;
            nE = totE;
            OBFS(n);
            if (n >= N || n == prevSize) { finish = true; break; }
            if (n + 3 * nE <= N) continue;
        }
        toExpand = false;
        {
            // -- "expand the biggest first" phase (:647-700)
            while (!finish) {
                const int prev2 = n;
                if (nE == 0) { finish = true; break; }  // nothing left to expand: list size cannot change (:696)
                int seqBits = 1;
                while ((1 << seqBits) < max(nE, 2)) seqBits++;
                int cntBits = 1;
                while ((1 << cntBits) <= M) cntBits++;
                for (int j = tid; j < nE; j += kOctBlock) {
                    S.sk[0][j] = ((unsigned) S.Ecnt[j] << seqBits) | (unsigned) j;
                    S.sv[0][j] = (unsigned) j;
                }
                for (int j = tid; j < n; j += kOctBlock) S.flag[j] = 0;
                if (tid == 0) s_cut = nE - 1;
                __syncthreads();
                unsigned *ek, *ev;
                if (nE <= kOctBlock) {
                    // a few hundred distinct keys (they carry the creation sequence): the position of a key is the number of smaller keys -- counted off
                    // LDS reads, no histogram, no scan, ONE barrier (the radix sort: three passes of four).  T = 1 .. 16 adjacent lanes share a key, each
                    // counting a quarter-aligned slice of the list with 16-byte reads: one thread per key reading word by word made a round of 512
                    // expandable nodes 8192 wave-wide LDS reads -- 14 of the 22 us of a 1920x1080 level's only expand round.
                    // (launches of many frames have other workgroups to fill the wait: there one thread per key issues the fewest instructions)
                    int tl = 0;
                    while (gridDim.y <= 16 && tl < 4 && (nE << (tl + 1)) <= kOctBlock) tl++;
                    const int T = 1 << tl, j = tid >> tl, sub = tid & (T - 1);
                    const bool vec = (((unsigned) (uintptr_t) S.sk[0]) & 15u) == 0;   // (LDS offset; the host rounds the list capacity to a multiple of 4)
                    unsigned key = 0;
                    int rank = 0;
                    if (j < nE) {
                        key = S.sk[0][j];
                        const int per = (((nE + T - 1) >> tl) + 3) & ~3;
                        int q = sub * per;
                        const int q1 = min(nE, q + per);
                        // rank += (k < key) as compare + add-with-carry: two instructions per key (the compiler's select-and-add form is three)
                        auto count = [&](unsigned k) { asm("v_cmp_lt_u32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, 0, %0, vcc" : "+v"(rank) : "v"(k), "v"(key) : "vcc"); };
                        if (vec)
                            for (; q + 4 <= q1; q += 4) {
                                const uint4 k4 = *(const uint4 *) &S.sk[0][q];
                                count(k4.x); count(k4.y); count(k4.z); count(k4.w);
                            }
                        for (; q < q1; q++) count(S.sk[0][q]);
                    }
                    for (int d = 1; d < T; d <<= 1) rank += __shfl_xor(rank, d);
                    if (j < nE && sub == 0) {
                        S.sk[1][rank] = key;
                        S.sv[1][rank] = S.sv[0][j];
                    }
                    __syncthreads();
                    ek = S.sk[1];
                    ev = S.sv[1];
                } else
                    block_radix_sort(S.sk[0], S.sv[0], S.sk[1], S.sv[1], nE, seqBits + cntBits, histT, s_tmp, &ek, &ev);
                // processing order j: descending (size, creation seq)
                for (int j = tid; j < nE; j += kOctBlock) {
                    const int e = (int) ev[nE - 1 - j];
                    const int pos = S.Epos[e];
                    const int cnt = (cur ? S.ncnt[1] : S.ncnt[0])[pos], lo = (cur ? S.nlo[1] : S.nlo[0])[pos];
                    int a1, a2, a3;
                    bounds(lo, cnt, (cur ? S.ndep[1] : S.ndep[0])[pos], &a1, &a2, &a3);
                    S.b1[j] = a1; S.b2[j] = a2; S.b3[j] = a3;
                    const int c0 = a1 - lo, c1 = a2 - a1, c2 = a3 - a2, c3 = lo + cnt - a3;
                    S.kArr[j] = (c0 > 0) + (c1 > 0) + (c2 > 0) + (c3 > 0) - 1;  // growth of the list
                    S.eArr[j] = (c0 > 1) + (c1 > 1) + (c2 > 1) + (c3 > 1);
                }
                __syncthreads();
                block_scan_array(S.kArr, nE, s_tmp);  // exclusive growth before j
                for (int j = tid; j < nE; j += kOctBlock) {
                    const int e = (int) ev[nE - 1 - j];
                    const int pos = S.Epos[e];
                    const int cnt = (cur ? S.ncnt[1] : S.ncnt[0])[pos], lo = (cur ? S.nlo[1] : S.nlo[0])[pos];
                    const int bb[5] = {lo, S.b1[j], S.b2[j], S.b3[j], lo + cnt};
                    int k = 0;
                    for (int c = 0; c < 4; c++) k += (bb[c + 1] - bb[c]) > 0;
                    const int excl = S.kArr[j], incl = excl + k - 1;
                    if (n + incl >= N && n + excl < N) s_cut = j;  // first expansion that reaches N nodes: break
                }
                __syncthreads();
                const int nProc = s_cut + 1;
                // nodes j >= nProc are not expanded: no children, no new expandable entries
                for (int j = tid; j < nE; j += kOctBlock) {
                    if (j >= nProc) S.eArr[j] = 0;
                    else S.flag[S.Epos[(int) ev[nE - 1 - j]]] = 1;
                }
                __syncthreads();
                const int totE2 = block_scan_array(S.eArr, nE, s_tmp);
                // children of the processed nodes: total = growth + nProc
                int totC;
                {
                    // growth before nProc-1 plus its own
                    const int jl = nProc - 1;
                    const int e = (int) ev[nE - 1 - jl];
                    const int pos = S.Epos[e];
                    const int cnt = (cur ? S.ncnt[1] : S.ncnt[0])[pos], lo = (cur ? S.nlo[1] : S.nlo[0])[pos];
                    const int bb[5] = {lo, S.b1[jl], S.b2[jl], S.b3[jl], lo + cnt};
                    int k = 0;
                    for (int c = 0; c < 4; c++) k += (bb[c + 1] - bb[c]) > 0;
                    totC = S.kArr[jl] + jl + k;
                }
                // unprocessed old nodes keep their order behind the new children
                for (int i = tid; i < n; i += kOctBlock) S.sArr[i] = 1 - S.flag[i];
                __syncthreads();
                block_scan_array(S.sArr, n, s_tmp);
                const int nxt2 = cur ^ 1;
                for (int i = tid; i < n; i += kOctBlock) {
                    if (!S.flag[i]) {
                        const int p = totC + S.sArr[i];
                        (nxt2 ? S.nlo[1] : S.nlo[0])[p] = (cur ? S.nlo[1] : S.nlo[0])[i]; (nxt2 ? S.ncnt[1] : S.ncnt[0])[p] = (cur ? S.ncnt[1] : S.ncnt[0])[i]; (nxt2 ? S.ndep[1] : S.ndep[0])[p] = (cur ? S.ndep[1] : S.ndep[0])[i];
                    }
                }
                // new expandable list goes to the sort buffers first (Epos/Ecnt are still being read)
                unsigned *nEpos = (ek == S.sk[0]) ? S.sk[1] : S.sk[0];
                unsigned *nEcnt = (ev == S.sv[0]) ? S.sv[1] : S.sv[0];
                for (int j = tid; j < nProc; j += kOctBlock) {
                    const int e = (int) ev[nE - 1 - j];
                    const int pos = S.Epos[e];
                    const int cnt = (cur ? S.ncnt[1] : S.ncnt[0])[pos], lo = (cur ? S.nlo[1] : S.nlo[0])[pos], dep = (cur ? S.ndep[1] : S.ndep[0])[pos];
                    const int bb[5] = {lo, S.b1[j], S.b2[j], S.b3[j], lo + cnt};
                    int k = 0;
                    for (int c = 0; c < 4; c++) k += (bb[c + 1] - bb[c]) > 0;
                    const int before = S.kArr[j] + j;     // children created by earlier-processed nodes
                    int p = totC - (before + k);
                    int epos[4];
                    for (int c = 3; c >= 0; c--) {
                        const int cc2 = bb[c + 1] - bb[c];
                        epos[c] = p;
                        if (cc2 > 0) { (nxt2 ? S.nlo[1] : S.nlo[0])[p] = bb[c]; (nxt2 ? S.ncnt[1] : S.ncnt[0])[p] = cc2; (nxt2 ? S.ndep[1] : S.ndep[0])[p] = dep + 1; p++; }
                    }
                    int es = S.eArr[j];
                    for (int c = 0; c < 4; c++) {
                        const int cc2 = bb[c + 1] - bb[c];
                        if (cc2 > 1) { nEpos[es] = (unsigned) epos[c]; nEcnt[es] = (unsigned) cc2; es++; }
                    }
                }
                __syncthreads();
                for (int j = tid; j < totE2; j += kOctBlock) { S.Epos[j] = (int) nEpos[j]; S.Ecnt[j] = (int) nEcnt[j]; }
                __syncthreads();
                cur = nxt2;
                n = totC + (n - nProc);
                nE = totE2;
                OBFS(-n);
                if (n >= N || n == prev2) finish = true;
            }
        }
    }
    if (kHist && useHist) {
        if (s_overflow) {          // (uniform: every write of the flag lies before the barrier that ended the tree passes)
            __syncthreads();
            useHist = false;
            if (dbg && tid == 0) atomicAdd((unsigned long long *) &dbg[16 * 8 + l], 1ull);   // restarts of this level over the launch's frames
            goto restart;
        }
    }
    OSTAMP(4);
    // ---- 5. best response per node (:702-720), output in list order ----
    unsigned *oxy = lvlKpXY + (long long) f * kpStride + g.kpBase;
    unsigned char *osc = lvlKpScore + (long long) f * kpStride + g.kpBase;
    const int lane = lane_id(), wave = wave_id();
    if (kHist && useHist) {
        // first bin and log2 of the bin count of every final node (PS is still the prefix table)
        for (int i = tid; i < n; i += kOctBlock) {
            const int lo = (cur ? S.nlo[1] : S.nlo[0])[i], sh = 2 * (dm - (cur ? S.ndep[1] : S.ndep[0])[i]);
            int a = 0, b = nBins >> sh;
            while (b - a > 1) {
                const int m = (a + b) >> 1;
                if (PS[m << sh] <= lo) a = m; else b = m;
            }
            S.b1[i] = a << sh;
            S.b2[i] = sh;
            S.kArr[i] = 0;
        }
        __syncthreads();
        // bin -> final node, written over the prefix table (sixteen lanes per node; bins outside every node hold no candidate)
        for (int i0 = wave * 4; i0 < n; i0 += (kOctBlock / 64) * 4) {
            const int i = i0 + (lane >> 4), sl16 = lane & 15;
            if (i < n) {
                const int f0 = S.b1[i], len = 1 << S.b2[i];
                for (int k = sl16; k < len; k += 16) PS[f0 + k] = i;
            }
        }
        __syncthreads();
        // the candidates once more, linearly: best (score, then smallest index) per node
        constexpr int kFU = 4;   // (every load of a round in flight before the first LDS operation: 60 dependent global round trips per thread were 40 us of a 3840x2160 level)
        for (int i0 = tid; i0 < M; i0 += kFU * kOctBlock) {
            unsigned w[kFU];
#pragma unroll
            for (int u = 0; u < kFU; u++) w[u] = key0[min(i0 + u * kOctBlock, M - 1)];
#pragma unroll
            for (int u = 0; u < kFU; u++) {
                const int i = i0 + u * kOctBlock;
                const int node = i < M ? PS[w[u] & 0xFFFFu] : -2 - lane;
                const unsigned v = row_run_max(node, ((w[u] >> 16) << 24) | (0xFFFFFFu - (unsigned) i));
                if (node >= 0 && row_run_last(node)) atomicMax((unsigned *) &S.kArr[node], v);
            }
        }
    } else
    for (int i0 = wave * 4; i0 < n; i0 += (kOctBlock / 64) * 4) {   // sixteen lanes (a DPP row) per node: arg-max over its range (LDS / DPP only);
        const int i = i0 + (lane >> 4), sl16 = lane & 15;           // the final nodes hold ~10 candidates each
        const bool ok = i < n;
        const int lo = ok ? (cur ? S.nlo[1] : S.nlo[0])[i] : 0, cnt = ok ? (cur ? S.ncnt[1] : S.ncnt[0])[i] : 0;
        unsigned best = 0;
        for (int k = sl16; k < cnt; k += 16) best = max(best, svals[lo + k]);
        best = max(best, (unsigned) __builtin_amdgcn_update_dpp(0, (int) best, 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
        best = max(best, (unsigned) __builtin_amdgcn_update_dpp(0, (int) best, 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
        best = max(best, (unsigned) __builtin_amdgcn_update_dpp(0, (int) best, 0x141, 0xF, 0xF, true));   // row_half_mirror
        best = max(best, (unsigned) __builtin_amdgcn_update_dpp(0, (int) best, 0x140, 0xF, 0xF, true));   // row_mirror
        if (ok && sl16 == 0) S.kArr[i] = (int) best;
    }
    __syncthreads();
    for (int i = tid; i < n; i += kOctBlock) {         // a thread per node: the dependent global read of the winner's position, all in flight at once
        const unsigned best = (unsigned) S.kArr[i];
        const unsigned idx = 0xFFFFFFu - (best & 0xFFFFFFu);
        const unsigned p = xy[idx];
        const unsigned kx = (p & 0xFFFFu) + kBorder, ky = (p >> 16) + kBorder;
        oxy[i] = kx | (ky << 16);
        osc[i] = (unsigned char) (best >> 24);
        S.b1[i] = (int) (kx | (ky << 16));      // (the child-boundary arrays are free after the tree passes)
        S.b2[i] = (int) (best >> 24);
        // spatial key for the PROCESSING order of k_describe: 64x64-px tile id, stable within a tile (locality of the
        // 43x43 window gathers; the output order is untouched)
        S.sk[0][i] = ((ky >> 6) << 6) | (kx >> 6);
        S.sv[0][i] = (unsigned) i;
    }
    __syncthreads();
    {
        // (launches of a few frames -- one Tracking frame -- keep the list order: the sort buys cache locality across many frames' windows and
        // costs two block-wide passes of its own)
        unsigned *ok = S.sk[0], *ov = S.sv[0];
        if (gridDim.y > 4) block_radix_sort(S.sk[0], S.sv[0], S.sk[1], S.sv[1], n, 12, histT, s_tmp, &ok, &ov);
        // k_describe's work list: processing position i -> (x | y << 16, score | list position << 8) in ONE record, so that a describe wave
        // knows its keypoint after a single memory round trip (it used to follow procOrder -> position / score: two dependent ones)
        // (score | list position << 8 | level << 24; positions past the level's count carry kNoKeypoint so that the wave there leaves at once)
        uint2 *pr = procRec + (long long) f * kpStride + g.kpBase;
        for (int i = tid; i < g.kpCap; i += kOctBlock) {
            if (i < n) {
                const unsigned li = ov[i];
                pr[i] = make_uint2((unsigned) S.b1[li], (unsigned) S.b2[li] | (li << 8) | ((unsigned) l << 24));
            } else pr[i] = make_uint2(0u, kNoKeypoint);
        }
    }
    if (tid == 0) *lvlCnt = n;
    OSTAMP(5);
    if (dbg && tid == 0 && f == 0) { dbg[l * 8 + 6] = M; dbg[l * 8 + 7] = n; }
#undef OSTAMP
#undef OBFS
}

// ------------------------------------------------------------------------------------------------------------------
// K5+K6+K7  orientation + local blur + rBRIEF.  One wave per keypoint: a 43x43 window of the (un-blurred) level is
// staged in LDS (REFLECT_101 at the image border, as the blur of the border-less clone does); the intensity-centroid
// moments use the inner 31x31 disc; the 7x7 sigma=2 fixed-point Gaussian {18,34,49,55,49,34,18}/256 is evaluated only
// on the 37x37 pixels the rotated pattern can reach (|coord| <= 18); 4 ballots give the 4x64 descriptor bits.
// ------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float fast_atan2_deg(float y, float x) {  // cv::fastAtan2, source order, no FMA
    const float k = (float) (180.0 / 3.14159265358979323846);
    const float p1 = 0.9997878412794807f * k, p3 = -0.3258083974640975f * k, p5 = 0.1555786518463281f * k,
                p7 = -0.04432655554792128f * k;
    const float ax = fabsf(x), ay = fabsf(y);
    float a, c, c2;
    if (ax >= ay) {   // (wave-uniform where one wave = one keypoint: only the taken side executes -- a select-then-divide-once form executed MORE, round 6)
        c = ay / (ax + (float) 2.2204460492503131e-16);
        c2 = c * c;
        a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    } else {
        c = ax / (ay + (float) 2.2204460492503131e-16);
        c2 = c * c;
        a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

// float cos/sin of angle_deg * (float)(pi/180): double-precision quadrant reduction + fdlibm kernel polynomials,
// rounded to float.  Operation sequence identical to the CPU definition (plain * and +, no FMA).
__device__ __forceinline__ void sincos_deg(float angle_deg, float *c_out, float *s_out) {
    const float factorPI = (float) (3.14159265358979323846 / 180.f);
    const float angle = angle_deg * factorPI;
    const double x = (double) angle;
    const double TWO_OVER_PI = 6.36619772367581382433e-01, PIO2_1 = 1.57079632673412561417e+00,
                 PIO2_1T = 6.07710050650619224932e-11;
    const double kd = floor(x * TWO_OVER_PI + 0.5);
    const int k = (int) kd;
    const double r = (x - kd * PIO2_1) - kd * PIO2_1T;
    const double z = r * r;
    const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03, S3 = -1.98412698298579493134e-04,
                 S4 = 2.75573137070700676789e-06, S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03, C3 = 2.48015872894767294178e-05,
                 C4 = -2.75573143513906633035e-07, C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
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

__device__ __forceinline__ int reflect101(int i, int n) {
    if (i < 0) i = -i;
    if (i >= n) i = 2 * n - 2 - i;
    return i;
}

// Orders this wave's LDS writes before its later LDS reads by other lanes (each wave owns a private LDS region, so
// no block barrier is needed -- and none is allowed: waves of a block may exit early).
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

constexpr int kWin = 43, kWinP = 64;    // raw window: one 16-byte aligned 64-byte span per row (+ alignment slack)
constexpr int kHb = 37, kHbP = 40;      // horizontally blurred: 43 rows x 37 cols (u16), pitch even for 32-bit stores
constexpr int kBl = 37, kBlP = 40;      // blurred 37x37 (u8)
// waves (= keypoints) per workgroup of k_describe / k_describe_list.  A workgroup's LDS and wave slots stay allocated until its slowest wave
// is done (interior windows and border windows differ by a factor of several): measured per 256 frames, same box, 1 / 2 / 4 / 8 waves per
// workgroup: 279 / 284 / 292-298 / 315 us -- two it is (236.5 k frames/s in the bench against 233.0 k with four).
#ifndef YGZF_DESC_WAVES
#define YGZF_DESC_WAVES 2
#endif
constexpr int kDescWaves = YGZF_DESC_WAVES;

// Per-wave LDS: the row-blurred window hb (43 rows x 40 u16 = 3440 B) and the raw window (43 rows x 64 B) share memory: hb rows 0..25
// sit in front of raw, hb rows 26..42 overlay raw rows 0..21, which are dead by the time they are written (the orientation moments
// are taken first, the horizontal pass consumes raw rows in ascending order and a wave's LDS operations execute in order).
// 4.9 KB per wave -> the CU's 32 wave slots fill before its LDS does.
constexpr int kHbFront = 26 * kHbP * 2;   // bytes of hb in front of raw (rows 0..25)
struct DescLds {
    __attribute__((aligned(16))) uint8_t mem[kHbFront + kWin * kWinP + 16];
    __device__ __forceinline__ uint8_t *rawp() { return mem + kHbFront; }
    __device__ __forceinline__ unsigned short *hbp() { return (unsigned short *) mem; }
};

// umax of the 31-px circular patch (src/ORBextractor.cc:455-469 for HALF_PATCH_SIZE = 15); ygzf_create checks the context's table against it
constexpr int kUmax15[16] = {15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3};
// Disc membership as byte masks: item = (row v + 15) * 8 + j covers columns u + 15 = 4j .. 4j + 3 of row v; byte k of m[item] is 0xFF
// when (u, v) lies inside the patch (|u| <= umax[|v|]).  Items 248..255 (past the last row) are empty.
struct DiscBytes { unsigned m[256]; };
constexpr DiscBytes make_disc_bytes() {
    DiscBytes d{};
    for (int item = 0; item < 248; item++) {
        const int v = (item >> 3) - 15, j = item & 7;
        const int av = v < 0 ? -v : v;
        unsigned m = 0;
        for (int k = 0; k < 4; k++) {
            const int u = 4 * j + k - 15;
            const int au = u < 0 ? -u : u;
            if (u <= 15 && au <= kUmax15[av]) m |= 0xFFu << (8 * k);
        }
        d.m[item] = m;
    }
    return d;
}
__constant__ DiscBytes c_disc = make_disc_bytes();

__constant__ __attribute__((aligned(16))) float4 c_pattern[256];   // (x0, y0, x1, y1) of test pair p, already as floats
__constant__ int c_umax[16];

// One wave: orientation + blurred patch + 256 rBRIEF bits of the keypoint at integer (kx,ky) of a gw x gh level image.
// CVM = ygzf_cv_mode: which OpenCV generation's 8-bit GaussianBlur the descriptor samples (include/ygzf.h):
//   YGZF_CV_LEGACY_SSE2  kernel {18,34,49,55,49,34,18}; the x86 column pass rounds the exact quotient sum/65536 half to EVEN on the
//                        columns of its SSE2 body ([0, width & ~3)) and half UP ((sum + 2^15) >> 16) on the scalar tail
//   YGZF_CV_LEGACY_INT   same kernel, (sum + 2^15) >> 16 everywhere
//   YGZF_CV_4            Q8.8 kernel {18,34,48,56,48,34,18}, (sum + 2^15) >> 16
template <int CVM>
__device__ __forceinline__ void describe_window(const uint8_t *__restrict__ img, int pitch, int gw, int gh, int kx, int ky, DescLds &L,
                                                int lane, float *angleOut, unsigned long long bits[4], PhaseClk &pc, bool presetAngle = false,
                                                float preset = 0.f) {
    // ---- stage the 43x43 window (rows ky-21..ky+21).  Interior keypoints: the 43 bytes of a row lie inside one 16-byte
    // aligned 64-byte span -> 4 lanes x dwordx4 per row, 3 wave-level loads for the whole window, all in flight at once.
    // Keypoints within 21 px of the image border need BORDER_REFLECT_101 (what the blur of the border-less level clone
    // sees) and take the byte path.  Window column c of row r lives at raw[r*64 + ((rowOff0 + r*rowOffStep) & 15) + c].
    unsigned discM[4];   // disc membership bytes of this lane's four centroid work items: loaded first, used after the window is staged
#pragma unroll
    for (int it = 0; it < 4; it++) discM[it] = c_disc.m[it * 64 + lane];
    const bool interior = kx >= 21 && kx + 21 < gw && ky >= 21 && ky + 22 < gh && ((pitch & 3) == 0);
    int rowOff0 = 0, rowOffStep = 0;
    if (interior) {
        const unsigned long long a00 = (unsigned long long) img + (unsigned) (ky - 21) * (unsigned) pitch + (unsigned) (kx - 21);
        rowOff0 = (int) (a00 & 15);
        rowOffStep = pitch & 15;
        // LDS-DMA: lane idx & 63 of piece idx >> 6 delivers LDS bytes 16 idx .. 16 idx + 15 = row idx >> 2, quarter idx & 3 -- the window's own
        // layout, so the 43 x 64 bytes go global memory -> LDS in three instructions without a register round trip or a ds_write
#pragma unroll
        for (int i0 = 0; i0 < kWin * 4; i0 += 64) {
            const int idx = i0 + lane, r = idx >> 2, q = idx & 3;
            const unsigned long long a = a00 + __umul24((unsigned) r, (unsigned) pitch);
            if (idx < kWin * 4)
                __builtin_amdgcn_global_load_lds((const unsigned *) ((a & ~15ull) + 16 * q), (lds_void_t *) (L.rawp() + 16 * i0), 16, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
        for (int idx = lane; idx < kWin * kWin; idx += 64) {
            const int r = idx / kWin, c = idx - r * kWin;
            const int yy = reflect101(ky - 21 + r, gh), xx = reflect101(kx - 21 + c, gw);
            L.rawp()[r * kWinP + c] = img[(long long) yy * pitch + xx];
        }
    }
#define RAWP(r) (&L.rawp()[(r) * kWinP + ((rowOff0 + (r) * rowOffStep) & 15)])
    wave_lds_sync();
    pc.mark(0);
    // ---- intensity centroid on the 31x31 disc (centre = window (21,21)).  Work item = four pixels (row v, columns u = 4j-15 .. 4j-12):
    // two aligned dwords shifted into place, masked with the disc membership bytes, then m01 += v * (sum of the bytes) [v_sad_u8] and
    // m10 += <bytes, (4j .. 4j+3)> - 15 * sum [v_dot4_u32_u8].  248 items = 4 steps of the wave; integer sums, any order is exact.
    int m10, m01;
    {
        unsigned wsum = 0, ssum = 0;
        int vsum = 0;
        const unsigned ucol = 0x03020100u + 0x04040404u * (unsigned) (lane & 7);
#pragma unroll
        for (int it = 0; it < 4; it++) {
            const int r = 6 + it * 8 + (lane >> 3);                      // window row 21 + v
            const unsigned A = (unsigned) (RAWP(r) - L.rawp()) + 6u + 4u * (unsigned) (lane & 7);
            const unsigned *dw = (const unsigned *) L.rawp() + (A >> 2);
            const unsigned px = __builtin_amdgcn_alignbyte(dw[1], dw[0], A & 3u) & discM[it];
            const unsigned sm = __builtin_amdgcn_sad_u8(px, 0u, 0u);
            wsum = __builtin_amdgcn_udot4(px, ucol, wsum, false);
            ssum += sm;
            vsum += __mul24(r - 21, (int) sm);
        }
        m10 = (int) wsum - __mul24(15, (int) ssum);
        m01 = vsum;
    }
    m10 = wave_sum(m10);
    m01 = wave_sum(m01);
    pc.mark(1);
    const float angle = presetAngle ? preset : fast_atan2_deg((float) m01, (float) m10);
#ifdef YGZF_PHASE_CLOCK
    asm volatile("" :: "v"(angle));   // the stamp below must see the angle computed
#endif
    pc.mark(2);
    // ---- separable 7-tap blur {18,34,k2,k3,k2,34,18}: each lane produces runs of 8 outputs
    constexpr int k2 = CVM == YGZF_CV_4 ? 48 : 49, k3 = CVM == YGZF_CV_4 ? 56 : 55;
    constexpr unsigned kTapLo = 0x00002212u | ((unsigned) k2 << 16) | ((unsigned) k3 << 24);   // bytes (18, 34, k2, k3)
    constexpr unsigned kTapHi = 0x00122200u | (unsigned) k2;                                  // bytes (k2, 34, 18, 0)
    // horizontal: 43 rows x 4 segments of 10 columns (the last one's columns 37..39 land in the row's slack and are never read): 172 items =
    // three steps of the wave (eight-column segments made 215 items: a fourth step for 23 of the 64 lanes)
    for (int t = lane; t < kWin * 4; t += 64) {
        const int r = t >> 2, sg = t & 3;
        // 16 source bytes from an arbitrary byte address: five aligned dwords, shifted into place once (e[k] = bytes 4k..4k+3 of the
        // run); output j = <bytes j..j+3, (18,34,k2,k3)> + <bytes j+4..j+7, (k2,34,18,0)> -- two v_dot4_u32_u8 per output, exact
        // (the sum is at most 65535); byte j+7 meets the zero tap, so output 9 never needs a sixth dword.
        const unsigned A = (unsigned) (RAWP(r) - L.rawp()) + 10u * (unsigned) sg;
        const unsigned *dw = (const unsigned *) L.rawp() + (A >> 2);
        const unsigned sh = A & 3u;
        const unsigned d0 = dw[0], d1 = dw[1], d2 = dw[2], d3 = dw[3], d4 = dw[4];
        // e[m] = bytes 4m .. 4m + 3 of the run.  Output j = 4m + r sums bytes j .. j + 6 against (18, 34, k2, k3, k2, 34, 18): instead of shifting the
        // bytes into place for every j (three more v_alignbyte per dword) the TAPS are shifted -- eight constant words -- and the aligned dwords meet
        // them as they are: r = 0, 1 take two v_dot4 (e[m], e[m + 1]), r = 2, 3 three (+ e[m + 2]): 24 + 4 instead of 20 + 14 instructions per ten outputs.
        const unsigned e[4] = {__builtin_amdgcn_alignbyte(d1, d0, sh), __builtin_amdgcn_alignbyte(d2, d1, sh), __builtin_amdgcn_alignbyte(d3, d2, sh),
                               __builtin_amdgcn_alignbyte(d4, d3, sh)};
        constexpr unsigned T0a = kTapLo, T0b = kTapHi;                                                                     // r = 0: (18,34,k2,k3) (k2,34,18,0)
        constexpr unsigned T1a = kTapLo << 8, T1b = (kTapLo >> 24) | (kTapHi << 8);                                        // r = 1: (0,18,34,k2) (k3,k2,34,18)
        constexpr unsigned T2a = kTapLo << 16, T2b = (kTapLo >> 16) | (kTapHi << 16), T2c = kTapHi >> 16;                  // r = 2: (0,0,18,34) (k2,k3,k2,34) (18,0,0,0)
        constexpr unsigned T3a = kTapLo << 24, T3b = (kTapLo >> 8) | (kTapHi << 24), T3c = kTapHi >> 8;                    // r = 3: (0,0,0,18) (34,k2,k3,k2) (34,18,0,0)
        auto out4 = [&](unsigned a, unsigned b, unsigned c, unsigned (&o)[4]) {
            o[0] = __builtin_amdgcn_udot4(a, T0a, __builtin_amdgcn_udot4(b, T0b, 0u, false), false);
            o[1] = __builtin_amdgcn_udot4(a, T1a, __builtin_amdgcn_udot4(b, T1b, 0u, false), false);
            o[2] = __builtin_amdgcn_udot4(a, T2a, __builtin_amdgcn_udot4(b, T2b, __builtin_amdgcn_udot4(c, T2c, 0u, false), false), false);
            o[3] = __builtin_amdgcn_udot4(a, T3a, __builtin_amdgcn_udot4(b, T3b, __builtin_amdgcn_udot4(c, T3c, 0u, false), false), false);
        };
        unsigned oa[4], ob[4];
        out4(e[0], e[1], e[2], oa);
        out4(e[1], e[2], e[3], ob);
        const unsigned o8 = __builtin_amdgcn_udot4(e[2], T0a, __builtin_amdgcn_udot4(e[3], T0b, 0u, false), false);
        const unsigned o9 = __builtin_amdgcn_udot4(e[2], T1a, __builtin_amdgcn_udot4(e[3], T1b, 0u, false), false);
        const unsigned O[5] = {oa[0] | (oa[1] << 16), oa[2] | (oa[3] << 16), ob[0] | (ob[1] << 16), ob[2] | (ob[3] << 16), o8 | (o9 << 16)};
        unsigned *dst = (unsigned *) &L.hbp()[r * kHbP + 10 * sg];
        dst[0] = O[0]; dst[1] = O[1]; dst[2] = O[2]; dst[3] = O[3]; dst[4] = O[4];
    }
    wave_lds_sync();
    pc.mark(3);
    // vertical pass only where the rotated pattern samples (512 points instead of the 37 x 37 patch): 7 taps straight from hb
    float a, b;
    sincos_deg(angle, &a, &b);
#ifdef YGZF_PHASE_CLOCK
    asm volatile("" :: "v"(a), "v"(b));
#endif
    pc.mark(4);
#pragma unroll
    for (int it = 0; it < 4; it++) {
        const int p = it * 64 + lane;
        const float4 pk = c_pattern[p];
        const float x0 = pk.x, y0 = pk.y, x1 = pk.z, y1 = pk.w;
        const int r0 = __float2int_rn(x0 * b + y0 * a), q0 = __float2int_rn(x0 * a - y0 * b);
        const int r1 = __float2int_rn(x1 * b + y1 * a), q1 = __float2int_rn(x1 * a - y1 * b);
        const unsigned short *p0 = &L.hbp()[(18 + r0) * kHbP + 18 + q0], *p1 = &L.hbp()[(18 + r1) * kHbP + 18 + q1];
        const int s0 = 18 * (p0[0] + p0[6 * kHbP]) + 34 * (p0[kHbP] + p0[5 * kHbP]) + k2 * (p0[2 * kHbP] + p0[4 * kHbP]) + k3 * p0[3 * kHbP];
        const int s1 = 18 * (p1[0] + p1[6 * kHbP]) + 34 * (p1[kHbP] + p1[5 * kHbP]) + k2 * (p1[2 * kHbP] + p1[4 * kHbP]) + k3 * p1[3 * kHbP];
        int t0 = (s0 + 32768) >> 16, t1 = (s1 + 32768) >> 16;
        if (CVM == YGZF_CV_LEGACY_SSE2) {
            // exact tie (sum = q*65536 + 32768) inside the SSE2 body: cvtps2dq rounds to even, i.e. the half-up result loses its low bit.
            // A tie is one sample in tens of thousands: the column bookkeeping it needs (ten vector instructions per sample, a tenth of this
            // phase) runs only when some lane of the wave has one
            const bool tie0 = (s0 & 0xFFFF) == 0x8000, tie1 = (s1 & 0xFFFF) == 0x8000;
            if (__builtin_expect(__ballot(tie0 || tie1) != 0ull, 0)) {
                const int wv = gw & ~3;
                const int c0 = reflect101(kx + q0, gw), c1 = reflect101(kx + q1, gw);
                if (tie0 && c0 < wv) t0 &= ~1;
                if (tie1 && c1 < wv) t1 &= ~1;
            }
        }
        t0 = min(t0, 255);
        t1 = min(t1, 255);
        bits[it] = __ballot(t0 < t1);
    }
    pc.mark(5);
    *angleOut = angle;
#undef RAWP
}

// Output bookkeeping between the octree and k_describe: first output index of every level of every frame (the levels' keypoints are
// concatenated in level order, src/ORBextractor.cc:1004-1026) and the frame's keypoint count.  One thread per frame.
__global__ void k_level_bases(const int *__restrict__ lvlKpCnt, int nlevels, int nFrames, int *__restrict__ lvlBase, int *__restrict__ outCnt) {
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= nFrames) return;
    int acc = 0;
    for (int l = 0; l < nlevels; l++) {
        lvlBase[f * kMaxLevels + l] = acc;
        acc += lvlKpCnt[f * nlevels + l];
    }
    outCnt[f] = acc;
}

// kOwnBases (launches of a few frames: one Tracking frame): the wave sums the level counts itself (lvlBase = the octree's per-level counts,
// nlevels of them per frame) and the wave at processing position 0 writes the frame's keypoint count -- k_level_bases, a launch of its own
// between two short kernels, is not needed then.
template <int CVM, bool kOwnBases>
__global__ __launch_bounds__(64 * kDescWaves) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_describe(FrameSet fs, const LevelGeom *__restrict__ geom,
                                                            const int *__restrict__ lvlBase,
                                                            const uint2 *__restrict__ procRec, int kpStride,
                                                            ygzf_kp *__restrict__ outKp, uint8_t *__restrict__ outDesc,
                                                            int outStride, int blocksPerXcd, int nlevels, int *__restrict__ outCnt) {
    __shared__ DescLds lds[kDescWaves];
    const int lane = lane_id(), wave = __builtin_amdgcn_readfirstlane(wave_id());   // wave-uniform: the keypoint record loads go scalar
    PhaseClk pc;
    pc.start();
    const int f = blockIdx.y;
    // XCD-aware: workgroup b runs on XCD b % 8; every XCD gets a contiguous run of (spatially ordered) keypoint slots so
    // that overlapping 43x43 windows meet in the same L2.  s = processing position inside the frame's level-major capacity layout.
    if ((int) (blockIdx.x >> 3) >= blocksPerXcd) return;
    const int s = ((blockIdx.x & 7) * blocksPerXcd + (blockIdx.x >> 3)) * kDescWaves + wave;
    if (s >= kpStride) return;
    // The octree leaves everything the wave needs to know about its keypoint in ONE 8-byte record (position, score, list position,
    // level; kNoKeypoint where the level has fewer keypoints than capacity), k_level_bases the level's first output index: two scalar
    // loads and a few scalar operations instead of a 16-way level search and a loop over the level counts (counters: as many issue
    // slots went into this bookkeeping as into the row blur).
    const uint2 rec = procRec[(long long) f * kpStride + s];
    if (kOwnBases && s == 0) {
        int tot = 0;
        for (int q = 0; q < nlevels; q++) tot += lvlBase[f * nlevels + q];
        if (lane == 0) outCnt[f] = tot;
    }
    if (rec.y == kNoKeypoint) return;
    const int l = (int) (rec.y >> 24);
    const int li = (int) ((rec.y >> 8) & 0xFFFFu);                                  // list position handled by this wave
    int slot = li;                                                                  // output index: level-major, list order
    if (kOwnBases) {
        for (int q = 0; q < l; q++) slot += lvlBase[f * nlevels + q];
    } else slot += lvlBase[f * kMaxLevels + l];
    const LevelGeom *gp = geom + l;
    const int gw = gp->w, gh = gp->h;
    const int kx = rec.x & 0xFFFFu, ky = rec.x >> 16;
    const int score = rec.y & 0xFFu;
    int pitch;
    const uint8_t *img;
    if (l == 0) { pitch = fs.img0_pitch; img = fs.img0 + (long long) f * fs.img0_stride; }
    else { pitch = gp->pitch; img = fs.pyr + (long long) f * fs.pyr_stride + gp->off; }
    float angle;
    unsigned long long bits[4];
    describe_window<CVM>(img, pitch, gw, gh, kx, ky, lds[wave], lane, &angle, bits, pc);
    ygzf_kp *ok = outKp + (long long) f * outStride + slot;
    uint8_t *od = outDesc + ((long long) f * outStride + slot) * 32;
    if (lane < 4) ((unsigned long long *) od)[lane] = bits[lane];
    if (lane == 0) {
        ygzf_kp kp;
        if (l == 0) { kp.x = (float) kx; kp.y = (float) ky; }
        else { const float sc = gp->scale; kp.x = (float) kx * sc; kp.y = (float) ky * sc; }  // keypoint->pt *= scale (:1016-1021)
        kp.size = gp->kpSize;
        kp.angle = angle;
        kp.response = (float) score;
        kp.octave = l;
        kp.class_id = -1;
        *ok = kp;
    }
    pc.mark(6);
    pc.flush(1, lane, (unsigned) s + (unsigned) f * (unsigned) kpStride);
}

// Orientation + descriptor of an explicit keypoint list (the Frame* overload, src/ORBextractor.cc:1031-1127: keys that already
// exist in the frame get IC_Angle (:1380-1383) and a descriptor at pt*invScale[octave] (:1104-1116), new level-0 keys follow).
// list[i] = {x, y, level | flag, angle bits} in LEVEL coordinates (already rounded as cvRound does); flag 0x100: keep the given angle
// (the ORBSLAM_KEYPOINT / FAST_KEYPOINT branches leave the angle of existing keys alone, :1096-1099).
template <int CVM>
__global__ __launch_bounds__(64 * kDescWaves) void k_describe_list(FrameSet fs, const LevelGeom *__restrict__ geom, const int4 *__restrict__ list,
                                                                 int n, int frame, float *__restrict__ outAngle, uint8_t *__restrict__ outDesc) {
    __shared__ DescLds lds[kDescWaves];
    const int lane = lane_id(), wave = wave_id();
    const int i = blockIdx.x * kDescWaves + wave;
    if (i >= n) return;
    const int4 e = list[i];
    const int level = e.z & 0xFF;
    const LevelGeom g = geom[level];
    int pitch;
    const uint8_t *img = level_ptr(fs, g, level, frame, &pitch);
    float angle;
    unsigned long long bits[4];
    PhaseClk pc;   // (the phase clock reports k_describe; here it is only the argument)
    pc.start();
    describe_window<CVM>(img, pitch, g.w, g.h, e.x, e.y, lds[wave], lane, &angle, bits, pc, (e.z & 0x100) != 0, __int_as_float(e.w));
    if (lane < 4) ((unsigned long long *) (outDesc + (long long) i * 32))[lane] = bits[lane];
    if (lane == 0) outAngle[i] = angle;
}

// batched DescriptorDistance: popcount over 4 x u64
__global__ void k_hamming_pairs(const unsigned long long *__restrict__ a, const unsigned long long *__restrict__ b, int n,
                                int *__restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int d = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) d += __popcll(a[4 * i + k] ^ b[4 * i + k]);
    out[i] = d;
}

// ------------------------------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------------------------------
hipError_t upload_constants(const int *umax16) {
    for (int i = 0; i < 16; i++)
        if (umax16[i] != kUmax15[i]) return hipErrorInvalidValue;   // the orientation kernel's disc masks are built for this table
    float pf[1024];
    for (int i = 0; i < 1024; i++) pf[i] = (float) kBriefPattern[i];
    hipError_t e = hipMemcpyToSymbol(HIP_SYMBOL(c_pattern), pf, sizeof pf);
    if (e != hipSuccess) return e;
    return hipMemcpyToSymbol(HIP_SYMBOL(c_umax), umax16, 16 * sizeof(int));
}

// All levels of frame 0 as tight (pitch = width) images, one after the other, into a linear buffer: what ORBextractor::mvImagePyramid holds
// on the host.  The device levels have 64-byte pitches; a pitched device-to-host copy of an odd-width level is executed row by row by the
// copy engine (1.3 ms per level), so the levels are packed here and leave in ONE linear copy.
struct PackOffsets { unsigned v[kMaxLevels + 1]; };
// A thread produces 16 consecutive bytes of the packed stream (one aligned 16-byte store: the destination may be page-locked HOST memory, written over
// the link, where byte stores would be a transaction each): it finds the level and the position of its first byte, then walks along the row, into the
// next row, into the next level.
__global__ __launch_bounds__(256) void k_pack_levels(FrameSet fs, const LevelGeom *__restrict__ geom, PackOffsets off, uint8_t *__restrict__ dst, int firstLevel, int nlevels) {
    const unsigned total = off.v[nlevels], first = off.v[firstLevel];
    for (unsigned o = first + 16u * (blockIdx.x * 256u + threadIdx.x); o < total; o += 16u * gridDim.x * 256u) {
        int l = firstLevel;
        while (l + 1 < nlevels && o >= off.v[l + 1]) l++;
        LevelGeom g = geom[l];
        int pitch;
        const uint8_t *src = level_ptr(fs, g, l, 0, &pitch);
        unsigned i = o - off.v[l], y = i / (unsigned) g.w, x = i - y * (unsigned) g.w;
        const uint8_t *row = src + (size_t) y * pitch;
        unsigned w4[4] = {0u, 0u, 0u, 0u};
        const unsigned nb = min(16u, total - o);
#pragma unroll
        for (unsigned b = 0; b < 16; b++) {
            if (b < nb) {
                w4[b >> 2] |= (unsigned) row[x] << (8 * (b & 3));
                if (++x == (unsigned) g.w) {
                    x = 0;
                    if (++y == (unsigned) g.h) {            // next level (never past the last one: b < nb)
                        if (l + 1 < nlevels) { l++; g = geom[l]; src = level_ptr(fs, g, l, 0, &pitch); }
                        y = 0;
                    }
                    row = src + (size_t) y * pitch;
                }
            }
        }
        if (nb == 16) *(uint4 *) (dst + o) = make_uint4(w4[0], w4[1], w4[2], w4[3]);
        else for (unsigned b = 0; b < nb; b++) dst[o + b] = (uint8_t) (w4[b >> 2] >> (8 * (b & 3)));
    }
}

void launch_pack_levels(hipStream_t st, const FrameSet &fs, const LevelGeom *dGeom, int firstLevel, int nlevels, const unsigned *offsets, uint8_t *dst) {
    if (firstLevel >= nlevels) return;
    PackOffsets off;
    for (int l = 0; l <= kMaxLevels; l++) off.v[l] = l <= nlevels ? offsets[l] : 0;
    const unsigned n16 = (offsets[nlevels] - offsets[firstLevel] + 15u) / 16u;
    if (!n16) return;
    hipLaunchKernelGGL(k_pack_levels, dim3(std::min(256u, (n16 + 255u) / 256u)), dim3(256), 0, st, fs, dGeom, off, dst, firstLevel, nlevels);
}

// rows of `w` bytes from a linear staging buffer (row pitch srcPitch) into a pitched image: the device half of an upload whose pitched
// host-to-device copy the copy engine would execute row by row (widths or pitches that are not multiples of 4: 2.6-3.3 ms for a 1241x376
// or 641x479 frame instead of 0.1 ms)
__global__ __launch_bounds__(256) void k_repitch_rows(const uint8_t *__restrict__ src, unsigned srcPitch, uint8_t *__restrict__ dst, unsigned dstPitch,
                                                      unsigned w, unsigned long long n) {
    for (unsigned long long i = (unsigned long long) blockIdx.x * 256u + threadIdx.x; i < n; i += (unsigned long long) gridDim.x * 256u) {
        const unsigned long long y = i / w;
        const unsigned x = (unsigned) (i - y * w);
        dst[y * dstPitch + x] = src[y * srcPitch + x];
    }
}

void launch_repitch_rows(hipStream_t st, const uint8_t *src, size_t srcPitch, uint8_t *dst, size_t dstPitch, int w, size_t rows) {
    const unsigned long long n = (unsigned long long) w * rows;
    if (n == 0) return;
    const unsigned blocks = (unsigned) std::min<unsigned long long>((n + 1023) / 1024, 4096);
    hipLaunchKernelGGL(k_repitch_rows, dim3(blocks), dim3(256), 0, st, src, (unsigned) srcPitch, dst, (unsigned) dstPitch, (unsigned) w, n);
}

// Frames that lie in page-locked, device-visible HOST memory, read by the kernel itself over the link (no copy-engine call per frame: a batch whose
// frames do not sit at one stride -- a round-robin share of a clip, the rows of an array of cv::Mat -- would cost one pitched copy per frame, ~40 us
// of copy-engine set-up each).  Block (x, frame): 16 bytes per lane and step where source and destination allow it (a wave asks for 1 KB of
// consecutive host bytes at a time), single bytes at the ragged ends.
__global__ __launch_bounds__(256) void k_gather_host_frames(HostFrameList L, unsigned srcPitch, uint8_t *__restrict__ dst, unsigned dstPitch,
                                                            unsigned long long dstFrameStride, unsigned w, unsigned h) {
    const uint8_t *src = (const uint8_t *) L.addr[blockIdx.y];
    uint8_t *out = dst + (unsigned long long) blockIdx.y * dstFrameStride;
    const bool vec = (((unsigned long long) src | srcPitch) & 15u) == 0 && (dstPitch & 15u) == 0;   // (dst is 256-byte aligned: the context's own buffer)
    const unsigned wq = vec ? w >> 4 : 0;                       // whole 16-byte pieces per row
    const unsigned long long nq = (unsigned long long) wq * h;
    for (unsigned long long i = (unsigned long long) blockIdx.x * 256u + threadIdx.x; i < nq; i += (unsigned long long) gridDim.x * 256u) {
        const unsigned y = (unsigned) (i / wq), q = (unsigned) (i - (unsigned long long) y * wq);
        *(uint4 *) (out + (unsigned long long) y * dstPitch + 16u * q) = *(const uint4 *) (src + (unsigned long long) y * srcPitch + 16u * q);
    }
    const unsigned tail = w - 16u * wq;                          // the ragged end of every row (or whole rows when nothing is aligned)
    const unsigned long long nt = (unsigned long long) tail * h;
    for (unsigned long long i = (unsigned long long) blockIdx.x * 256u + threadIdx.x; i < nt; i += (unsigned long long) gridDim.x * 256u) {
        const unsigned y = (unsigned) (i / tail), x = 16u * wq + (unsigned) (i - (unsigned long long) y * tail);
        out[(unsigned long long) y * dstPitch + x] = src[(unsigned long long) y * srcPitch + x];
    }
}

void launch_gather_host_frames(hipStream_t st, const HostFrameList &L, int nFrames, size_t srcPitch, uint8_t *dst, size_t dstPitch, size_t dstFrameStride, int w, int h) {
    if (nFrames <= 0 || w <= 0 || h <= 0) return;
    const unsigned long long n16 = ((unsigned long long) w * h + 15) / 16;
    const unsigned blocks = (unsigned) std::min<unsigned long long>((n16 + 255) / 256, 64);
    hipLaunchKernelGGL(k_gather_host_frames, dim3(blocks, nFrames), dim3(256), 0, st, L, (unsigned) srcPitch, dst, (unsigned) dstPitch,
                       (unsigned long long) dstFrameStride, (unsigned) w, (unsigned) h);
}

// The pyramid chain of ONE frame (levels 1 .. nlevels - 1, each from the one before) as an explicit graph: built node by node (no stream
// capture, which would put process-wide restrictions on other threads' calls while it is open), retargeted to another frame's buffers by
// rewriting the nodes' FrameSet argument, launched with one call.
static void pyr_node_params(hipKernelNodeParams *np, void **args, PyrChainGraph *pg, int l) {
    const LevelGeom &g = pg->lv[l];
    const bool perPixel = g.area2x || !g.tiledOk;
    std::memset(np, 0, sizeof *np);
    np->blockDim = dim3(perPixel ? 256 : kPyrThreads);
    np->sharedMemBytes = 0;
    np->kernelParams = args;
    np->extra = nullptr;
    args[0] = &pg->fs; args[1] = &pg->geom; args[2] = &pg->level[l];
    if (perPixel) {
        args[3] = &pg->tabs.xofs; args[4] = &pg->tabs.xalpha; args[5] = &pg->tabs.yofs; args[6] = &pg->tabs.ybeta;
        np->func = (void *) k_pyr_resize;
        np->gridDim = dim3((g.w + 255) / 256, g.h, 1);
    } else {
        args[3] = &pg->tabs.cols; args[4] = &pg->tabs.rows; args[5] = &pg->tabs.tiles;
        np->func = (void *) k_pyr_resize_tiled;
        np->gridDim = dim3(g.nPyrTiles, 1, 1);
    }
}

hipError_t pyr_chain_graph_build(PyrChainGraph *pg, const FrameSet &fs, const LevelGeom *dGeom, const LevelGeom *lv, int nlevels, const PyrTabs &T) {
    std::memset(pg, 0, sizeof *pg);
    hipError_t e = hipGraphCreate(&pg->graph, 0);
    if (e != hipSuccess) return e;
    pg->fs = fs; pg->geom = dGeom; pg->tabs = T;
    pg->nNodes = 0;
    for (int l = 1; l < nlevels; l++) {
        pg->level[l] = l;
        pg->lv[l] = lv[l];
        void *args[7];
        hipKernelNodeParams np;
        pyr_node_params(&np, args, pg, l);
        e = hipGraphAddKernelNode(&pg->nodes[l], pg->graph, l > 1 ? &pg->nodes[l - 1] : nullptr, l > 1 ? 1 : 0, &np);
        if (e != hipSuccess) break;
        pg->nNodes = l;
    }
    if (e == hipSuccess) e = hipGraphInstantiate(&pg->exec, pg->graph, nullptr, nullptr, 0);
    if (e != hipSuccess) pyr_chain_graph_destroy(pg);
    return e;
}

hipError_t pyr_chain_graph_retarget(PyrChainGraph *pg, const FrameSet &fs) {
    pg->fs = fs;
    for (int l = 1; l <= pg->nNodes; l++) {
        void *args[7];
        hipKernelNodeParams np;
        pyr_node_params(&np, args, pg, l);
        const hipError_t e = hipGraphExecKernelNodeSetParams(pg->exec, pg->nodes[l], &np);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

void pyr_chain_graph_destroy(PyrChainGraph *pg) {
    if (pg->exec) (void) hipGraphExecDestroy(pg->exec);
    if (pg->graph) (void) hipGraphDestroy(pg->graph);
    pg->exec = nullptr;
    pg->graph = nullptr;
    pg->nNodes = 0;
}

void launch_pyr_resize(hipStream_t st, const FrameSet &fs, const LevelGeom *dGeom, const LevelGeom &g, int level, int nFrames, const PyrTabs &T) {
    if (g.area2x || !g.tiledOk) {   // exact 2x levels (area mean) and steep pyramids (a tile's source would not fit LDS): per-pixel kernel
        dim3 grid((g.w + 255) / 256, g.h, nFrames);
        hipLaunchKernelGGL(k_pyr_resize, grid, dim3(256), 0, st, fs, dGeom, level, T.xofs, T.xalpha, T.yofs, T.ybeta);
        return;
    }
    hipLaunchKernelGGL(k_pyr_resize_tiled, dim3(g.nPyrTiles, nFrames), dim3(kPyrThreads), 0, st, fs, dGeom, level, T.cols, T.rows, T.tiles);
}

// The attribute belongs to the kernel on a device, not to a context: contexts of different geometries share it, so it only ever grows (a
// context of small images prepared after one of large images must not take the large one's LDS allotment away).
hipError_t pyr_strips_prepare(size_t ldsBytes) {
    static std::mutex mu;
    static size_t granted[64] = {0};
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= 64) return hipFuncSetAttribute((const void *) k_pyr_strips, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    std::lock_guard<std::mutex> lock(mu);
    if (ldsBytes <= granted[dev]) return hipSuccess;
    e = hipFuncSetAttribute((const void *) k_pyr_strips, hipFuncAttributeMaxDynamicSharedMemorySize, (int) ldsBytes);
    if (e == hipSuccess) granted[dev] = ldsBytes;
    return e;
}

void launch_pyr_strips(hipStream_t st, const FrameSet &fs, int nlevels, const PyrStripPlan *plans, const PyrStripLevel *levels, int nStrips,
                       int offA, int offB, size_t ldsBytes, int nFrames, const int *xofs, const short *xalpha, const int *yofs, const short *ybeta) {
    hipLaunchKernelGGL(k_pyr_strips, dim3(nStrips, nFrames), dim3(kPyrStripThreads), ldsBytes, st, fs, nlevels, plans, levels, offA, offB, xofs, xalpha, yofs, ybeta);
}

size_t fast_quads_lds_bytes(int winPitch, int winRows, int smapRows, int quadCap) {
    const size_t smapBytes = (std::max((size_t) smapRows * winPitch, (size_t) quadCap * 4) + 15) & ~(size_t) 15;   // score map and quad list share their bytes
    return (size_t) (kFastBlock / 64) * ((((size_t) winRows * winPitch + 16 + 15) & ~(size_t) 15) + smapBytes + kCornerCap * sizeof(unsigned short)) + 64;
}

void launch_fast_cells(hipStream_t st, const FrameSet &fs, const LevelGeom *dGeom, int nlevels, int iniTh, int minTh,
                       unsigned short *cellCnt, unsigned *slots, int totalCells, long long totalSlots, int totalGroups, int smapRows,
                       int nFrames, int winPitch, int winRows, int quadCap, const int *groupBaseHost, bool iniFirst, unsigned *stats) {
    if (totalGroups <= 0) return;
    FastGroupBases gb;
    for (int k = 0; k < kMaxLevels; k++) gb.v[k] = k < nlevels ? groupBaseHost[k] : 0x7fffffff;
    const int groupsPerXcd = ((totalGroups + kFastRep - 1) / kFastRep + 7) / 8;    // workgroups (of kFastRep groups) per XCD
    const dim3 grid(8 * groupsPerXcd, nFrames), block(kFastBlock);
    const size_t lds = fast_quads_lds_bytes(winPitch, winRows, smapRows, quadCap);
#define YGZF_FAST_LAUNCH(KP)                                                                                                                       \
    do {                                                                                                                                           \
        if (iniFirst)                                                                                                                              \
            hipLaunchKernelGGL((k_fast_quads<KP, true>), grid, block, lds, st, fs, dGeom, nlevels, iniTh, minTh, cellCnt, slots, totalCells,       \
                               totalSlots, totalGroups, groupsPerXcd, winPitch, winRows, smapRows, quadCap, gb, stats);                            \
        else                                                                                                                                       \
            hipLaunchKernelGGL((k_fast_quads<KP, false>), grid, block, lds, st, fs, dGeom, nlevels, iniTh, minTh, cellCnt, slots, totalCells,      \
                               totalSlots, totalGroups, groupsPerXcd, winPitch, winRows, smapRows, quadCap, gb, stats);                            \
    } while (0)
    switch (winPitch) {   // the usual pitches (cells of 30..41 pixels) get immediate LDS offsets; anything else the run-time pitch
        case 40: YGZF_FAST_LAUNCH(40); break;
        case 44: YGZF_FAST_LAUNCH(44); break;
        case 48: YGZF_FAST_LAUNCH(48); break;
        default: YGZF_FAST_LAUNCH(0); break;
    }
#undef YGZF_FAST_LAUNCH
}

// ---- table-driven form (k_fast_tab) ----
size_t fast_tab_lds_bytes(int winRows, int smapRows, int quadCap) {
    return (fast_quads_lds_bytes(kTabPitch, winRows, smapRows, quadCap) - 64) / (kFastBlock / 64) * kFastTabWaves + 64;
}
void launch_fast_tab(hipStream_t st, const FrameSet &fs, const FastCellRec *dCells, int iniTh, int minTh, unsigned short *cellCnt, unsigned *slots,
                     int totalCells, long long totalSlots, int totalGroups, int smapRows, int nFrames, int winRows, int quadCap, bool iniFirst,
                     unsigned *stats) {
    if (totalGroups <= 0) return;
    const int totalWgs = totalGroups * (4 / kFastTabWaves);                 // workgroups of kFastTabWaves cells
    const int groupsPerXcd = (totalWgs + 7) / 8;
    const dim3 grid(8 * groupsPerXcd, nFrames), block(64 * kFastTabWaves);
    const size_t lds = fast_tab_lds_bytes(winRows, smapRows, quadCap);
#define YGZF_LAUNCH_TAB(INI) hipLaunchKernelGGL((k_fast_tab<INI>), grid, block, lds, st, fs, dCells, iniTh, minTh, cellCnt, slots, totalCells, totalSlots, totalWgs, \
                                                   groupsPerXcd, winRows, smapRows, quadCap, stats)
    if (iniFirst) YGZF_LAUNCH_TAB(true); else YGZF_LAUNCH_TAB(false);
#undef YGZF_LAUNCH_TAB
}

// persistent form: `counters` = kFastPersistCounterBytes zeroed bytes (64 draw counters, 256 bytes apart); nWorkgroups = what the device holds at once
void launch_fast_tab_persist(hipStream_t st, const FrameSet &fs, const FastCellRec *dCells, int iniTh, int minTh, unsigned short *cellCnt, unsigned *slots,
                             int totalCells, long long totalSlots, int totalGroups, int smapRows, int nFrames, int winRows, int quadCap, bool iniFirst,
                             unsigned *stats, unsigned *counters, int nWorkgroups) {
    if (totalGroups <= 0) return;
    const unsigned recsPerFrame = (unsigned) totalGroups * 4u;
    const unsigned long long magic = ((1ull << 40) + recsPerFrame - 1) / recsPerFrame;
    const unsigned totalItems = recsPerFrame * (unsigned) nFrames;
    // whole frames per XCD where the frames divide (a frame's windows then meet in one L2), equal shares of the items otherwise
    const unsigned perXcd = nFrames >= 8 ? recsPerFrame * (unsigned) ((nFrames + 7) / 8) : (totalItems + 7u) / 8u;
    const size_t lds = fast_tab_lds_bytes(winRows, smapRows, quadCap);
    const dim3 grid((unsigned) nWorkgroups), block(256);
    if (iniFirst)
        hipLaunchKernelGGL((k_fast_tab_persist<true>), grid, block, lds, st, fs, dCells, iniTh, minTh, cellCnt, slots, totalCells, totalSlots, recsPerFrame, magic,
                           totalItems, perXcd, winRows, smapRows, quadCap, stats, counters);
    else
        hipLaunchKernelGGL((k_fast_tab_persist<false>), grid, block, lds, st, fs, dCells, iniTh, minTh, cellCnt, slots, totalCells, totalSlots, recsPerFrame, magic,
                           totalItems, perXcd, winRows, smapRows, quadCap, stats, counters);
}

size_t octree_lds_bytes(int maxCellsPerLevel, int cap, int ldsCand, bool globalNodes) {
    return sizeof(int) * ((size_t) maxCellsPerLevel + 1 + 3 + (globalNodes ? 0 : 19 * (size_t) cap) + 4 * (size_t) ldsCand);
}

size_t octree_hist_lds_bytes(int regionInts, int histBins) { return sizeof(int) * ((size_t) regionInts + (size_t) histBins + 1); }

hipError_t octree_prepare(size_t ldsBytes, bool globalNodes, bool hist) {
    constexpr int kOctMaxDyn = 160 * 1024 - 10 * 1024;   // the kernel's static LDS (radix histogram, scan scratch) is ~8.5 KB
    if (hist) return hipFuncSetAttribute((const void *) k_octree<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, kOctMaxDyn);
    return globalNodes ? hipFuncSetAttribute((const void *) k_octree<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, kOctMaxDyn)
                       : hipFuncSetAttribute((const void *) k_octree<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, kOctMaxDyn);
}

// levels [level0, level0 + nLaunchLevels) of every frame; histBins > 0 selects the histogram plan (regionInts ints shared by the cell table, the
// key tables and the node arrays, then histBins + 1 ints of prefix table)
void launch_octree(hipStream_t st, const LevelGeom *dGeom, int nlevels, int level0, int nLaunchLevels, const unsigned short *cellCnt, const unsigned *slots,
                   int totalCells, long long totalSlots, unsigned *k0, unsigned *v0, unsigned *k1, unsigned *v1, unsigned *xy,
                   long long candStride, unsigned *lvlKpXY, unsigned char *lvlKpScore, int *lvlKpCnt, int *lvlCandCnt,
                   uint2 *procRec, int kpStride, int cap, int ldsCand, size_t ldsBytes, int nFrames, long long *dbg, int *nodeArena,
                   int regionInts, int histBins, int helpers, int *gHist, int *gDone, int doneTarget, int spinBudget) {
    if (histBins > 0)
        hipLaunchKernelGGL((k_octree<false, true>), dim3(nLaunchLevels, nFrames, helpers > 1 && gHist ? helpers : 1), dim3(kOctBlock), ldsBytes, st, dGeom, nlevels, level0, cellCnt, slots, totalCells,
                           totalSlots, k0, v0, k1, v1, xy, candStride, lvlKpXY, lvlKpScore, lvlKpCnt, lvlCandCnt, procRec, kpStride, cap, 0,
                           dbg, nullptr, regionInts, histBins, helpers > 1 ? gHist : nullptr, gDone, doneTarget, spinBudget);
    else if (nodeArena)
        hipLaunchKernelGGL((k_octree<true, false>), dim3(nLaunchLevels, nFrames), dim3(kOctBlock), ldsBytes, st, dGeom, nlevels, level0, cellCnt, slots, totalCells,
                           totalSlots, k0, v0, k1, v1, xy, candStride, lvlKpXY, lvlKpScore, lvlKpCnt, lvlCandCnt, procRec, kpStride, cap, ldsCand,
                           dbg, nodeArena, 0, 0, nullptr, nullptr, 0, 0);
    else
        hipLaunchKernelGGL((k_octree<false, false>), dim3(nLaunchLevels, nFrames), dim3(kOctBlock), ldsBytes, st, dGeom, nlevels, level0, cellCnt, slots, totalCells,
                           totalSlots, k0, v0, k1, v1, xy, candStride, lvlKpXY, lvlKpScore, lvlKpCnt, lvlCandCnt, procRec, kpStride, cap, ldsCand,
                           dbg, nodeArena, 0, 0, nullptr, nullptr, 0, 0);
}

void launch_describe(hipStream_t st, const FrameSet &fs, const LevelGeom *dGeom, int nlevels, const int *lvlKpCnt, int *lvlBase,
                     const uint2 *procRec, int kpStride, ygzf_kp *outKp, uint8_t *outDesc, int *outCnt, int outStride, int nFrames, int cvMode) {
    const bool own = nFrames <= 4;
    if (!own) hipLaunchKernelGGL(k_level_bases, dim3((nFrames + 255) / 256), dim3(256), 0, st, lvlKpCnt, nlevels, nFrames, lvlBase, outCnt);
    const int nblk = (kpStride + kDescWaves - 1) / kDescWaves;
    const int blocksPerXcd = (nblk + 7) / 8;
    dim3 grid(8 * blocksPerXcd, nFrames);
#define YGZF_DESC_LAUNCH(M)                                                                                                                   \
    do {                                                                                                                                      \
        if (own) hipLaunchKernelGGL((k_describe<M, true>), grid, dim3(64 * kDescWaves), 0, st, fs, dGeom, lvlKpCnt, procRec, kpStride, outKp, outDesc, \
                                    outStride, blocksPerXcd, nlevels, outCnt);                                                                \
        else hipLaunchKernelGGL((k_describe<M, false>), grid, dim3(64 * kDescWaves), 0, st, fs, dGeom, (const int *) lvlBase, procRec, kpStride, outKp, \
                                outDesc, outStride, blocksPerXcd, nlevels, outCnt);                                                           \
    } while (0)
    switch (cvMode) {
        case YGZF_CV_LEGACY_INT: YGZF_DESC_LAUNCH(YGZF_CV_LEGACY_INT); break;
        case YGZF_CV_4: YGZF_DESC_LAUNCH(YGZF_CV_4); break;
        default: YGZF_DESC_LAUNCH(YGZF_CV_LEGACY_SSE2); break;
    }
#undef YGZF_DESC_LAUNCH
}

// The last frame's results of the previous launch become slot 0 of the output arrays (the "previous frame" of the batch matchers), or slot 0
// is emptied: one launch instead of three device-to-device copies issued by the host.
__global__ __launch_bounds__(256) void k_carry_slot(ygzf_kp *__restrict__ outKp, uint8_t *__restrict__ outDesc, int *__restrict__ outCnt, long long srcSlot,
                                                    int kpStride) {
    if (srcSlot <= 0) {
        if (blockIdx.x == 0 && threadIdx.x == 0) outCnt[0] = 0;
        return;
    }
    const int n = outCnt[srcSlot];
    const uint4 *sk = (const uint4 *) (outKp + srcSlot * kpStride);          // ygzf_kp: 28 bytes; kpStride entries = 7 * kpStride dwords
    const unsigned *skw = (const unsigned *) (outKp + srcSlot * kpStride);
    unsigned *dkw = (unsigned *) outKp;
    const uint4 *sd = (const uint4 *) (outDesc + srcSlot * kpStride * 32);
    uint4 *dd = (uint4 *) outDesc;
    const int nk = n * 7, ndv = n * 2;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < nk; i += gridDim.x * 256) dkw[i] = skw[i];
    for (int i = blockIdx.x * 256 + threadIdx.x; i < ndv; i += gridDim.x * 256) dd[i] = sd[i];
    if (blockIdx.x == 0 && threadIdx.x == 0) outCnt[0] = n;
}

// Results of the frames of a SMALL launch gathered into one contiguous block -- [counts | keypoint rows | descriptor rows], every part 256-byte
// aligned -- so that they cross the link as ONE copy: three device-to-host copies of a one-frame call cost 7 us each plus 8 us between them, this
// kernel 2 us (ygzf_batch_fetch_packed).  Dword copies; nKp / nDesc = dwords of the keypoint / descriptor rows of all frames.
__global__ __launch_bounds__(256) void k_pack_results(const unsigned *__restrict__ cnt, const unsigned *__restrict__ kp, const unsigned *__restrict__ desc, unsigned nCnt,
                                                      unsigned nKp, unsigned nDesc, unsigned *__restrict__ dst, unsigned offKp, unsigned offDesc) {
    const unsigned stride = gridDim.x * 256u;
    for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < nCnt; i += stride) dst[i] = cnt[i];
    for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < nKp; i += stride) dst[offKp + i] = kp[i];
    for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < nDesc; i += stride) dst[offDesc + i] = desc[i];
}
__global__ __launch_bounds__(256) void k_link_copy(uint4 *__restrict__ dst, const uint4 *__restrict__ src, unsigned n16) {
    for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < n16; i += gridDim.x * 256u) dst[i] = src[i];
}
void launch_link_copy(hipStream_t st, void *dst, const void *src, size_t bytes) {
    const unsigned n16 = (unsigned) (bytes / 16);
    if (!n16) return;
    hipLaunchKernelGGL(k_link_copy, dim3(std::min(64u, (n16 + 255u) / 256u)), dim3(256), 0, st, (uint4 *) dst, (const uint4 *) src, n16);
}
void launch_pack_results(hipStream_t st, const int *cnt, const ygzf_kp *kp, const uint8_t *desc, int nFrames, int kpStride, void *dst, size_t offKp, size_t offDesc) {
    const unsigned nKp = (unsigned) ((size_t) nFrames * kpStride * sizeof(ygzf_kp) / 4), nDesc = (unsigned) ((size_t) nFrames * kpStride * 8);
    const unsigned blocks = std::min(1024u, (nDesc + 255u) / 256u + 1u);
    hipLaunchKernelGGL(k_pack_results, dim3(blocks), dim3(256), 0, st, (const unsigned *) cnt, (const unsigned *) kp, (const unsigned *) desc, (unsigned) nFrames, nKp, nDesc,
                       (unsigned *) dst, (unsigned) (offKp / 4), (unsigned) (offDesc / 4));
}

void launch_carry_slot(hipStream_t st, ygzf_kp *outKp, uint8_t *outDesc, int *outCnt, long long srcSlot, int kpStride) {
    hipLaunchKernelGGL(k_carry_slot, dim3(srcSlot > 0 ? 16 : 1), dim3(256), 0, st, outKp, outDesc, outCnt, srcSlot, kpStride);
}

void launch_describe_list(hipStream_t st, const FrameSet &fs, const LevelGeom *dGeom, const void *list, int n, int frame,
                          float *outAngle, uint8_t *outDesc, int cvMode) {
    if (n <= 0) return;
    const dim3 grid((n + kDescWaves - 1) / kDescWaves), block(64 * kDescWaves);
    switch (cvMode) {
        case YGZF_CV_LEGACY_INT: hipLaunchKernelGGL(k_describe_list<YGZF_CV_LEGACY_INT>, grid, block, 0, st, fs, dGeom, (const int4 *) list, n, frame, outAngle, outDesc); break;
        case YGZF_CV_4: hipLaunchKernelGGL(k_describe_list<YGZF_CV_4>, grid, block, 0, st, fs, dGeom, (const int4 *) list, n, frame, outAngle, outDesc); break;
        default: hipLaunchKernelGGL(k_describe_list<YGZF_CV_LEGACY_SSE2>, grid, block, 0, st, fs, dGeom, (const int4 *) list, n, frame, outAngle, outDesc); break;
    }
}

void launch_hamming_pairs(hipStream_t st, const void *a, const void *b, int n, int *out) {
    hipLaunchKernelGGL(k_hamming_pairs, dim3((n + 255) / 256), dim3(256), 0, st, (const unsigned long long *) a,
                       (const unsigned long long *) b, n, out);
}

}  // namespace ygzf
