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
#include "wave_ops.h"

namespace ygzf {

constexpr int GRID_COLS = 64, GRID_ROWS = 48, GRID_CELLS = GRID_COLS * GRID_ROWS;
constexpr int TH_HIGH = 100;
constexpr int HISTO_LENGTH = 30;

__device__ __forceinline__ int m_lane() { return threadIdx.x & 63; }

__device__ __forceinline__ int m_wave_incl_scan(int v) { return wave_incl_scan(v); }

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

struct QueryParam {  // per Last keypoint, precomputed by all waves (32 bytes)
    float u, v, radius, ur, angle;    // ur: expected right-image column (u - mbf/z, or MapPoint::mTrackProjXR)
    unsigned char minCx, maxCx, minCy, maxCy;
    signed char minLevel, maxLevel;   // GetFeaturesInArea arguments (-1 = unbounded)
    unsigned char valid, hasObs;
    unsigned pad;
};

constexpr int kMatchBlock = 1024;
constexpr unsigned kNoKey = (256u << 16);
constexpr int kExtSlots = 64;

// The Cur descriptors live either in LDS or in global memory (MatchArgs::descInLds).  Read through generic pointers the compiler folds the two
// branches into ONE flat load with a selected address -- and a flat load that lands in LDS costs ~1 us per candidate at this kernel's occupancy
// (measured: 13 candidates per lane = 12.5 us).  The LDS side is therefore read through an LDS-typed pointer (ds_read_b128).
typedef unsigned v4u_t __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) const v4u_t lds_uint4_t;
__device__ __forceinline__ void lds_desc(const unsigned long long *base, int i2, unsigned long long &d0, unsigned long long &d1, unsigned long long &d2,
                                         unsigned long long &d3) {
    lds_uint4_t *p = (lds_uint4_t *) (base + 4 * (size_t) i2);
    const v4u_t a = p[0], b = p[1];
    d0 = ((unsigned long long) a.y << 32) | a.x; d1 = ((unsigned long long) a.w << 32) | a.z;
    d2 = ((unsigned long long) b.y << 32) | b.x; d3 = ((unsigned long long) b.w << 32) | b.z;
}

struct MatchLds {
    int *cellStart, *cellFill, *list, *events;
    QueryParam *qp;
    unsigned long long *desc;   // Cur descriptors (optional)
    float *cx, *cy, *cang;      // Cur keypoint x, y, angle
    uint4 *specKey;             // per query: the 4 best acceptable candidates against the INITIAL ownership,
    ushort4 *specI2;            //   keys (dist << 16 | visiting order) ascending; key >= kNoKey: none
    uint4 *specKeyB;            // entries 5..8 of the same list when A.specDeep (SearchByProjection(F, MapPoints) needs best AND
    ushort4 *specI2B;           //   runner-up among the still-free candidates: four entries run out on 15 % of the queries)
    float *qang;                // Last keypoint angle
    unsigned char *qobs;        // MapPoint has observations
    unsigned char *owner, *octave;
    int *match;                 // per Cur keypoint: accepted Last index (flushed to global at the end)
    int *claim;                 // per Cur keypoint: lowest pending lane that wants it this round (64 = none)
    // fixpoint pass: list extensions.  extOf[2 q], extOf[2 q + 1]: slots of query q's first / second extension (0xFF: none)
    unsigned char *extOf;
    unsigned *extKey;           // kExtSlots x 8 keys
    unsigned short *extI2;      // kExtSlots x 8 Cur indices
    int *extWork;               // queries waiting for an extension this round
};

// GetFeaturesInArea + best-candidate scan of one query by one wave.
// The reference visits cells `for ix: for iy:` (src/Frame.cc:451-452); cells of one grid column are adjacent in the
// counting-sorted list (cell id = ix*48 + iy), so the candidates of a query are (maxCx-minCx+1) <= 64 contiguous list
// ranges, already in the reference's visiting order.  The ranges are flattened: lane j takes the j-th candidate, so the
// dependent LDS chain (index -> attributes -> descriptor) is walked once per 64 candidates instead of once per cell.
// Returns the wave-uniform best key ((dist << 16) | order; dist 256 = none) and the Cur index that holds it.
// Wave-wide unsigned min without LDS traffic: DPP inside each row of 16 lanes (quad_perm swaps, row_ror 4/8), then the
// four row results are combined through SGPRs (v_readlane).  ds_bpermute-based shuffles cost an LDS round trip per step,
// which dominated the scan when it was written with __shfl_xor.
__device__ __forceinline__ unsigned wave_min_dpp(unsigned v) {
    unsigned t;
    t = (unsigned) __builtin_amdgcn_update_dpp((int) v, (int) v, 0xb1, 0xf, 0xf, false); v = t < v ? t : v;   // quad_perm [1,0,3,2]
    t = (unsigned) __builtin_amdgcn_update_dpp((int) v, (int) v, 0x4e, 0xf, 0xf, false); v = t < v ? t : v;   // quad_perm [2,3,0,1]
    t = (unsigned) __builtin_amdgcn_update_dpp((int) v, (int) v, 0x124, 0xf, 0xf, false); v = t < v ? t : v;  // row_ror:4
    t = (unsigned) __builtin_amdgcn_update_dpp((int) v, (int) v, 0x128, 0xf, 0xf, false); v = t < v ? t : v;  // row_ror:8
    const unsigned a = (unsigned) __builtin_amdgcn_readlane((int) v, 0), b = (unsigned) __builtin_amdgcn_readlane((int) v, 16);
    const unsigned c = (unsigned) __builtin_amdgcn_readlane((int) v, 32), d = (unsigned) __builtin_amdgcn_readlane((int) v, 48);
    const unsigned ab = a < b ? a : b, cd = c < d ? c : d;
    return ab < cd ? ab : cd;
}

// Device-scope accesses for what one workgroup of a pair hands to another (several workgroups per pair, MatchArgs::split): written through /
// read past the XCD's own L2 per access.  The alternative -- plain stores and a __threadfence() on either side -- writes back and invalidates
// the WHOLE L2 of the XCD, dirty lines of the kernels before included: 6 of the 9 us of the hand-over.
//
// What makes the fence-free hand-over correct is a property of THIS target, not of the HIP memory model (under which relaxed accesses order
// nothing): on gfx942 / gfx950 (LLVM AMDGPU memory model, "gfx942" code sequences) a monotonic agent-scope atomic store IS `global_store ... sc1`
// (written through the XCD's L2 to the device-coherent level), a monotonic agent-scope atomic load IS `global_load ... sc1` (served from that
// level, never from a stale line of this XCD's L2), and gfx9 counts stores in vmcnt, so `s_waitcnt vmcnt(0)` after the stores means every one of them
// has been acknowledged at that level before the counter moves; the winner's sc1 loads, issued after its agent-scope atomicAdd returned, then
// read exactly those bytes.  Both halves are checked instead of assumed: (1) this translation unit refuses to take the fence-free path on any
// other target (kHandoverScopedAccess below: targets that count stores separately, vscnt on gfx10+, or lower these builtins differently, get the
// fenced path unconditionally); (2) the build (orb_ygz_slam_amd/build.py: check_handover_isa) disassembles this file and fails unless
// k_handover_probe -- the same two inline functions, same flags -- is exactly one sc1 load and one sc1 store and k_match_last carries the
// scoped accesses of its hand-over; (3) tests/test_gpu_handover.py runs the split path hundreds of times under load against the serial pass.
#if (defined(__gfx950__) || defined(__gfx942__)) && !defined(YGZF_FORCE_HANDOVER_FENCE)   // (the macro: build.py, when its ISA check fails)
constexpr bool kHandoverScopedAccess = true;
#else
constexpr bool kHandoverScopedAccess = false;   // host pass of the compilation, or a target the argument above was not made for
#endif
__device__ __forceinline__ void st_dev(void *p, unsigned v) { __hip_atomic_store((unsigned *) p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned ld_dev(const void *p) { return __hip_atomic_load((const unsigned *) p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// (build-time probe, never launched: see above)
__global__ void k_handover_probe(unsigned *dst, const unsigned *src) { st_dev(dst, ld_dev(src)); }
__device__ __forceinline__ void st_dev4(void *p, unsigned a, unsigned b, unsigned c, unsigned d) {
    unsigned *q = (unsigned *) p;
    st_dev(q, a); st_dev(q + 1, b); st_dev(q + 2, c); st_dev(q + 3, d);
}
__device__ __forceinline__ uint4 ld_dev4(const void *p) {
    const unsigned *q = (const unsigned *) p;
    return make_uint4(ld_dev(q), ld_dev(q + 1), ld_dev(q + 2), ld_dev(q + 3));
}
__device__ __forceinline__ void store_qp(QueryParam *dst, const QueryParam &q) {
    unsigned w[8];
    __builtin_memcpy(w, &q, 32);
    st_dev4(dst, w[0], w[1], w[2], w[3]);
    st_dev4((unsigned *) dst + 4, w[4], w[5], w[6], w[7]);
}
__device__ __forceinline__ QueryParam load_qp(const QueryParam *src) {
    const uint4 a = ld_dev4(src), b = ld_dev4((const unsigned *) src + 4);
    const unsigned w[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    QueryParam q;
    __builtin_memcpy(&q, w, 32);
    return q;
}
static_assert(sizeof(QueryParam) == 32, "QueryParam is handed over as eight dwords");

template <class Visit>
__device__ __forceinline__ void for_each_candidate(const MatchArgs &A, const MatchLds &L, const QueryParam &q, unsigned long long q0,
                                                   unsigned long long q1, unsigned long long q2, unsigned long long q3,
                                                   const uint8_t *curDesc, const float *uRight, int lane, const volatile int *matchedDist,
                                                   const int *claimant, int self, Visit visit) {
    const int nCx = q.maxCx - q.minCx + 1;   // <= 64 (one lane per grid column)
    const bool bCheckLevels = (q.minLevel > 0) || (q.maxLevel >= 0);
    int rs = 0, rlen = 0;
    if (lane < nCx) {
        const int c0 = (q.minCx + lane) * GRID_ROWS;
        rs = L.cellStart[c0 + q.minCy];
        rlen = L.cellStart[c0 + q.maxCy + 1] - rs;
    }
    // lane j -> list index of the j-th candidate; the (<= 64) ranges are walked with wave-uniform scalars
    int li0 = -1, li1 = -1, li2 = -1, li3 = -1;   // candidates lane, lane+64, lane+128, lane+192
    int total = 0;
    for (int r = 0; r < nCx; r++) {
        const int st = __builtin_amdgcn_readlane(rs, r), ln = __builtin_amdgcn_readlane(rlen, r);
        const int a0 = lane - total, a1 = a0 + 64, a2 = a0 + 128, a3 = a0 + 192;
        if (a0 >= 0 && a0 < ln) li0 = st + a0;
        if (a1 >= 0 && a1 < ln) li1 = st + a1;
        if (a2 >= 0 && a2 < ln) li2 = st + a2;
        if (a3 >= 0 && a3 < ln) li3 = st + a3;
        total += ln;
    }
    for (int jb = 0; jb < total; jb += 64) {
        const int j = jb + lane;
        int li;
        if (jb == 0) li = li0;
        else if (jb == 64) li = li1;
        else if (jb == 128) li = li2;
        else if (jb == 192) li = li3;
        else {   // more than 256 candidates in one window: rare, resolve by walking the ranges again
            li = -1;
            int acc = 0;
            for (int r = 0; r < nCx; r++) {
                const int st = __builtin_amdgcn_readlane(rs, r), ln = __builtin_amdgcn_readlane(rlen, r);
                if (j >= acc && j < acc + ln) li = st + (j - acc);
                acc += ln;
            }
        }
        if (li < 0) continue;
        const int i2 = L.list[li];
        if (bCheckLevels) {
            const int o = L.octave[i2];
            if (o < q.minLevel) continue;
            if (q.maxLevel >= 0 && o > q.maxLevel) continue;
        }
        const float distx = L.cx[i2] - q.u, disty = L.cy[i2] - q.v;
        if (!(fabsf(distx) < q.radius && fabsf(disty) < q.radius)) continue;
        if (L.owner[i2] == 2) continue;  // mvpMapPoints[i2] && Observations() > 0
        if (claimant && claimant[i2] < self) continue;   // taken by an earlier query whose MapPoint has observations
        if (uRight && uRight[i2] > 0) {
            const float er = fabsf(q.ur - uRight[i2]);
            if (er > q.radius) continue;
        }
        unsigned long long d0, d1, d2, d3;
        if (A.descInLds) {
            lds_desc(L.desc, i2, d0, d1, d2, d3);
        } else {
            const unsigned long long *d = (const unsigned long long *) (curDesc + (size_t) i2 * 32);
            d0 = d[0]; d1 = d[1]; d2 = d[2]; d3 = d[3];
        }
        const unsigned dist = __popcll(q0 ^ d0) + __popcll(q1 ^ d1) + __popcll(q2 ^ d2) + __popcll(q3 ^ d3);
        if (matchedDist && matchedDist[i2] <= (int) dist) continue;   // SearchForInitialization :414-415
        visit((dist << 16) | (unsigned) j, i2);
    }
}

__device__ __forceinline__ unsigned scan_query(const MatchArgs &A, const MatchLds &L, const QueryParam &q, unsigned long long q0,
                                               unsigned long long q1, unsigned long long q2, unsigned long long q3,
                                               const uint8_t *curDesc, const float *uRight, int lane, int *bestIdx2,
                                               unsigned *secondKey = nullptr, int *secondIdx2 = nullptr,
                                               const volatile int *matchedDist = nullptr) {
    unsigned best = (256u << 16) | 0xFFFFu, best2 = (256u << 16) | 0xFFFFu;   // per-lane best and runner-up
    int bestI2 = -1, best2I2 = -1;
    for_each_candidate(A, L, q, q0, q1, q2, q3, curDesc, uRight, lane, matchedDist, nullptr, 0, [&](unsigned key, int i2) {
        if (key < best) { best2 = best; best2I2 = bestI2; best = key; bestI2 = i2; }
        else if (key < best2) { best2 = key; best2I2 = i2; }
    });
    const unsigned wbest = wave_min_dpp(best);
    const unsigned long long who = __ballot(best == wbest);
    const int src = __ffsll((long long) who) - 1;
    *bestIdx2 = __builtin_amdgcn_readlane(bestI2, src);
    if (secondKey) {   // second smallest (dist, order) key over the same candidates (SearchByProjection(F, MapPoints) ratio test)
        const unsigned second = wave_min_dpp(lane == src ? best2 : min(best, best2));
        const unsigned long long who2 = __ballot(lane == src ? best2 == second : (best == second || best2 == second));
        const int src2 = __ffsll((long long) who2) - 1;
        const int cand2 = (lane == src) ? best2I2 : (best == second ? bestI2 : best2I2);
        *secondKey = second;
        *secondIdx2 = __builtin_amdgcn_readlane(cand2, src2);
    }
    return wbest;
}

// The next eight entries of a query's (dist, order)-sorted candidate list after `afterKey`, by one wave: what the speculative scan would have
// put into positions 9..16 (17..24) had its lists been longer.  Independent of who has taken what since (only the INITIAL ownership excludes,
// as in the speculative scan), so any number of queries can be extended at any time.  maxDist < 256: candidates beyond it do not count (modes
// 0 / 2: the list simply ends).  outK / outJ are wave-uniform; missing entries are kNoKey-or-larger.
__device__ __forceinline__ void scan_after(const MatchArgs &A, const MatchLds &L, const QueryParam &q, unsigned long long q0, unsigned long long q1,
                                           unsigned long long q2, unsigned long long q3, const uint8_t *curDesc, const float *uRight, int lane,
                                           unsigned afterKey, unsigned maxDist, unsigned (&outK)[8], unsigned (&outJ)[8]) {
    unsigned K[8], J[8];
#pragma unroll
    for (int e = 0; e < 8; e++) { K[e] = 0xFFFFFFFFu; J[e] = 0; }
    for_each_candidate(A, L, q, q0, q1, q2, q3, curDesc, uRight, lane, nullptr, nullptr, 0, [&](unsigned key, int i2) {
        if (key <= afterKey || (key >> 16) > maxDist || !(key < K[7])) return;
        unsigned ck = key, cj = (unsigned) i2;
#pragma unroll
        for (int e = 0; e < 8; e++)
            if (ck < K[e]) { const unsigned tk = K[e], tj = J[e]; K[e] = ck; J[e] = cj; ck = tk; cj = tj; }
    });
#pragma unroll
    for (int r = 0; r < 8; r++) {   // the wave's r-th smallest = the least of the lanes' heads; its lane pops
        const unsigned m = wave_min_dpp(K[0]);
        const unsigned long long who = __ballot(K[0] == m);
        const int src = __ffsll((long long) who) - 1;
        outK[r] = m;
        outJ[r] = (unsigned) __builtin_amdgcn_readlane((int) J[0], src);
        if (lane == src && m != 0xFFFFFFFFu) {
#pragma unroll
            for (int e = 0; e < 7; e++) { K[e] = K[e + 1]; J[e] = J[e + 1]; }
            K[7] = 0xFFFFFFFFu;
        }
    }
}

// Two lanes' ascending lists of eight (key, index) folded into the eight smallest of both, ascending, in BOTH lanes: the element-wise
// minimum against the partner's reversed list is the lower half of a bitonic merge and itself bitonic; three compare-exchange stages
// sort it.  Keys are unique (they carry the visiting order) except for the 0xFFFFFFFF of empty entries.
template <int kDppCtrl>
__device__ __forceinline__ void spec_merge8(unsigned (&K)[8], unsigned (&J)[8]) {
    unsigned pk[8], pj[8];
#pragma unroll
    for (int e = 0; e < 8; e++) {
        pk[e] = (unsigned) __builtin_amdgcn_update_dpp(0, (int) K[7 - e], kDppCtrl, 0xf, 0xf, false);
        pj[e] = (unsigned) __builtin_amdgcn_update_dpp(0, (int) J[7 - e], kDppCtrl, 0xf, 0xf, false);
    }
#pragma unroll
    for (int e = 0; e < 8; e++)
        if (pk[e] < K[e]) { K[e] = pk[e]; J[e] = pj[e]; }
#pragma unroll
    for (int st = 4; st >= 1; st >>= 1)
#pragma unroll
        for (int e = 0; e < 8; e++)
            if ((e & st) == 0 && K[e + st] < K[e]) {
                const unsigned tk = K[e], tj = J[e];
                K[e] = K[e + st]; J[e] = J[e + st];
                K[e + st] = tk; J[e + st] = tj;
            }
}

// The same fold with the partner lane ^ kXor (16, 32: across DPP rows), through the LDS crossbar.
template <int kXor>
__device__ __forceinline__ void spec_merge8_xor(unsigned (&K)[8], unsigned (&J)[8], int lane) {
    const int addr = (lane ^ kXor) << 2;
    unsigned pk[8], pj[8];
#pragma unroll
    for (int e = 0; e < 8; e++) {
        pk[e] = (unsigned) __builtin_amdgcn_ds_bpermute(addr, (int) K[7 - e]);
        pj[e] = (unsigned) __builtin_amdgcn_ds_bpermute(addr, (int) J[7 - e]);
    }
#pragma unroll
    for (int e = 0; e < 8; e++)
        if (pk[e] < K[e]) { K[e] = pk[e]; J[e] = pj[e]; }
#pragma unroll
    for (int st = 4; st >= 1; st >>= 1)
#pragma unroll
        for (int e = 0; e < 8; e++)
            if ((e & st) == 0 && K[e + st] < K[e]) {
                const unsigned tk = K[e], tj = J[e];
                K[e] = K[e + st]; J[e] = J[e + st];
                K[e + st] = tk; J[e + st] = tj;
            }
}

__global__ __launch_bounds__(kMatchBlock) void k_match_last(MatchArgs A) {
    extern __shared__ __attribute__((aligned(16))) unsigned char dyn[];
    __shared__ int s_tmp[20];
    __shared__ int s_hist[HISTO_LENGTH];
    const int tid = threadIdx.x, lane = m_lane(), wave = tid >> 6;
    const int S = A.split > 1 ? A.split : 1;
    const int pair = S > 1 ? (int) blockIdx.x / S : (int) blockIdx.x, part = (int) blockIdx.x - pair * S;
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

    long long *dbg = A.dbg ? A.dbg + (long long) pair * 8 : nullptr;
#define STAMP(k) do { if (dbg && tid == 0) dbg[k] = wall_clock64(); } while (0)
    const long long tStart = dbg ? wall_clock64() : 0;   // (with several workgroups per pair only the one that survives the hand-over reports)
    // ---- LDS carve-up ----
    MatchLds L;
    // Arrays that do not fit the LDS budget (large keypoint counts: 4000 / 8000 features) live in a per-pair global
    // scratch arena instead; `spill` is the host's plan, bit per array group.
    unsigned char *p = dyn;
    unsigned char *gp = (unsigned char *) A.spillScratch + (long long) pair * A.spillStride;
#define CARVE(ptr, type, count, bit)                                                        \
    do {                                                                                    \
        if (A.spill & (bit)) { ptr = (type *) gp; gp += (((size_t) sizeof(type) * (count)) + 15) & ~(size_t) 15; } \
        else { ptr = (type *) p; p += (((size_t) sizeof(type) * (count)) + 15) & ~(size_t) 15; }                   \
    } while (0)
    CARVE(L.cellStart, int, GRID_CELLS + 1, 0);
    CARVE(L.cellFill, int, GRID_CELLS, 0);
    CARVE(L.list, int, A.capCur, 0);
    CARVE(L.cx, float, A.capCur, 0);
    CARVE(L.cy, float, A.capCur, 0);
    CARVE(L.owner, unsigned char, A.capCur, 0);
    CARVE(L.octave, unsigned char, A.capCur, 0);
    CARVE(L.qobs, unsigned char, A.capLast, 0);
    CARVE(L.specKey, uint4, A.capLast, kSpillSpec);
    CARVE(L.specI2, ushort4, A.capLast, kSpillSpec);
    L.specKeyB = nullptr;
    L.specI2B = nullptr;
    if (A.specDeep) {
        CARVE(L.specKeyB, uint4, A.capLast, kSpillSpec);
        CARVE(L.specI2B, ushort4, A.capLast, kSpillSpec);
    }
    CARVE(L.events, int, A.capLast, kSpillMisc);
    CARVE(L.qang, float, A.capLast, kSpillMisc);
    CARVE(L.cang, float, A.capCur, kSpillMisc);
    CARVE(L.match, int, A.capCur, kSpillMisc);
    CARVE(L.claim, int, A.capCur, kSpillMisc);
    CARVE(L.extOf, unsigned char, 2 * (size_t) A.capLast, kSpillMisc);
    CARVE(L.extKey, unsigned, 8 * kExtSlots, kSpillMisc);
    CARVE(L.extI2, unsigned short, 8 * kExtSlots, kSpillMisc);
    CARVE(L.extWork, int, kExtSlots, kSpillMisc);
    L.desc = nullptr;
    if (A.descInLds) CARVE(L.desc, unsigned long long, 4 * (size_t) A.capCur, 0);
#undef CARVE
    L.qp = (QueryParam *) A.qpScratch + (long long) pair * A.capLast;   // global: only the rare full rescans read it back

    // ---- Frame::AssignFeaturesToGrid ----
    for (int i = tid; i < GRID_CELLS; i += kMatchBlock) L.cellFill[i] = 0;
    if (tid < HISTO_LENGTH) s_hist[tid] = 0;
    __syncthreads();
    for (int i = tid; i < nt; i += kMatchBlock) {
        const ygzf_kp k = curKeys[i];
        const int px = (int) roundf((k.x - A.minX) * A.gridInvW);
        const int py = (int) roundf((k.y - A.minY) * A.gridInvH);
        if (!(px < 0 || px >= GRID_COLS || py < 0 || py >= GRID_ROWS)) atomicAdd(&L.cellFill[px * GRID_ROWS + py], 1);
        L.owner[i] = A.ownerIn ? A.ownerIn[(long long) pair * A.kpStrideCur + i] : 0;
        L.octave[i] = (unsigned char) k.octave;
        L.cx[i] = k.x; L.cy[i] = k.y; L.cang[i] = k.angle;
        L.match[i] = -1;
        L.claim[i] = 64;
        if (A.descInLds) {
            const unsigned long long *d = (const unsigned long long *) (curDesc + (size_t) i * 32);
            L.desc[4 * i] = d[0]; L.desc[4 * i + 1] = d[1]; L.desc[4 * i + 2] = d[2]; L.desc[4 * i + 3] = d[3];
        }
    }
    __syncthreads();
    {   // exclusive scan of 3072 counts: 3 per thread
        const int per = GRID_CELLS / kMatchBlock;
        int s = 0;
        for (int k = 0; k < per; k++) s += L.cellFill[tid * per + k];
        int incl = m_wave_incl_scan(s);
        if (lane == 63) s_tmp[wave] = incl;
        __syncthreads();
        int woff = 0;
        for (int w2 = 0; w2 < wave; w2++) woff += s_tmp[w2];
        int off = woff + incl - s;
        for (int k = 0; k < per; k++) {
            const int c = L.cellFill[tid * per + k];
            L.cellStart[tid * per + k] = off;
            off += c;
        }
        if (tid == kMatchBlock - 1) L.cellStart[GRID_CELLS] = off;
    }
    __syncthreads();
    for (int i = tid; i < GRID_CELLS; i += kMatchBlock) L.cellFill[i] = L.cellStart[i];
    __syncthreads();
    for (int i = tid; i < nt; i += kMatchBlock) {
        const int px = (int) roundf((L.cx[i] - A.minX) * A.gridInvW);
        const int py = (int) roundf((L.cy[i] - A.minY) * A.gridInvH);
        if (!(px < 0 || px >= GRID_COLS || py < 0 || py >= GRID_ROWS)) L.list[atomicAdd(&L.cellFill[px * GRID_ROWS + py], 1)] = i;
    }
    __syncthreads();
    for (int c = tid; c < GRID_CELLS; c += kMatchBlock) {  // cells keep ascending keypoint index (push_back order)
        const int s = L.cellStart[c], e = L.cellStart[c + 1];
        for (int a = s + 1; a < e; a++) {
            const int v = L.list[a];
            int b = a - 1;
            while (b >= s && L.list[b] > v) { L.list[b + 1] = L.list[b]; b--; }
            L.list[b + 1] = v;
        }
    }
    __syncthreads();   // the candidate scans below read the cells other threads have just sorted
    const long long tGrid = dbg ? wall_clock64() : 0;
    // ---- per-query projection (:1243-1272) ----
    const float *Rcw = pose, *tcw = pose + 9, *Rlw = pose + 12, *tlw = pose + 21;
    float twc[3], tlc2;
    for (int i = 0; i < 3; i++) twc[i] = -1 * (Rcw[i] * tcw[0] + Rcw[3 + i] * tcw[1] + Rcw[6 + i] * tcw[2]);
    tlc2 = (Rlw[6] * twc[0] + Rlw[7] * twc[1] + Rlw[8] * twc[2]) + tlw[2];
    const bool bForward = tlc2 > A.mb && !A.bMono;
    const bool bBackward = -tlc2 > A.mb && !A.bMono;
    // split mode: the queries of this part, eight to a wave (a wave waits for its slowest lane: few queries per wave, all SIMDs busy)
    // and EIGHT LANES to a query: lane qr of the group scans every eighth entry of the query's candidate ranges and the group merges its
    // sorted lists (the serial scan of the widest window was the phase's duration: ~100 ns per candidate, hundreds of candidates)
    const int nLocal = (nq - part + S - 1) / S;
    int LPQ = S > 1 ? 8 : 1, qr = S > 1 ? (tid & 7) : 0;
    int jFirst = S > 1 ? (tid >> 3) : tid, jStep = S > 1 ? (kMatchBlock >> 3) : kMatchBlock, jEnd = nLocal;
    if (nLocal <= kMatchBlock && !A.fixedLanes) {
        // Lanes by expected work (one workgroup per pair, the form of large batches, included: 256 pairs 161 -> 149 us).  A query's window grows with the square of its level's scale factor (radius = th * scale), and the queries
        // arrive sorted by level, so with eight lanes for everybody the wave that holds the coarsest level's queries ran ten times longer than
        // the first while the others idled.  Each of the nUse waves that take part gets a contiguous run of queries of equal total weight
        // scale^2 and spreads its 64 lanes over them (2 lanes per level-0 query, 32 per level-7 query when a workgroup holds a hundred of them).
        const int nUse = min(kMatchBlock / 64, max(4, (nLocal * 8 + 63) / 64));
        int jb, je;
        if (nLocal <= 32) {                   // a handful of queries per workgroup (one pair over 64 workgroups): equal counts, no weighing
            jb = nLocal * min(wave, nUse) / nUse;
            je = nLocal * min(wave + 1, nUse) / nUse;
        } else {
        int *wsum = L.events;                 // free until the in-order phase
        int wgt = 0;
        if (tid < nLocal) {
            const int i = tid * S + part;
            int lvl = (A.mode == 0 || A.mode == 3) ? lastKeys[i].octave : A.mpLevel[(long long) pair * A.kpStrideLast + i];
            lvl = min(max(lvl, 0), (int) kMaxLevels - 1);
            const float sf = A.scaleFactors[lvl];
            wgt = max(1, (int) (sf * sf * 16.f));
        }
        const int incl = m_wave_incl_scan(wgt);
        if (lane == 63) s_tmp[wave] = incl;
        __syncthreads();
        int woff = 0;
        for (int w2 = 0; w2 < wave; w2++) woff += s_tmp[w2];
        if (tid < nLocal) wsum[tid] = woff + incl;
        __syncthreads();
        const int W = wsum[nLocal - 1];
        // ... over as many waves as give a query eight lanes on average, at least one per SIMD, not always over all sixteen: a wave's prologue,
        // folds and stores are ~1300 instructions whatever its lanes do, and the SIMDs issue them one at a time
        int bound = 0;                        // lane k <= nUse: first query whose inclusive weight exceeds k parts of the total
        if (lane <= nUse) {
            const long long target = (long long) W * lane / nUse;
            int lo = 0, hi = nLocal;
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if ((long long) wsum[mid] <= target) lo = mid + 1;
                else hi = mid;
            }
            bound = lane == nUse ? nLocal : lo;
        }
        jb = __shfl(bound, min(wave, nUse));
        je = __shfl(bound, min(wave + 1, nUse));
        __syncthreads();                      // wsum (= L.events) is free again
        }
        const int cnt = je - jb;
        int g = 1;
        while (g < cnt && g < 64) g <<= 1;
        LPQ = 64 / g;
        qr = lane & (LPQ - 1);
        jFirst = jb + lane / LPQ;
        jStep = g;
        jEnd = je;
    }
    for (int j = jFirst; j < jEnd; j += jStep) {
        const int i = j * S + part;
        QueryParam q;
        q.valid = 0;
        q.u = q.v = q.radius = q.ur = q.angle = 0;
        q.minCx = q.maxCx = q.minCy = q.maxCy = 0;
        q.minLevel = q.maxLevel = -1;
        q.hasObs = hasObs ? (hasObs[i] != 0) : 1;
        q.pad = 0;
        // everything this query reads from global memory is requested up front, together (flags -> position -> keypoint -> descriptor were
        // dependent round trips)
        const unsigned long long *qdp = (const unsigned long long *) (mpDesc + (size_t) i * 32);
        const unsigned long long q0 = qdp[0], q1 = qdp[1], q2 = qdp[2], q3 = qdp[3];
        ygzf_kp lk0;
        lk0.x = lk0.y = lk0.angle = 0; lk0.octave = 0;
        if (A.mode == 0 || A.mode == 3) lk0 = lastKeys[i];
        float X0 = 0, X1 = 0, X2 = 0;
        if (A.mode == 0 && !A.unitWorld) { X0 = world[3 * (size_t) i]; X1 = world[3 * (size_t) i + 1]; X2 = world[3 * (size_t) i + 2]; }
        float pX = 0, pY = 0, pXR = 0, pVC = 0, pAng = 0;
        int pLvl = 0;
        if (A.mode != 0) {
            const long long o = (long long) pair * A.kpStrideLast + i;
            pX = A.mpProjX[o]; pY = A.mpProjY[o];
            if (A.mode != 3) {
                pLvl = A.mpLevel[o];
                if (A.mpProjXR) pXR = A.mpProjXR[o];
                if (A.mode == 2) pAng = A.mpAngle[o];
                else pVC = A.mpViewCos[o];
            }
        }
        const bool has = (mpValid ? mpValid[i] != 0 : true) && !(outlier ? outlier[i] != 0 : false);
        if (A.mode == 3) {
            // SearchForInitialization(F1, F2, vbPrevMatched, vnMatches12, windowSize)  src/ORBmatcher.cc:375-478: queries = the level-0
            // keys of F1 (:390-392), window windowSize around the previously matched position, level 0 only
            const ygzf_kp lk = lk0;
            if (lk.octave <= 0) {
                const float u = pX, v = pY;
                const float rad = A.th;
                const int nMinCellX = max(0, (int) floorf((u - A.minX - rad) * A.gridInvW));
                const int nMaxCellX = min(GRID_COLS - 1, (int) ceilf((u - A.minX + rad) * A.gridInvW));
                const int nMinCellY = max(0, (int) floorf((v - A.minY - rad) * A.gridInvH));
                const int nMaxCellY = min(GRID_ROWS - 1, (int) ceilf((v - A.minY + rad) * A.gridInvH));
                if (!(nMinCellX >= GRID_COLS || nMaxCellX < 0 || nMinCellY >= GRID_ROWS || nMaxCellY < 0) && nMaxCellX >= nMinCellX &&
                    nMaxCellY >= nMinCellY) {
                    q.valid = 1;
                    q.u = u; q.v = v; q.radius = rad; q.angle = lk.angle;
                    q.minCx = (unsigned char) nMinCellX; q.maxCx = (unsigned char) nMaxCellX;
                    q.minCy = (unsigned char) nMinCellY; q.maxCy = (unsigned char) nMaxCellY;
                    q.minLevel = (signed char) lk.octave; q.maxLevel = (signed char) lk.octave;
                }
            }
        } else if (A.mode != 0) {
            // mode 1: SearchByProjection(Frame &F, const vector<MapPoint*> &, th, checkLevel)  src/ORBmatcher.cc:43-126: the projection was
            // done by Frame::isInFrustum; mpValid = mbTrackInView, outlier = isBad()
            // mode 2: SearchByProjection(Cur, KeyFrame, found, th, ORBdist)  :1352-1469: projection / distance gate / PredictScale on the host
            if (has) {
                const int lvl = pLvl;
                float r;
                if (A.mode == 2) r = A.th;                                                             // radius = th * scale[nPredictedLevel]
                else {
                    r = pVC > 0.998 ? 2.5f : 4.0f;                                                     // RadiusByViewingCos
                    if (A.th != 1.0) r *= A.th;
                }
                const float u = pX, v = pY;
                const float rad = r * A.scaleFactors[lvl];
                const int nMinCellX = max(0, (int) floorf((u - A.minX - rad) * A.gridInvW));
                const int nMaxCellX = min(GRID_COLS - 1, (int) ceilf((u - A.minX + rad) * A.gridInvW));
                const int nMinCellY = max(0, (int) floorf((v - A.minY - rad) * A.gridInvH));
                const int nMaxCellY = min(GRID_ROWS - 1, (int) ceilf((v - A.minY + rad) * A.gridInvH));
                if (!(nMinCellX >= GRID_COLS || nMaxCellX < 0 || nMinCellY >= GRID_ROWS || nMaxCellY < 0) && nMaxCellX >= nMinCellX &&
                    nMaxCellY >= nMinCellY) {
                    q.valid = 1;
                    q.u = u; q.v = v; q.radius = rad;
                    q.ur = A.mpProjXR ? pXR : 0.f;
                    q.minCx = (unsigned char) nMinCellX; q.maxCx = (unsigned char) nMaxCellX;
                    q.minCy = (unsigned char) nMinCellY; q.maxCy = (unsigned char) nMaxCellY;
                    if (A.mode == 2) {
                        q.minLevel = (signed char) (lvl - 1); q.maxLevel = (signed char) (lvl + 1);
                        q.angle = pAng;
                    } else {
                        q.minLevel = (signed char) (A.checkLevel ? lvl - 1 : -1);
                        q.maxLevel = (signed char) (A.checkLevel ? lvl : -1);
                    }
                }
            }
        } else if (has) {
            float X[3] = {X0, X1, X2};
            if (A.unitWorld) { X[0] = (lk0.x - A.cx) / A.fx; X[1] = (lk0.y - A.cy) / A.fy; X[2] = 1.f; }   // k_backproject_unit's expressions
            const float xc = (Rcw[0] * X[0] + Rcw[1] * X[1] + Rcw[2] * X[2]) + tcw[0];
            const float yc = (Rcw[3] * X[0] + Rcw[4] * X[1] + Rcw[5] * X[2]) + tcw[1];
            const float zc = (Rcw[6] * X[0] + Rcw[7] * X[1] + Rcw[8] * X[2]) + tcw[2];
            const float invzc = (float) (1.0 / (double) zc);
            if (!(invzc < 0)) {
                const float u = A.fx * xc * invzc + A.cx;
                const float v = A.fy * yc * invzc + A.cy;
                if (!(u < A.minX || u > A.maxX) && !(v < A.minY || v > A.maxY)) {
                    const ygzf_kp lk = lk0;
                    const int oct = lk.octave;
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
                        q.u = u; q.v = v; q.radius = r; q.ur = u - A.mbf * invzc; q.angle = lk.angle;
                        q.minCx = (unsigned char) nMinCellX; q.maxCx = (unsigned char) nMaxCellX;
                        q.minCy = (unsigned char) nMinCellY; q.maxCy = (unsigned char) nMaxCellY;
                        q.minLevel = (signed char) minL; q.maxLevel = (signed char) maxL;
                    }
                }
            }
        }
        if (qr == 0) {
            if (S > 1) store_qp(&L.qp[i], q);   // read back by the workgroup that takes the pair over
            else L.qp[i] = q;
            L.qang[i] = q.angle;
            L.qobs[i] = q.hasObs;
        }
        // ---- speculative candidates: the 4 best acceptable (dist <= TH_HIGH) candidates against the INITIAL ownership,
        // one thread per query, no cross-lane traffic.  The in-order pass takes the first of them that is still free: the
        // current candidate set is a subset of the initial one, so that IS the current minimum.  Only when all four have
        // been taken does it fall back to a full cooperative rescan.
        unsigned K[8], J[8];   // (dist << 16 | visiting order, Cur index); entries 5..8 only with A.specDeep
#pragma unroll
        for (int e = 0; e < 8; e++) { K[e] = 0xFFFFFFFFu; J[e] = 0; }
        if (q.valid) {
            const bool bCheckLevels = (q.minLevel > 0) || (q.maxLevel >= 0);
            // One candidate = a chain of dependent LDS reads (list -> level / position / owner -> descriptor): the filters are evaluated as
            // predicates, not as early exits, so that the reads behind them are issued together (three waits per candidate instead of six).
            // Measured and dropped: four candidates per step (the runs are short: wasted evaluations), whole grid columns per lane.
            auto eval = [&](int li, unsigned ord, unsigned &jj) -> unsigned {
                const int i2 = L.list[li];
                jj = (unsigned) i2;
                bool ok = true;
                if (bCheckLevels) {
                    const int o = L.octave[i2];
                    ok = ok && !(o < q.minLevel) && !(q.maxLevel >= 0 && o > q.maxLevel);
                }
                const float distx = L.cx[i2] - q.u, disty = L.cy[i2] - q.v;
                ok = ok && (fabsf(distx) < q.radius && fabsf(disty) < q.radius) && L.owner[i2] != 2;
                if (uRight) {
                    const float ur2 = uRight[i2];
                    if (ur2 > 0 && fabsf(q.ur - ur2) > q.radius) ok = false;
                }
                unsigned long long d0, d1, d2, d3;
                if (A.descInLds) {
                    lds_desc(L.desc, i2, d0, d1, d2, d3);
                } else {
                    const unsigned long long *d = (const unsigned long long *) (curDesc + (size_t) i2 * 32);
                    d0 = d[0]; d1 = d[1]; d2 = d[2]; d3 = d[3];
                }
                const unsigned dist = __popcll(q0 ^ d0) + __popcll(q1 ^ d1) + __popcll(q2 ^ d2) + __popcll(q3 ^ d3);
                if ((A.mode == 0 || A.mode == 2) && dist > (unsigned) A.maxDist) ok = false;   // modes 1, 3 need the runner-up even when it is far
                return ok ? ((dist << 16) | (ord & 0xFFFFu)) : 0xFFFFFFFFu;
            };
            auto insert = [&](unsigned key, unsigned jj) {
                if (A.specDeep) {   // sorted list of eight: one compare-exchange sweep (branch-free)
                    if (key < K[7]) {
                        unsigned ck = key;
                        unsigned cj = jj;
#define YGZF_CEX(K, J) do { if (ck < K) { const unsigned tk = K; const unsigned tj = J; K = ck; J = cj; ck = tk; cj = tj; } } while (0)
                        YGZF_CEX(K[0], J[0]); YGZF_CEX(K[1], J[1]); YGZF_CEX(K[2], J[2]); YGZF_CEX(K[3], J[3]);
                        YGZF_CEX(K[4], J[4]); YGZF_CEX(K[5], J[5]); YGZF_CEX(K[6], J[6]); YGZF_CEX(K[7], J[7]);
#undef YGZF_CEX
                    }
                } else if (key < K[3]) {   // insert into the sorted quadruple
                    if (key < K[2]) {
                        K[3] = K[2]; J[3] = J[2];
                        if (key < K[1]) {
                            K[2] = K[1]; J[2] = J[1];
                            if (key < K[0]) { K[1] = K[0]; J[1] = J[0]; K[0] = key; J[0] = jj; }
                            else { K[1] = key; J[1] = jj; }
                        } else { K[2] = key; J[2] = jj; }
                    } else { K[3] = key; J[3] = jj; }
                }
            };
            unsigned colBase = 0;   // candidates of the grid columns before ix: `ord` is the position in the reference's visiting order
            for (int ix = q.minCx; ix <= q.maxCx; ix++) {
                const int c0 = ix * GRID_ROWS;
                const int s = L.cellStart[c0 + q.minCy], e = L.cellStart[c0 + q.maxCy + 1];
                for (int li = s + qr; li < e; li += LPQ) {
                    unsigned jj;
                    const unsigned key = eval(li, colBase + (unsigned) (li - s), jj);
                    insert(key, jj);
                }
                colBase += (unsigned) (e - s);
            }
        }
        if (LPQ > 1) {   // the lanes of a query fold their lists: pairs, quads, eights, ... (LPQ is the same for the whole wave)
            spec_merge8<0xb1>(K, J);                      // quad_perm [1,0,3,2]
            if (LPQ >= 4) spec_merge8<0x4e>(K, J);        // quad_perm [2,3,0,1]
            if (LPQ >= 8) spec_merge8<0x141>(K, J);       // row_half_mirror
            if (LPQ >= 16) spec_merge8<0x140>(K, J);      // row_mirror
            if (LPQ >= 32) spec_merge8_xor<16>(K, J, lane);
            if (LPQ >= 64) spec_merge8_xor<32>(K, J, lane);
#pragma unroll
            for (int e = 0; e < 8; e++) J[e] = K[e] == 0xFFFFFFFFu ? 0u : J[e];
        }
        if (qr != 0) continue;
        if (S > 1) {
            unsigned char *X = A.splitX + (long long) pair * A.capLast * kMatchSplitRec;
            st_dev4(X + 16 * (size_t) i, K[0], K[1], K[2], K[3]);
            st_dev4(X + 16 * (size_t) A.capLast + 16 * (size_t) i, K[4], K[5], K[6], K[7]);
            st_dev(X + 32 * (size_t) A.capLast + 8 * (size_t) i, (J[0] & 0xFFFFu) | (J[1] << 16));
            st_dev(X + 32 * (size_t) A.capLast + 8 * (size_t) i + 4, (J[2] & 0xFFFFu) | (J[3] << 16));
            st_dev(X + 40 * (size_t) A.capLast + 8 * (size_t) i, (J[4] & 0xFFFFu) | (J[5] << 16));
            st_dev(X + 40 * (size_t) A.capLast + 8 * (size_t) i + 4, (J[6] & 0xFFFFu) | (J[7] << 16));
            st_dev(X + 48 * (size_t) A.capLast + 4 * (size_t) i, __float_as_uint(q.angle));
            st_dev(X + 52 * (size_t) A.capLast + 4 * (size_t) i, q.hasObs);
        } else {
            L.specKey[i] = make_uint4(K[0], K[1], K[2], K[3]);
            L.specI2[i] = make_ushort4((unsigned short) J[0], (unsigned short) J[1], (unsigned short) J[2], (unsigned short) J[3]);
            if (A.specDeep) {
                L.specKeyB[i] = make_uint4(K[4], K[5], K[6], K[7]);
                L.specI2B[i] = make_ushort4((unsigned short) J[4], (unsigned short) J[5], (unsigned short) J[6], (unsigned short) J[7]);
            }
        }
    }
    if (dbg) __syncthreads();   // (debug builds of the launch only: the stamp is the workgroup's scan time, not the first wave's)
    const long long tScan = dbg ? wall_clock64() : 0;
    if (S > 1) {   // hand-over: the last workgroup to arrive owns the pair from here on
        // the records and query parameters above went out as device-scope stores: once they are acknowledged (vmcnt 0) the counter may move
        if (A.handoverFence || !kHandoverScopedAccess) __threadfence();
        __builtin_amdgcn_s_waitcnt(0);
        __syncthreads();
        if (tid == 0) s_tmp[19] = atomicAdd(&A.splitCnt[pair], 1);
        __syncthreads();
        if (s_tmp[19] != S - 1) return;
        if (tid == 0) A.splitCnt[pair] = 0;
        if (A.handoverFence || !kHandoverScopedAccess) __threadfence();
        const unsigned char *X = A.splitX + (long long) pair * A.capLast * kMatchSplitRec;
        for (int i = tid; i < nq; i += kMatchBlock) {
            L.specKey[i] = ld_dev4(X + 16 * (size_t) i);
            const unsigned ja = ld_dev(X + 32 * (size_t) A.capLast + 8 * (size_t) i), jb = ld_dev(X + 32 * (size_t) A.capLast + 8 * (size_t) i + 4);
            L.specI2[i] = make_ushort4((unsigned short) ja, (unsigned short) (ja >> 16), (unsigned short) jb, (unsigned short) (jb >> 16));
            if (A.specDeep) {
                L.specKeyB[i] = ld_dev4(X + 16 * (size_t) A.capLast + 16 * (size_t) i);
                const unsigned jc = ld_dev(X + 40 * (size_t) A.capLast + 8 * (size_t) i), jd = ld_dev(X + 40 * (size_t) A.capLast + 8 * (size_t) i + 4);
                L.specI2B[i] = make_ushort4((unsigned short) jc, (unsigned short) (jc >> 16), (unsigned short) jd, (unsigned short) (jd >> 16));
            }
            L.qang[i] = __uint_as_float(ld_dev(X + 48 * (size_t) A.capLast + 4 * (size_t) i));
            L.qobs[i] = (unsigned char) ld_dev(X + 52 * (size_t) A.capLast + 4 * (size_t) i);
        }
    }
    __syncthreads();
    if (dbg && tid == 0) { dbg[0] = tStart; dbg[1] = tGrid; dbg[2] = tScan; }
    STAMP(3);
    if (A.mode != 3 && A.serialOrder != 1) {
        // ---- in-order resolution as a FIXPOINT, all waves (modes 0, 1, 2).
        // Sequentially, query q sees as taken what EARLIER queries whose MapPoint has observations have taken (entries owned at the start are not in
        // its list; a MapPoint without observations does not block, :1301-1303): it picks the first free entry of its (dist, order)-sorted list
        // (mode 1: the first two, best and runner-up, and applies the accept rule :112-121 to them).  Iterate
        //     claim(e) = min{ q' : q' blocking, q' currently takes e },   q re-picks among the entries with claim(e) >= q
        // from "everybody takes his first entry" (mode 1: "nobody takes anything"): after k rounds the first k queries hold their sequential answer (a query's pick only depends on the picks
        // of the queries before it), so the iteration ends, and a fixpoint IS the sequential assignment (same induction).  The chains of queries
        // that push each other along are short (6-10 rounds for 1000 queries; the one-wave pass below retires a handful of queries per round).
        // A query whose list runs out gets the NEXT eight entries of its sorted candidate list (scan_after: independent of the claims, so every
        // such query is extended in the same pass, one wave each); two extensions per query, kExtSlots in all, 96 rounds -- beyond that the
        // pair is handed, untouched, to the one-wave pass.
        constexpr int BIG = 0x7FFFFFFF;
        const bool two = A.mode == 1;
        int *claimBuf[2] = {L.claim, L.match};
        int *choiceOf = L.events;                    // entry i is only ever touched by the thread that owns query i (i mod kMatchBlock)
        for (int i = tid; i < nt; i += kMatchBlock) { L.claim[i] = BIG; L.match[i] = BIG; }
        for (int i = tid; i < nq; i += kMatchBlock) {
            choiceOf[i] = (!two && L.specKey[i].x < kNoKey) ? (int) L.specI2[i].x : -1;   // (modes 0 / 2: the first entries, one round saved)
            L.extOf[2 * i] = 0xFF; L.extOf[2 * i + 1] = 0xFF;
        }
        if (tid < 4) s_tmp[16 + tid] = 0;            // [16], [17]: "a pick moved" of even / odd rounds; [18]: queries to extend; [19]: slots in use
        if (tid == 0) s_tmp[15] = 0;                 // hand the pair to the one-wave pass
        __syncthreads();
        int rounds = 0, nExtended = 0;
        for (;; rounds++) {
            int *cl = claimBuf[rounds & 1], *other = claimBuf[(rounds & 1) ^ 1];
            for (int i = tid; i < nq; i += kMatchBlock) {
                const int ch = choiceOf[i];
                if (ch >= 0 && L.qobs[i]) atomicMin(&cl[ch], i);
            }
            __syncthreads();
            int changed = 0;
            for (int i = tid; i < nq; i += kMatchBlock) {
                const uint4 keys = L.specKey[i];
                if (keys.x >= kNoKey) continue;
                const ushort4 idx = L.specI2[i];
                unsigned k1 = kNoKey, k2 = kNoKey;
                int b1 = -1, b2 = -1;
                bool ended = false, enough = false;      // the list has no further entry / the picks are complete
                if (!two) {     // modes 0 / 2, lists of four: the first free entry
                    const int c0 = cl[idx.x], c1 = cl[idx.y], c2 = cl[idx.z], c3 = cl[idx.w];   // four reads in flight together
                    if (c0 >= i) { b1 = idx.x; enough = true; }
                    else if (keys.y >= kNoKey) ended = true;
                    else if (c1 >= i) { b1 = idx.y; enough = true; }
                    else if (keys.z >= kNoKey) ended = true;
                    else if (c2 >= i) { b1 = idx.z; enough = true; }
                    else if (keys.w >= kNoKey) ended = true;
                    else if (c3 >= i) { b1 = idx.w; enough = true; }
                } else {        // mode 1, lists of eight: the first two free entries
                    const uint4 kb = L.specKeyB[i];
                    const ushort4 ib = L.specI2B[i];
                    const unsigned kk[8] = {keys.x, keys.y, keys.z, keys.w, kb.x, kb.y, kb.z, kb.w};
                    const int ii[8] = {idx.x, idx.y, idx.z, idx.w, ib.x, ib.y, ib.z, ib.w};
                    int cv[8];
#pragma unroll
                    for (int e = 0; e < 8; e++) cv[e] = cl[ii[e]];      // the claims of the whole list in flight together
#pragma unroll
                    for (int e = 0; e < 8; e++) {
                        if (ended || enough) continue;
                        if (kk[e] >= kNoKey) { ended = true; continue; }
                        if (cv[e] < i) continue;
                        if (b1 < 0) { b1 = ii[e]; k1 = kk[e]; }
                        else { b2 = ii[e]; k2 = kk[e]; enough = true; }
                    }
                }
                bool exhausted = !ended && !enough;
                if (exhausted && L.extOf[2 * i] != 0xFF) {              // the extensions of this query (rare)
                    for (int blk = 0; blk < 2 && exhausted; blk++) {
                        const int slot = L.extOf[2 * i + blk];
                        if (slot == 0xFF) break;
                        for (int e = 0; e < 8 && !ended && !enough; e++) {
                            const unsigned key = L.extKey[8 * slot + e];
                            const int i2 = L.extI2[8 * slot + e];
                            if (key >= kNoKey) { ended = true; break; }
                            if (cl[i2] < i) continue;
                            if (b1 < 0) { b1 = i2; k1 = key; enough = !two; }
                            else { b2 = i2; k2 = key; enough = true; }
                        }
                        exhausted = !ended && !enough;
                    }
                }
                int pick;
                if (exhausted) {
                    pick = -2;
                    const int w = atomicAdd(&s_tmp[18], 1);
                    if (w < kExtSlots) L.extWork[w] = i;
                } else if (!two) {
                    pick = b1;                           // modes 0 / 2: every list entry is acceptable (dist <= maxDist)
                } else {
                    pick = -1;
                    const int bestDist = (int) (k1 >> 16);
                    if (b1 >= 0 && bestDist <= TH_HIGH) {
                        const int bestDist2 = (int) (k2 >> 16);      // 256 when there is no runner-up
                        const int bestLevel = L.octave[b1], bestLevel2 = (b2 >= 0 && bestDist2 < 256) ? (int) L.octave[b2] : -1;
                        if (!(bestLevel == bestLevel2 && (float) bestDist > A.nnratio * (float) bestDist2)) pick = b1;
                    }
                }
                if (pick != choiceOf[i]) { choiceOf[i] = pick; changed = 1; }
            }
            for (int i = tid; i < nt; i += kMatchBlock) other[i] = BIG;     // the next round's claims start clean (nobody reads `other` now)
            if (tid == 0) s_tmp[16 + ((rounds & 1) ^ 1)] = 0;
            if (changed) s_tmp[16 + (rounds & 1)] = 1;
            __syncthreads();
            const int nWork = min(s_tmp[18], kExtSlots);
            if (nWork == 0 && !s_tmp[16 + (rounds & 1)]) break;   // fixpoint, nobody waits for an extension (tested BEFORE the round cap: a converged state is never thrown away)
            if (rounds >= 95) { if (tid == 0) s_tmp[15] = 1; break; }
            if (nWork == 0) continue;                    // some pick moved: another round, on the other claim buffer
            // extensions: one wave per query; the claims play no part, so all of them at once
            for (int w = wave; w < nWork; w += kMatchBlock / 64) {
                const int qx = L.extWork[w];
                const int blk = L.extOf[2 * qx] == 0xFF ? 0 : (L.extOf[2 * qx + 1] == 0xFF ? 1 : 2);
                int slot = 0;
                if (lane == 0) slot = (blk < 2 && A.serialOrder != 2) ? atomicAdd(&s_tmp[19], 1) : kExtSlots;
                slot = __builtin_amdgcn_readfirstlane(slot);
                if (slot >= kExtSlots) { if (lane == 0) s_tmp[15] = 1; continue; }
                unsigned afterKey;
                if (blk == 0) afterKey = two ? L.specKeyB[qx].w : L.specKey[qx].w;
                else afterKey = L.extKey[8 * (int) L.extOf[2 * qx] + 7];
                const QueryParam q = load_qp(&L.qp[qx]);
                const unsigned long long *qd = (const unsigned long long *) (mpDesc + (size_t) qx * 32);
                unsigned oK[8], oJ[8];
                scan_after(A, L, q, qd[0], qd[1], qd[2], qd[3], curDesc, uRight, lane, afterKey, two ? 256u : (unsigned) A.maxDist, oK, oJ);
                if (lane == 0) {
#pragma unroll
                    for (int e = 0; e < 8; e++) { L.extKey[8 * slot + e] = oK[e]; L.extI2[8 * slot + e] = (unsigned short) oJ[e]; }
                    L.extOf[2 * qx + blk] = (unsigned char) slot;
                }
            }
            nExtended += nWork;
            __syncthreads();
            if (tid == 0) s_tmp[18] = 0;
            if (s_tmp[15]) break;
            // (the barrier at the top of the next round orders the reset above against that round's requests)
        }
        __syncthreads();
        const bool serial = s_tmp[15] != 0;
        if (dbg && tid == 0) dbg[6] = serial ? -1 : nExtended * 1000 + rounds + 1;
        if (serial && tid == 0 && A.serialFallbacks) atomicAdd(A.serialFallbacks, 1u);   // always on: a crowded frame that loses the fixpoint's speed shows up in the profile
        if (serial) {
            for (int i = tid; i < nt; i += kMatchBlock) { L.claim[i] = 64; L.match[i] = -1; }
            __syncthreads();
        } else {
            // commit: the LAST query that picked a keypoint holds it (an earlier holder without observations is overwritten, :1301-1303);
            // every pick counts as a match and votes (the reference's rotHist keeps the overwritten vote too, :1327-1345)
            STAMP(4);
            for (int i = tid; i < nt; i += kMatchBlock) L.match[i] = -1;
            if (tid == 0) { s_tmp[16] = 0; s_tmp[17] = 0; }
            __syncthreads();
            const bool doOri = A.checkOri != 0 && !two;
            const float factor = 1.0f / HISTO_LENGTH;
            int mine = 0;
            for (int i = tid; i < nq; i += kMatchBlock) {
                const int ch = choiceOf[i];
                if (ch < 0) { choiceOf[i] = -1; continue; }
                atomicMax(&L.match[ch], i);
                mine++;
                if (doOri) {
                    float rot = L.qang[i] - L.cang[ch];
                    if (rot < 0.0) rot += 360.0f;
                    int bin = (int) roundf(rot * factor);
                    if (bin == HISTO_LENGTH) bin = 0;
                    atomicAdd(&s_hist[bin], 1);
                    choiceOf[i] = (bin << 24) | ch;
                }
            }
            mine = wave_sum(mine);
            if (lane == 0 && mine) atomicAdd(&s_tmp[16], mine);
            __syncthreads();
            for (int i = tid; i < nq; i += kMatchBlock) {
                const int ev = choiceOf[i];
                if (ev < 0) continue;
                const int ch = ev & 0xFFFFFF;
                if (L.match[ch] == i) L.owner[ch] = L.qobs[i] ? 2 : 1;
            }
            __syncthreads();
            if (doOri) {
                int ind1 = -1, ind2 = -1, ind3 = -1;
                int max1 = 0, max2 = 0, max3 = 0;
                for (int b = 0; b < HISTO_LENGTH; b++) {
                    const int sc = s_hist[b];
                    if (sc > max1) { max3 = max2; max2 = max1; max1 = sc; ind3 = ind2; ind2 = ind1; ind1 = b; }
                    else if (sc > max2) { max3 = max2; max2 = sc; ind3 = ind2; ind2 = b; }
                    else if (sc > max3) { max3 = sc; ind3 = b; }
                }
                if (max2 < 0.1f * (float) max1) { ind2 = -1; ind3 = -1; }
                else if (max3 < 0.1f * (float) max1) { ind3 = -1; }
                int removed = 0;
                for (int i = tid; i < nq; i += kMatchBlock) {
                    const int ev = choiceOf[i];
                    if (ev < 0) continue;
                    const int bin = ev >> 24, idx = ev & 0xFFFFFF;
                    if (bin != ind1 && bin != ind2 && bin != ind3) {
                        L.owner[idx] = 0;
                        L.match[idx] = -2;
                        removed++;
                    }
                }
                removed = wave_sum(removed);
                if (lane == 0 && removed) atomicAdd(&s_tmp[17], removed);
                __syncthreads();
            }
            for (int i = tid; i < nt; i += kMatchBlock) { ownerOut[i] = L.owner[i]; matchOut[i] = L.match[i]; }
            if (tid == 0) A.nmatches[pair] = s_tmp[16] - s_tmp[17];
            STAMP(5);
            if (dbg && tid == 0) dbg[7] = nq;
            return;
        }
    }
    if (wave != 0) return;

    // ---- in-order resolution by one wave.  Only LDS is touched inside the loop (a global store followed by the
    // ordering the next iteration needs would cost a full memory round trip per query).
    int nmatches = 0, nEvents = 0, nRescan = 0;
    const float factor = 1.0f / HISTO_LENGTH;
    volatile unsigned char *vowner = L.owner;
    const bool doOri = A.checkOri && (A.mode == 0 || A.mode == 2);
    const unsigned long long lane_lt = (1ull << lane) - 1ull;
    if (A.mode == 3) {
        // SearchForInitialization :394-447.  State: L.claim[i2] = vMatchedDistance, L.match[i2] = vnMatches21, L.events[i1] = vnMatches12
        // (| bin << 24).  A candidate counts only while its recorded distance is larger than this query's (:414-415); recorded
        // distances only ever shrink, so the candidates valid NOW are a subset of the speculative (dist, order)-sorted list: best
        // and runner-up are its first two valid entries, a full list that runs out first is rescanned.  A later query may take a
        // keypoint over from an earlier one (:427-430); the histogram keeps the earlier vote (:441), as the reference does.
        volatile int *vdist = L.claim;
        volatile int *v21 = L.match;
        volatile int *v12 = L.events;
        for (int i = lane; i < nt; i += 64) vdist[i] = 0x7FFFFFFF;
        for (int i = lane; i < nq; i += 64) v12[i] = -1;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const int TH_LOW = 50;
        for (int i = 0; i < nq; i++) {
            const uint4 keys = L.specKey[i];
            if (keys.x >= kNoKey) continue;
            const ushort4 idx = L.specI2[i];
            const unsigned kk[4] = {keys.x, keys.y, keys.z, keys.w};
            const int ii[4] = {idx.x, idx.y, idx.z, idx.w};
            unsigned k1 = kNoKey, k2 = kNoKey;
            int b1 = -1, b2 = -1;
            bool exhausted = true;
            for (int e = 0; e < 4; e++) {
                if (kk[e] >= kNoKey) { exhausted = false; break; }
                if (vdist[ii[e]] <= (int) (kk[e] >> 16)) continue;
                if (b1 < 0) { b1 = ii[e]; k1 = kk[e]; }
                else { b2 = ii[e]; k2 = kk[e]; exhausted = false; break; }
            }
            if (exhausted) {
                const QueryParam q = load_qp(&L.qp[i]);
                const unsigned long long *qd = (const unsigned long long *) (mpDesc + (size_t) i * 32);
                k1 = scan_query(A, L, q, qd[0], qd[1], qd[2], qd[3], curDesc, nullptr, lane, &b1, &k2, &b2, vdist);
                nRescan++;
            }
            const int bestDist = (int) (k1 >> 16);
            if (b1 < 0 || bestDist > TH_LOW) continue;
            const int bestDist2 = (int) (k2 >> 16);
            const bool noSecond = b2 < 0 || bestDist2 >= 256;        // bestDist2 stays INT_MAX in the reference
            if (!noSecond && !((float) bestDist < (float) bestDist2 * A.nnratio)) continue;
            const int old = v21[b1];
            if (old >= 0) nmatches--;
            nmatches++;
            float rot = L.qang[i] - L.cang[b1];
            if (rot < 0.0) rot += 360.0f;
            int bin = (int) roundf(rot * factor);
            if (bin == HISTO_LENGTH) bin = 0;
            if (lane == 0) {
                if (old >= 0) v12[old] = -1;
                v12[i] = (bin << 24) | b1;
                v21[b1] = i;
                vdist[b1] = bestDist;
                if (A.checkOri) s_hist[bin]++;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        int ind1 = -1, ind2 = -1, ind3 = -1;
        if (A.checkOri) {
            int max1 = 0, max2 = 0, max3 = 0;
            for (int b = 0; b < HISTO_LENGTH; b++) {
                const int s = s_hist[b];
                if (s > max1) { max3 = max2; max2 = max1; max1 = s; ind3 = ind2; ind2 = ind1; ind1 = b; }
                else if (s > max2) { max3 = max2; max2 = s; ind3 = ind2; ind2 = b; }
                else if (s > max3) { max3 = s; ind3 = b; }
            }
            if (max2 < 0.1f * (float) max1) { ind2 = -1; ind3 = -1; }
            else if (max3 < 0.1f * (float) max1) { ind3 = -1; }
        }
        int removed = 0;
        int *m12 = A.match12 + (long long) pair * A.kpStrideLast;
        for (int i = lane; i < nq; i += 64) {
            int ev = v12[i];
            int m = -1;
            if (ev >= 0) {
                const int bin = ev >> 24;
                m = ev & 0xFFFFFF;
                if (A.checkOri && bin != ind1 && bin != ind2 && bin != ind3) { m = -1; removed++; }
            }
            m12[i] = m;
        }
        removed = wave_sum(removed);
        nmatches -= removed;
        if (lane == 0) A.nmatches[pair] = nmatches;
        return;
    }
    if (A.mode == 1) {
        // SearchByProjection(F, MapPoints) :43-126.  Best and second-best among the candidates that are free NOW = the first two free
        // entries of the (dist, order)-sorted speculative list; a full list that runs out before both are found is rescanned.  Accept rule
        // :112-121.  In-order semantics, 64 queries at a time (as the mode-0 loop below): a pending lane is "in conflict" when an EARLIER
        // pending lane whose MapPoint has observations commits, in this round, to one of the two keypoints it picked -- that would change
        // its best / runner-up -- or when its list is exhausted.  Lanes before the first conflict are final and commit together; the
        // first conflicting lane re-picks next round (or takes the cooperative rescan).  One query per step took 0.7 ms for 1000 points.
        volatile int *vclaim = L.claim;
        for (int tile = 0; tile < nq; tile += 64) {
            const int i = tile + lane;
            const bool active = i < nq;
            uint4 keys = make_uint4(kNoKey, kNoKey, kNoKey, kNoKey), keysB = keys;
            ushort4 idx = make_ushort4(0, 0, 0, 0), idxB = idx;
            bool obs = false;
            if (active) {
                keys = L.specKey[i]; idx = L.specI2[i]; obs = L.qobs[i] != 0;
                if (A.specDeep) { keysB = L.specKeyB[i]; idxB = L.specI2B[i]; }
            }
            const int depth = A.specDeep ? 8 : 4;
            bool pending = active && keys.x < kNoKey;
            const unsigned kk[8] = {keys.x, keys.y, keys.z, keys.w, keysB.x, keysB.y, keysB.z, keysB.w};
            const int ii[8] = {idx.x, idx.y, idx.z, idx.w, idxB.x, idxB.y, idxB.z, idxB.w};
            // the levels of the list's keypoints do not change: read once per tile; the ownership bytes are read once per round, all of them
            // together (the walk below then runs on registers: it used to follow the list through up to ten dependent LDS reads per round)
            int lev[8];
#pragma unroll
            for (int e = 0; e < 8; e++) lev[e] = (e < depth && kk[e] < kNoKey) ? (int) L.octave[ii[e]] : -1;
            while (__ballot(pending)) {
                unsigned char own[8];
#pragma unroll
                for (int e = 0; e < 8; e++) own[e] = e < depth ? vowner[ii[e]] : (unsigned char) 0;
                unsigned k1 = kNoKey, k2 = kNoKey;
                int b1 = -1, b2 = -1, l1 = -1, l2 = -1;
                bool exhausted = pending;     // walked all entries and every one was a real candidate
                if (pending) {
#pragma unroll
                    for (int e = 0; e < 8; e++) {
                        if (e >= depth) break;
                        if (kk[e] >= kNoKey) { exhausted = false; break; }
                        if (own[e] == 2) continue;
                        if (b1 < 0) { b1 = ii[e]; k1 = kk[e]; l1 = lev[e]; }
                        else { b2 = ii[e]; k2 = kk[e]; l2 = lev[e]; exhausted = false; break; }
                    }
                }
                const bool rescan = pending && exhausted;
                bool take = false;            // the query passes the accept rule with these picks
                if (pending && !rescan && b1 >= 0) {
                    const int bestDist = (int) (k1 >> 16);
                    if (bestDist <= TH_HIGH) {
                        const int bestDist2 = (int) (k2 >> 16);      // 256 when there is no runner-up
                        const int bestLevel = l1, bestLevel2 = (b2 >= 0 && bestDist2 < 256) ? l2 : -1;
                        take = !(bestLevel == bestLevel2 && (float) bestDist > A.nnratio * (float) bestDist2);
                    }
                }
                if (take && obs) atomicMin((int *) &vclaim[b1], lane);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                const bool conflict = pending && (rescan || (b1 >= 0 && vclaim[b1] < lane) || (b2 >= 0 && vclaim[b2] < lane));
                __builtin_amdgcn_wave_barrier();
                if (take && obs) vclaim[b1] = 64;
                const unsigned long long cm = __ballot(conflict);
                const int first = cm ? (int) __ffsll((long long) cm) - 1 : 64;
                const bool done = pending && lane < first;
                const bool commit = done && take;
                const unsigned long long mcommit = __ballot(commit);
                if (mcommit) {
                    const unsigned long long noobs = __ballot(commit && !obs);
                    if (noobs == 0) {
                        if (commit) { vowner[b1] = 2; L.match[b1] = i; }
                    } else {
                        // a MapPoint without observations does not block its keypoint: later queries may overwrite it, so these commits
                        // must land in query order
                        unsigned long long mm = mcommit;
                        while (mm) {
                            const int k = (int) __ffsll((long long) mm) - 1;
                            mm &= mm - 1;
                            if (lane == k) { vowner[b1] = obs ? 2 : 1; L.match[b1] = i; }
                            __builtin_amdgcn_wave_barrier();
                        }
                    }
                    nmatches += __popcll(mcommit);
                }
                if (done) pending = false;
                // the first conflicting lane with an exhausted list: cooperative rescan against the current ownership
                const bool firstRescan = first < 64 && __builtin_amdgcn_readlane((int) rescan, first) != 0;
                if (firstRescan) {
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    const int qi = tile + first;
                    const QueryParam q = load_qp(&L.qp[qi]);
                    const unsigned long long *qd = (const unsigned long long *) (mpDesc + (size_t) qi * 32);
                    int c1 = -1, c2 = -1;
                    unsigned s2 = kNoKey;
                    const unsigned s1 = scan_query(A, L, q, qd[0], qd[1], qd[2], qd[3], curDesc, uRight, lane, &c1, &s2, &c2);
                    nRescan++;
                    const int bestDist = (int) (s1 >> 16);
                    if (c1 >= 0 && bestDist <= TH_HIGH) {
                        const int bestDist2 = (int) (s2 >> 16);
                        const int bestLevel = L.octave[c1], bestLevel2 = (c2 >= 0 && bestDist2 < 256) ? L.octave[c2] : -1;
                        if (!(bestLevel == bestLevel2 && (float) bestDist > A.nnratio * (float) bestDist2)) {
                            const int qobs = L.qobs[qi];
                            if (lane == 0) { vowner[c1] = qobs ? 2 : 1; L.match[c1] = qi; }
                            nmatches++;
                        }
                    }
                    if (lane == first) pending = false;
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            }
        }
    } else {
        // In-order semantics, 64 queries at a time.  Lane = query.  Every pending lane picks the first entry of its
        // speculative list that is free NOW; a lane is "in conflict" if an EARLIER pending lane (whose MapPoint has
        // observations, i.e. blocks) picked the same keypoint, or if its list is exhausted.  All lanes before the first
        // conflict are final -- no earlier query can still change what they see -- and commit together; the first
        // conflicting lane then re-picks against the updated ownership (or takes the cooperative rescan).  Each round
        // retires at least one query; rounds per tile = 1 + number of real conflicts.
        volatile int *vclaim = L.claim;
        for (int tile = 0; tile < nq; tile += 64) {
            const int i = tile + lane;
            const bool active = i < nq;
            uint4 keys = make_uint4(kNoKey, kNoKey, kNoKey, kNoKey);
            ushort4 idx = make_ushort4(0, 0, 0, 0);
            bool obs = false;
            if (active) { keys = L.specKey[i]; idx = L.specI2[i]; obs = L.qobs[i] != 0; }
            bool pending = active && keys.x < kNoKey;
            while (__ballot(pending)) {
                int choice = -1;
                bool rescan = false;
                const unsigned char o0 = vowner[idx.x], o1 = vowner[idx.y], o2 = vowner[idx.z], o3 = vowner[idx.w];   // four reads in flight together
                if (pending) {
                    if (o0 != 2) choice = idx.x;
                    else if (keys.y >= kNoKey) pending = false;
                    else if (o1 != 2) choice = idx.y;
                    else if (keys.z >= kNoKey) pending = false;
                    else if (o2 != 2) choice = idx.z;
                    else if (keys.w >= kNoKey) pending = false;
                    else if (o3 != 2) choice = idx.w;
                    else rescan = true;
                }
                if (pending && choice >= 0 && obs) atomicMin((int *) &vclaim[choice], lane);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                const bool conflict = pending && (rescan || (choice >= 0 && vclaim[choice] < lane));
                __builtin_amdgcn_wave_barrier();
                if (pending && choice >= 0 && obs) vclaim[choice] = 64;
                const unsigned long long cm = __ballot(conflict);
                const int first = cm ? (int) __ffsll((long long) cm) - 1 : 64;
                const bool commit = pending && lane < first;
                const unsigned long long mcommit = __ballot(commit);
                if (mcommit) {
                    const unsigned long long noobs = __ballot(commit && !obs);
                    if (noobs == 0) {
                        if (commit) { vowner[choice] = 2; L.match[choice] = i; }
                    } else {
                        // a MapPoint without observations does not block its keypoint: later queries may overwrite it, so
                        // these commits must land in query order
                        unsigned long long mm = mcommit;
                        while (mm) {
                            const int k = (int) __ffsll((long long) mm) - 1;
                            mm &= mm - 1;
                            if (lane == k) { vowner[choice] = obs ? 2 : 1; L.match[choice] = i; }
                            __builtin_amdgcn_wave_barrier();
                        }
                    }
                    if (doOri && commit) {
                        float rot = L.qang[i] - L.cang[choice];
                        if (rot < 0.0) rot += 360.0f;
                        int bin = (int) roundf(rot * factor);
                        if (bin == HISTO_LENGTH) bin = 0;
                        L.events[nEvents + __popcll(mcommit & lane_lt)] = (bin << 24) | choice;
                    }
                    const int nc = __popcll(mcommit);
                    nmatches += nc;
                    if (doOri) nEvents += nc;
                    if (commit) pending = false;
                }
                // the first conflicting lane with an exhausted list: cooperative rescan against the current ownership
                const bool firstRescan = first < 64 && __builtin_amdgcn_readlane((int) rescan, first) != 0;
                if (firstRescan) {
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    const int qi = tile + first;
                    const QueryParam q = load_qp(&L.qp[qi]);
                    const unsigned long long *qd = (const unsigned long long *) (mpDesc + (size_t) qi * 32);
                    int b = -1;
                    const unsigned key = scan_query(A, L, q, qd[0], qd[1], qd[2], qd[3], curDesc, uRight, lane, &b);
                    nRescan++;
                    if ((int) (key >> 16) <= A.maxDist) {
                        if (lane == first) { vowner[b] = obs ? 2 : 1; L.match[b] = qi; }
                        if (doOri) {
                            float rot = L.qang[qi] - L.cang[b];
                            if (rot < 0.0) rot += 360.0f;
                            int bin = (int) roundf(rot * factor);
                            if (bin == HISTO_LENGTH) bin = 0;
                            if (lane == 0) L.events[nEvents] = (bin << 24) | b;
                            nEvents++;
                        }
                        nmatches++;
                    }
                    if (lane == first) pending = false;
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    STAMP(4);
    // ---- rotation consistency (:1327-1345) + ComputeThreeMaxima (:1471-1502) ----
    if (doOri) {
        for (int e = lane; e < nEvents; e += 64) atomicAdd(&s_hist[L.events[e] >> 24], 1);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
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
            const int ev = L.events[e];
            const int bin = ev >> 24, idx = ev & 0xFFFFFF;
            if (bin != ind1 && bin != ind2 && bin != ind3) {
                L.owner[idx] = 0;
                L.match[idx] = -2;
                removed++;
            }
        }
        removed = wave_sum(removed);
        nmatches -= removed;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    for (int i = lane; i < nt; i += 64) { ownerOut[i] = L.owner[i]; matchOut[i] = L.match[i]; }
    if (lane == 0) A.nmatches[pair] = nmatches;
    STAMP(5);
    if (dbg && tid == 0) { dbg[6] = nRescan; dbg[7] = nq; }
#undef STAMP
}

// ------------------------------------------------------------------------------------------------------------------
// ORBmatcher::SearchByBoW(KeyFrame*, Frame&, vector<MapPoint*>&)  src/ORBmatcher.cc:155-263, the per-node brute force.
// A feature belongs to exactly one vocabulary node, so the nodes common to both FeatureVectors are independent problems: one wave
// per node walks that node's KeyFrame features in order (the "slot already matched" test :198-199 chains them), its lanes spread
// over the node's Frame features.  The rotation histogram (:215-246) spans all nodes: votes are counted with atomics and the
// culling runs in a second, tiny kernel.
// ------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_bow_nodes(int nNodes, const int *__restrict__ kfOff, const int *__restrict__ kfIdx,
                                                   const int *__restrict__ fOff, const int *__restrict__ fIdx,
                                                   const uint8_t *__restrict__ kfValid, const ygzf_kp *__restrict__ kfKeys,
                                                   const uint8_t *__restrict__ kfDesc, const ygzf_kp *__restrict__ fKeys,
                                                   const uint8_t *__restrict__ fDesc, float nnratio, int checkOri, int *__restrict__ match,
                                                   unsigned char *__restrict__ binOf, int *__restrict__ hist, int *__restrict__ nmatches) {
    const int lane = m_lane(), wave = threadIdx.x >> 6;
    const int node = blockIdx.x * 4 + wave;
    if (node >= nNodes) return;
    const int f0 = fOff[node], nF = fOff[node + 1] - f0;
    const int rounds = (nF + 63) >> 6;
    const float factor = 1.0f / HISTO_LENGTH;
    const int TH_LOW = 50;
    unsigned long long taken = 0;   // bit r: this lane's candidate of round r (position r*64 + lane) already carries a match
    int count = 0;
    for (int a = kfOff[node]; a < kfOff[node + 1]; a++) {
        const int iKF = kfIdx[a];
        if (!kfValid[iKF]) continue;
        const unsigned long long *qd = (const unsigned long long *) (kfDesc + (size_t) iKF * 32);
        const unsigned long long q0 = qd[0], q1 = qd[1], q2 = qd[2], q3 = qd[3];
        unsigned best = (256u << 16) | 0xFFFFu, best2 = (256u << 16) | 0xFFFFu;
        if (rounds <= 64) {
            for (int r = 0; r < rounds; r++) {
                const int b = r * 64 + lane;
                if (b >= nF || ((taken >> r) & 1ull)) continue;
                const unsigned long long *d = (const unsigned long long *) (fDesc + (size_t) fIdx[f0 + b] * 32);
                const unsigned dist = __popcll(q0 ^ d[0]) + __popcll(q1 ^ d[1]) + __popcll(q2 ^ d[2]) + __popcll(q3 ^ d[3]);
                const unsigned key = (dist << 16) | (unsigned) b;
                if (key < best) { best2 = best; best = key; }
                else if (key < best2) best2 = key;
            }
        }
        const unsigned wbest = wave_min_dpp(best);
        const unsigned long long who = __ballot(best == wbest);
        const int src = __ffsll((long long) who) - 1;
        const unsigned second = wave_min_dpp(lane == src ? best2 : min(best, best2));
        const int bestDist1 = (int) (wbest >> 16), bestDist2 = (int) (second >> 16);
        if (bestDist1 > TH_LOW) continue;
        if (!((float) bestDist1 < nnratio * (float) bestDist2)) continue;
        const int b = (int) (wbest & 0xFFFFu);
        if (lane == (b & 63)) taken |= 1ull << (b >> 6);
        if (lane == 0) {
            const int iF = fIdx[f0 + b];
            match[iF] = iKF;
            if (checkOri) {
                float rot = kfKeys[iKF].angle - fKeys[iF].angle;
                if (rot < 0.0) rot += 360.0f;
                int bin = (int) roundf(rot * factor);
                if (bin == HISTO_LENGTH) bin = 0;
                binOf[iF] = (unsigned char) bin;
                atomicAdd(&hist[bin], 1);
            }
        }
        count++;
    }
    if (lane == 0 && count) atomicAdd(nmatches, count);
}

__global__ __launch_bounds__(256) void k_bow_finish(int nF, int checkOri, int *__restrict__ match, const unsigned char *__restrict__ binOf,
                                                    const int *__restrict__ hist, int *__restrict__ nmatches) {
    if (!checkOri) return;
    __shared__ int s_removed;
    if (threadIdx.x == 0) s_removed = 0;
    __syncthreads();
    int ind1 = -1, ind2 = -1, ind3 = -1;
    int max1 = 0, max2 = 0, max3 = 0;
    for (int b = 0; b < HISTO_LENGTH; b++) {   // ComputeThreeMaxima :1471-1502
        const int s = hist[b];
        if (s > max1) { max3 = max2; max2 = max1; max1 = s; ind3 = ind2; ind2 = ind1; ind1 = b; }
        else if (s > max2) { max3 = max2; max2 = s; ind3 = ind2; ind2 = b; }
        else if (s > max3) { max3 = s; ind3 = b; }
    }
    if (max2 < 0.1f * (float) max1) { ind2 = -1; ind3 = -1; }
    else if (max3 < 0.1f * (float) max1) { ind3 = -1; }
    int removed = 0;
    for (int i = threadIdx.x; i < nF; i += blockDim.x) {
        if (match[i] < 0) continue;
        const int bin = binOf[i];
        if (bin != ind1 && bin != ind2 && bin != ind3) { match[i] = -2; removed++; }
    }
    if (removed) atomicAdd(&s_removed, removed);
    __syncthreads();
    if (threadIdx.x == 0) *nmatches -= s_removed;
}

void launch_bow(hipStream_t st, int nNodes, const int *kfOff, const int *kfIdx, const int *fOff, const int *fIdx, const uint8_t *kfValid,
                const ygzf_kp *kfKeys, const uint8_t *kfDesc, int nF, const ygzf_kp *fKeys, const uint8_t *fDesc, float nnratio, int checkOri, int *match,
                unsigned char *binOf, int *hist, int *nmatches) {
    if (nNodes > 0)
        hipLaunchKernelGGL(k_bow_nodes, dim3((nNodes + 3) / 4), dim3(256), 0, st, nNodes, kfOff, kfIdx, fOff, fIdx, kfValid, kfKeys, kfDesc, fKeys, fDesc,
                           nnratio, checkOri, match, binOf, hist, nmatches);
    hipLaunchKernelGGL(k_bow_finish, dim3(1), dim3(256), 0, st, nF, checkOri, match, binOf, hist, nmatches);
}

// ------------------------------------------------------------------------------------------------------------------
// ORBmatcher::SearchForTriangulation  src/ORBmatcher.cc:596-741 with CheckDistEpipolarLine :136-153 (LocalMapping::CreateNewMapPoints).
// The reference never sets vbMatched2 (:616 declares it, nothing writes it), so every KF1 feature is an independent problem: of the KF2
// features of its vocabulary node that carry no MapPoint, pass the stereo filter, lie within TH_LOW, outside the epipole's exclusion disc
// (both keypoints monocular, :668-673) and close enough to the epipolar line, it takes the least distance, the LAST of equals (the scan's
// `dist > bestDist` test lets an equal distance replace the holder, :662).  Whether a candidate passes does not depend on the scan's state,
// so the scan order is free: one wave per KF1 list entry, its lanes spread over the node's KF2 list, the answer is the minimum of
// (distance << 16 | 0xFFFF - position).  Rotation votes go to the same histogram + k_bow_finish as SearchByBoW.
// ------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_tri_nodes(TriArgs A) {
    const int lane = m_lane();
    const int a = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (a >= A.nEntries) return;
    int lo = 0, hi = A.nNodes;      // off1[lo] <= a < off1[hi]
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (A.off1[mid] <= a) lo = mid;
        else hi = mid;
    }
    const int i1 = A.idx1[a];
    if (A.hasMp1[i1]) return;
    const bool stereo1 = A.uR1 && A.uR1[i1] >= 0;
    if (A.onlyStereo && !stereo1) return;
    const ygzf_kp kp1 = A.keys1[i1];
    const unsigned long long *qd = (const unsigned long long *) (A.desc1 + (size_t) i1 * 32);
    const unsigned long long q0 = qd[0], q1 = qd[1], q2 = qd[2], q3 = qd[3];
    // epipolar line of kp1 in the second image, l = x1' F12 = [la lb lc]  (:139-141)
    const float la = kp1.x * A.F[0] + kp1.y * A.F[3] + A.F[6];
    const float lb = kp1.x * A.F[1] + kp1.y * A.F[4] + A.F[7];
    const float lc = kp1.x * A.F[2] + kp1.y * A.F[5] + A.F[8];
    const float den = la * la + lb * lb;
    const int b0 = A.off2[lo], nB = A.off2[lo + 1] - b0;
    const int TH_LOW = 50;
    unsigned best = 0xFFFFFFFFu;
    if (den != 0) {                 // den == 0: CheckDistEpipolarLine is false for every candidate (:146-147)
        for (int b = lane; b < nB; b += 64) {
            const int i2 = A.idx2[b0 + b];
            if (A.hasMp2[i2]) continue;
            const bool stereo2 = A.uR2 && A.uR2[i2] >= 0;
            if (A.onlyStereo && !stereo2) continue;
            const unsigned long long *d = (const unsigned long long *) (A.desc2 + (size_t) i2 * 32);
            const int dist = __popcll(q0 ^ d[0]) + __popcll(q1 ^ d[1]) + __popcll(q2 ^ d[2]) + __popcll(q3 ^ d[3]);
            if (dist > TH_LOW) continue;
            const ygzf_kp kp2 = A.keys2[i2];
            if (!stereo1 && !stereo2) {
                const float distex = A.ex - kp2.x;
                const float distey = A.ey - kp2.y;
                if (distex * distex + distey * distey < 100.0f * A.sf2[kp2.octave]) continue;
            }
            const float num = la * kp2.x + lb * kp2.y + lc;
            const float dsqr = num * num / den;
            if (!((double) dsqr < 3.84 * (double) A.sigma2[kp2.octave])) continue;
            const unsigned key = ((unsigned) dist << 16) | (unsigned) (0xFFFF - b);
            best = min(best, key);
        }
    }
    const unsigned wbest = wave_min_dpp(best);
    if (lane != 0) return;
    if (wbest == 0xFFFFFFFFu) return;
    const int i2 = A.idx2[b0 + (0xFFFF - (int) (wbest & 0xFFFFu))];
    A.match12[i1] = i2;
    atomicAdd(A.nmatches, 1);
    if (A.checkOri) {
        float rot = kp1.angle - A.keys2[i2].angle;
        if (rot < 0.0) rot += 360.0f;
        int bin = (int) roundf(rot * (1.0f / HISTO_LENGTH));
        if (bin == HISTO_LENGTH) bin = 0;
        A.binOf[i1] = (unsigned char) bin;
        atomicAdd(&A.hist[bin], 1);
    }
}

void launch_triangulation(hipStream_t st, const TriArgs &A) {
    if (A.nEntries > 0) hipLaunchKernelGGL(k_tri_nodes, dim3((A.nEntries + 3) / 4), dim3(256), 0, st, A);
    hipLaunchKernelGGL(k_bow_finish, dim3(1), dim3(256), 0, st, A.n1, A.checkOri, A.match12, A.binOf, A.hist, A.nmatches);
}

// ------------------------------------------------------------------------------------------------------------------
// Frame::isInFrustum for a batch of MapPoints  (src/Frame.cc:363-422; Tracking::SearchLocalPoints src/Tracking.cc:1544-1593 calls it
// for every local MapPoint before SearchByProjection).  One thread per point; its outputs are exactly the MapPoint fields mode 1 of
// k_match_last reads, so the two chain on the device without a host round trip.  MapPoint::PredictScale's
// ceil(logf(ratio) / logScaleFactor) is a step function of `ratio`; the host tabulates its steps with its own libm (levelStep[k] =
// smallest ratio that yields level >= k), so the level is found by comparisons and matches the CPU bit for bit.
// ------------------------------------------------------------------------------------------------------------------
__global__ void k_frustum(FrustumArgs A) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= A.n) return;
    A.inView[i] = 0;
    A.level[i] = 0;            // (the matcher indexes scaleFactors[level] only for in-view points; callers get 0 for the others -- no memset launch)
    if (A.candidate && !A.candidate[i]) return;
    const float *P = A.world + 3 * (size_t) i;
    const float PcX = (A.Rcw[0] * P[0] + A.Rcw[1] * P[1] + A.Rcw[2] * P[2]) + A.tcw[0];
    const float PcY = (A.Rcw[3] * P[0] + A.Rcw[4] * P[1] + A.Rcw[5] * P[2]) + A.tcw[1];
    const float PcZ = (A.Rcw[6] * P[0] + A.Rcw[7] * P[1] + A.Rcw[8] * P[2]) + A.tcw[2];
    if (PcZ < 0.0f) return;
    const float invz = 1.0f / PcZ;
    const float u = A.fx * PcX * invz + A.cx;
    const float v = A.fy * PcY * invz + A.cy;
    if (u < A.minX || u > A.maxX) return;
    if (v < A.minY || v > A.maxY) return;
    const float PO[3] = {P[0] - A.Ow[0], P[1] - A.Ow[1], P[2] - A.Ow[2]};
    const float dist = sqrtf(PO[0] * PO[0] + PO[1] * PO[1] + PO[2] * PO[2]);
    if (dist < A.minDistInv[i] || dist > A.maxDistInv[i]) return;
    const float *Pn = A.normal + 3 * (size_t) i;
    const float viewCos = (PO[0] * Pn[0] + PO[1] * Pn[1] + PO[2] * Pn[2]) / dist;
    if (viewCos < A.viewingCosLimit) return;
    const float ratio = A.mfMaxDistance[i] / dist;
    int level = 0;
    for (int k = 1; k < A.nLevels; k++) level += (ratio >= A.levelStep[k]) ? 1 : 0;
    A.inView[i] = 1;
    A.projX[i] = u;
    A.projXR[i] = u - A.mbf * invz;
    A.projY[i] = v;
    A.level[i] = level;
    A.viewCos[i] = viewCos;
}

void launch_frustum(hipStream_t st, const FrustumArgs &A) {
    if (A.n > 0) hipLaunchKernelGGL(k_frustum, dim3((A.n + 255) / 256), dim3(256), 0, st, A);
}

// ------------------------------------------------------------------------------------------------------------------
// MapPoint::ComputeDistinctiveDescriptors for a batch of MapPoints  (src/MapPoint.cc:211-271): per point, the observation whose
// descriptor has the least median Hamming distance to the others.  One wave per point, lane j owns observation j (+64, ...); the
// median of row i (the reference sorts the row and reads element (size_t)(0.5*(N-1))) is the k-th smallest, found by a 9-step
// bisection on the value with ballot counts instead of a sort.  First minimum wins (strict `<`).
// ------------------------------------------------------------------------------------------------------------------
constexpr int kDistinctMaxObs = 256;
__global__ __launch_bounds__(256) void k_distinctive(int nPoints, const int *__restrict__ obsOff, const uint8_t *__restrict__ desc,
                                                     int *__restrict__ best) {
    const int lane = m_lane(), p = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (p >= nPoints) return;
    const int o0 = obsOff[p], N = obsOff[p + 1] - o0;
    if (N <= 0) { if (lane == 0) best[p] = -1; return; }
    if (N > kDistinctMaxObs) return;   // k_distinctive_large's
    const int k = (int) (0.5 * (N - 1));
    const int per = (N + 63) >> 6;   // <= 4
    unsigned long long d[4][4];
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int j = r * 64 + lane;
        if (r < per && j < N) {
            const unsigned long long *q = (const unsigned long long *) (desc + (size_t) (o0 + j) * 32);
            d[r][0] = q[0]; d[r][1] = q[1]; d[r][2] = q[2]; d[r][3] = q[3];
        } else d[r][0] = d[r][1] = d[r][2] = d[r][3] = 0;
    }
    int bestMedian = 0x7FFFFFFF, bestIdx = 0;
    for (int i = 0; i < N; i++) {
        const unsigned long long *q = (const unsigned long long *) (desc + (size_t) (o0 + i) * 32);
        const unsigned long long q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3];
        int dist[4];
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int j = r * 64 + lane;
            dist[r] = (r < per && j < N) ? (int) (__popcll(q0 ^ d[r][0]) + __popcll(q1 ^ d[r][1]) + __popcll(q2 ^ d[r][2]) + __popcll(q3 ^ d[r][3])) : 1 << 20;
        }
        // smallest value m with #{dist <= m} > k
        int lo = 0, hi = 256;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            int cnt = 0;
#pragma unroll
            for (int r = 0; r < 4; r++) cnt += __popcll(__ballot(dist[r] <= mid));
            if (cnt > k) hi = mid; else lo = mid + 1;
        }
        if (lo < bestMedian) { bestMedian = lo; bestIdx = i; }
    }
    if (lane == 0) best[p] = bestIdx;
}

// The same for points with more than kDistinctMaxObs observations (long sessions: a landmark seen from hundreds of KeyFrames), which do not fit
// the register-resident form: one wave per such point, the median of row i from a 257-bin histogram of its distances (LDS atomics over the
// observations in steps of 64, then the first bin whose running count exceeds k).  `large` lists the points to process; the others are left
// to k_distinctive, which skips them.
__global__ __launch_bounds__(256) void k_distinctive_large(int nLarge, const int *__restrict__ large, const int *__restrict__ obsOff,
                                                           const uint8_t *__restrict__ desc, int *__restrict__ best) {
    __shared__ int s_hist[4][260];
    const int lane = m_lane(), wv = threadIdx.x >> 6, q = blockIdx.x * 4 + wv;
    if (q >= nLarge) return;
    const int p = large[q];
    const int o0 = obsOff[p], N = obsOff[p + 1] - o0;
    const int k = (int) (0.5 * (N - 1));
    int *hist = s_hist[wv];
    int bestMedian = 0x7FFFFFFF, bestIdx = 0;
    for (int i = 0; i < N; i++) {
        for (int b = lane; b < 260; b += 64) hist[b] = 0;
        __builtin_amdgcn_wave_barrier();
        const unsigned long long *qi = (const unsigned long long *) (desc + (size_t) (o0 + i) * 32);
        const unsigned long long q0 = qi[0], q1 = qi[1], q2 = qi[2], q3 = qi[3];
        for (int j = lane; j < N; j += 64) {
            const unsigned long long *d = (const unsigned long long *) (desc + (size_t) (o0 + j) * 32);
            atomicAdd(&hist[__popcll(q0 ^ d[0]) + __popcll(q1 ^ d[1]) + __popcll(q2 ^ d[2]) + __popcll(q3 ^ d[3])], 1);
        }
        __builtin_amdgcn_wave_barrier();
        // smallest value m with #{dist <= m} > k: lane l sums bins 5 l .. 5 l + 4 (257 bins over 52 lanes), wave prefix, then the lane whose run crosses k
        int loc[5], sum = 0;
#pragma unroll
        for (int t = 0; t < 5; t++) { const int b = 5 * lane + t; loc[t] = b < 257 ? hist[b] : 0; sum += loc[t]; }
        int incl = sum;
#pragma unroll
        for (int dlt = 1; dlt < 64; dlt <<= 1) {
            const int o = __shfl_up(incl, dlt);
            if (lane >= dlt) incl += o;
        }
        int run = incl - sum, med = 1 << 20;
#pragma unroll
        for (int t = 0; t < 5; t++) {
            run += loc[t];
            if (run > k && med == 1 << 20) med = 5 * lane + t;
        }
        const unsigned long long has = __ballot(med != 1 << 20);
        const int median = __shfl(med, __ffsll((long long) has) - 1);
        if (median < bestMedian) { bestMedian = median; bestIdx = i; }
        __builtin_amdgcn_wave_barrier();
    }
    if (lane == 0) best[p] = bestIdx;
}

void launch_distinctive(hipStream_t st, int nPoints, const int *obsOff, const uint8_t *desc, int *best, int nLarge, const int *large) {
    if (nPoints > 0) hipLaunchKernelGGL(k_distinctive, dim3((nPoints + 3) / 4), dim3(256), 0, st, nPoints, obsOff, desc, best);
    if (nLarge > 0) hipLaunchKernelGGL(k_distinctive_large, dim3((nLarge + 3) / 4), dim3(256), 0, st, nLarge, large, obsOff, desc, best);
}

// ------------------------------------------------------------------------------------------------------------------
// Frame::ComputeBoW (src/Frame.cc:495-500) -> DBoW2::TemplatedVocabulary::transform(feature, word_id, weight, nid, levelsup)
// (Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1240-1283): the descriptor is propagated down the k-ary vocabulary tree, at every node to
// the child whose centroid is nearest in Hamming distance (FORB::distance, FORB.cpp:82-101; first minimum wins: `d < best_d`).
// One wave per descriptor; the children of the current node go over the lanes (k = 10 for ORBvoc: one step per level), a 64-bit
// (distance << 32 | child position) minimum picks the first nearest child.  Outputs: the leaf NODE id (the host maps it to WordId /
// weight: those tables stay with the vocabulary object) and the node id at level L - levelsup (the FeatureVector key).
// ------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_bow_descend(int n, const uint8_t *__restrict__ desc, const int *__restrict__ childOff,
                                                     const int *__restrict__ childIdx, const uint8_t *__restrict__ nodeDesc, int nidLevel,
                                                     int *__restrict__ leafNode, int *__restrict__ levelNode) {
    const int lane = m_lane(), i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= n) return;
    const unsigned long long *q = (const unsigned long long *) (desc + (size_t) i * 32);
    const unsigned long long q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3];
    int node = 0, level = 0, nid = 0;
    for (;;) {
        const int c0 = childOff[node], nc = childOff[node + 1] - c0;
        if (nc <= 0) break;                       // leaf
        level++;
        unsigned long long best = ~0ull;
        for (int base = 0; base < nc; base += 64) {
            unsigned long long key = ~0ull;
            if (base + lane < nc) {
                const int id = childIdx[c0 + base + lane];
                const unsigned long long *d = (const unsigned long long *) (nodeDesc + (size_t) id * 32);
                const unsigned dist = (unsigned) (__popcll(q0 ^ d[0]) + __popcll(q1 ^ d[1]) + __popcll(q2 ^ d[2]) + __popcll(q3 ^ d[3]));
                key = ((unsigned long long) dist << 32) | (unsigned) (base + lane);
            }
            const unsigned long long m = ~wave_max_u64(~key);   // minimum
            best = m < best ? m : best;
        }
        node = childIdx[c0 + (int) (best & 0xFFFFFFFFu)];
        if (level == nidLevel) nid = node;
    }
    // a leaf above level L - levelsup (ragged tree) leaves *nid unassigned in the reference: defined here as that leaf
    if (nidLevel > 0 && level < nidLevel) nid = node;
    if (lane == 0) { leafNode[i] = node; levelNode[i] = nid; }
}

void launch_bow_descend(hipStream_t st, int n, const uint8_t *desc, const int *childOff, const int *childIdx, const uint8_t *nodeDesc, int nidLevel,
                        int *leafNode, int *levelNode) {
    if (n > 0) hipLaunchKernelGGL(k_bow_descend, dim3((n + 3) / 4), dim3(256), 0, st, n, desc, childOff, childIdx, nodeDesc, nidLevel, leafNode, levelNode);
}

static inline size_t al16(size_t b) { return (b + 15) & ~(size_t) 15; }

// ------------------------------------------------------------------------------------------------------------------
// Frame::AssignFeaturesToGrid + Frame::GetFeaturesInArea as a direct query (src/Frame.cc:314-330, 483-493, 424-481): the index list the
// reference returns for (x, y, r, minLevel, maxLevel), in its order -- grid columns ix ascending, cells iy ascending inside a column,
// keypoint indices ascending inside a cell (push_back order).  Every workgroup rebuilds the 64x48 grid of the frame in LDS (counting
// sort; the cells of one grid column are adjacent, so the cells [minCy, maxCy] of column ix are ONE contiguous run of the list) and
// serves a slice of the queries, one wave per query: 64 list entries per step, ordered append by ballot rank.
// ------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kMatchBlock) void k_features_in_area(FiaArgs A) {
    extern __shared__ __attribute__((aligned(16))) unsigned char dyn[];
    __shared__ int s_tmp[kMatchBlock / 64];
    int *cellStart = (int *) dyn;                    // GRID_CELLS + 1 (+ 3 pad)
    int *cellFill = cellStart + GRID_CELLS + 4;      // GRID_CELLS
    int *list = cellFill + GRID_CELLS;               // n
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = A.n;
    for (int i = tid; i < GRID_CELLS; i += kMatchBlock) cellFill[i] = 0;
    __syncthreads();
    for (int i = tid; i < n; i += kMatchBlock) {
        const ygzf_kp k = A.keys[i];
        const int px = (int) roundf((k.x - A.minX) * A.gridInvW);     // Frame::PosInGrid (round, as the reference)
        const int py = (int) roundf((k.y - A.minY) * A.gridInvH);
        if (!(px < 0 || px >= GRID_COLS || py < 0 || py >= GRID_ROWS)) atomicAdd(&cellFill[px * GRID_ROWS + py], 1);
    }
    __syncthreads();
    {   // exclusive scan of 3072 counts: 3 per thread
        const int per = GRID_CELLS / kMatchBlock;
        int sum = 0;
        for (int k = 0; k < per; k++) sum += cellFill[tid * per + k];
        const int incl = m_wave_incl_scan(sum);
        if (lane == 63) s_tmp[wave] = incl;
        __syncthreads();
        int woff = 0;
        for (int w2 = 0; w2 < wave; w2++) woff += s_tmp[w2];
        int off = woff + incl - sum;
        for (int k = 0; k < per; k++) {
            const int c = cellFill[tid * per + k];
            cellStart[tid * per + k] = off;
            off += c;
        }
        if (tid == kMatchBlock - 1) cellStart[GRID_CELLS] = off;
    }
    __syncthreads();
    for (int i = tid; i < GRID_CELLS; i += kMatchBlock) cellFill[i] = cellStart[i];
    __syncthreads();
    for (int i = tid; i < n; i += kMatchBlock) {
        const ygzf_kp k = A.keys[i];
        const int px = (int) roundf((k.x - A.minX) * A.gridInvW);
        const int py = (int) roundf((k.y - A.minY) * A.gridInvH);
        if (!(px < 0 || px >= GRID_COLS || py < 0 || py >= GRID_ROWS)) list[atomicAdd(&cellFill[px * GRID_ROWS + py], 1)] = i;
    }
    __syncthreads();
    for (int c = tid; c < GRID_CELLS; c += kMatchBlock) {  // cells keep ascending keypoint index (push_back order)
        const int s = cellStart[c], e = cellStart[c + 1];
        for (int a = s + 1; a < e; a++) {
            const int v = list[a];
            int b = a - 1;
            while (b >= s && list[b] > v) { list[b + 1] = list[b]; b--; }
            list[b + 1] = v;
        }
    }
    __syncthreads();
    const unsigned long long lt = (1ull << lane) - 1ull;
    const int wavesTotal = gridDim.x * (kMatchBlock / 64);
    for (int q = blockIdx.x * (kMatchBlock / 64) + wave; q < A.nq; q += wavesTotal) {
        const float x = A.xyr[3 * q], y = A.xyr[3 * q + 1], r = A.xyr[3 * q + 2];
        const int minLevel = A.levels ? A.levels[2 * q] : -1, maxLevel = A.levels ? A.levels[2 * q + 1] : -1;
        int *out = A.outIdx + (long long) q * A.cap;
        int total = 0;
        const int nMinCellX = max(0, (int) floorf((x - A.minX - r) * A.gridInvW));
        const int nMaxCellX = min(GRID_COLS - 1, (int) ceilf((x - A.minX + r) * A.gridInvW));
        const int nMinCellY = max(0, (int) floorf((y - A.minY - r) * A.gridInvH));
        const int nMaxCellY = min(GRID_ROWS - 1, (int) ceilf((y - A.minY + r) * A.gridInvH));
        if (!(nMinCellX >= GRID_COLS || nMaxCellX < 0 || nMinCellY >= GRID_ROWS || nMaxCellY < 0)) {
            const bool bCheckLevels = (minLevel > 0) || (maxLevel >= 0);
            for (int ix = nMinCellX; ix <= nMaxCellX; ix++) {
                if (nMinCellY > nMaxCellY) break;
                const int s = cellStart[ix * GRID_ROWS + nMinCellY], e = cellStart[ix * GRID_ROWS + nMaxCellY + 1];
                for (int base = s; base < e; base += 64) {
                    const int j = base + lane;
                    bool ok = false;
                    int idx = -1;
                    if (j < e) {
                        idx = list[j];
                        const ygzf_kp kp = A.keys[idx];
                        ok = true;
                        if (bCheckLevels) {
                            if (kp.octave < minLevel) ok = false;
                            if (maxLevel >= 0 && kp.octave > maxLevel) ok = false;
                        }
                        const float distx = kp.x - x, disty = kp.y - y;
                        if (!(fabsf(distx) < r && fabsf(disty) < r)) ok = false;
                    }
                    const unsigned long long m = __ballot(ok);
                    if (ok) {
                        const int pos = total + __popcll(m & lt);
                        if (pos < A.cap) out[pos] = idx;
                    }
                    total += __popcll(m);
                }
            }
        }
        if (lane == 0) A.outN[q] = total;
    }
}

size_t fia_lds_bytes(int n) { return sizeof(int) * ((size_t) GRID_CELLS + 4 + GRID_CELLS + (size_t) n) + 16; }

hipError_t launch_features_in_area(hipStream_t st, const FiaArgs &A) {
    const size_t lds = fia_lds_bytes(A.n);
    hipError_t e = hipFuncSetAttribute((const void *) k_features_in_area, hipFuncAttributeMaxDynamicSharedMemorySize, kMaxDynLds);
    if (e != hipSuccess) return e;
    const int perWg = kMatchBlock / 64;
    const int wgs = std::max(1, std::min(64, (A.nq + 4 * perWg - 1) / (4 * perWg)));   // >= 4 queries per wave before another workgroup rebuilds the grid
    hipLaunchKernelGGL(k_features_in_area, dim3(wgs), dim3(kMatchBlock), lds, st, A);
    return hipSuccess;
}

// LDS bytes / per-pair global spill bytes of the carve-up in k_match_last for a given plan
size_t match_lds_bytes(int capCur, int capLast, bool descInLds, int spill, size_t *spillBytes, bool specDeep) {
    size_t lds = al16(sizeof(int) * (GRID_CELLS + 1)) + al16(sizeof(int) * GRID_CELLS) + al16(sizeof(int) * (size_t) capCur) +
                 2 * al16(sizeof(float) * (size_t) capCur) + 2 * al16((size_t) capCur) + al16((size_t) capLast);
    size_t gl = 0;
    const size_t spec = (specDeep ? 2 : 1) * (al16(sizeof(uint4) * (size_t) capLast) + al16(sizeof(ushort4) * (size_t) capLast));
    const size_t misc = al16(sizeof(int) * (size_t) capLast) + al16(sizeof(float) * (size_t) capLast) + al16(sizeof(float) * (size_t) capCur) +
                        2 * al16(sizeof(int) * (size_t) capCur) + al16(2 * (size_t) capLast) + al16(sizeof(unsigned) * 8 * kExtSlots) +
                        al16(sizeof(unsigned short) * 8 * kExtSlots) + al16(sizeof(int) * kExtSlots);
    if (spill & kSpillSpec) gl += spec; else lds += spec;
    if (spill & kSpillMisc) gl += misc; else lds += misc;
    if (descInLds) lds += al16((size_t) 32 * capCur);
    if (spillBytes) *spillBytes = gl;
    return lds + 64;
}

hipError_t match_prepare(size_t ldsBytes) {
    return hipFuncSetAttribute((const void *) k_match_last, hipFuncAttributeMaxDynamicSharedMemorySize, kMaxDynLds);
}

void launch_backproject_unit(hipStream_t st, const ygzf_kp *keys, const int *cnt, long long kpStride, int maxKp, int nFrames, float fx,
                             float fy, float cx, float cy, float *world) {
    if (maxKp <= 0) return;
    hipLaunchKernelGGL(k_backproject_unit, dim3((maxKp + 255) / 256, nFrames), dim3(256), 0, st, keys, cnt, kpStride, fx, fy, cx, cy, world);
}

void launch_match_last(hipStream_t st, const MatchArgs &A, int nPairs, size_t ldsBytes) {
    hipLaunchKernelGGL(k_match_last, dim3(nPairs * (A.split > 1 ? A.split : 1)), dim3(kMatchBlock), ldsBytes, st, A);
}

}  // namespace ygzf
