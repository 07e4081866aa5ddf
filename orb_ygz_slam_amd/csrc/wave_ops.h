// wave_ops.h -- wave64 scans and reductions on DPP (data-parallel primitives: the operand of a VALU instruction is taken from another
// lane of the same row of 16, rows are combined through row_bcast / v_readlane).  A ds_bpermute (__shfl) step costs an LDS round trip,
// a DPP step one VALU slot: the octree's block scans and the per-keypoint moment sums were bound by exactly that latency.
// All callers run with the full wave active.
#ifndef YGZF_WAVE_OPS_H
#define YGZF_WAVE_OPS_H
#include <hip/hip_runtime.h>

namespace ygzf {

// quad_perm [1,0,3,2] = 0xb1, quad_perm [2,3,0,1] = 0x4e, row_ror:4 = 0x124, row_ror:8 = 0x128: after the four steps every lane holds the
// combination of its row of 16.
#define YGZF_ROW_REDUCE(v, OP)                                                         \
    v = OP(v, __builtin_amdgcn_update_dpp(0, v, 0xb1, 0xf, 0xf, false));               \
    v = OP(v, __builtin_amdgcn_update_dpp(0, v, 0x4e, 0xf, 0xf, false));               \
    v = OP(v, __builtin_amdgcn_update_dpp(0, v, 0x124, 0xf, 0xf, false));              \
    v = OP(v, __builtin_amdgcn_update_dpp(0, v, 0x128, 0xf, 0xf, false));

__device__ __forceinline__ int wops_add(int a, int b) { return a + b; }
__device__ __forceinline__ int wops_umax(int a, int b) { return (int) max((unsigned) a, (unsigned) b); }
__device__ __forceinline__ int wops_umin(int a, int b) { return (int) min((unsigned) a, (unsigned) b); }

// inclusive prefix sum over the 64 lanes (lanes a step cannot source add the identity through `old`)
__device__ __forceinline__ int wave_incl_scan(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);   // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);   // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);   // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);   // row_shr:8  -> inclusive inside every row of 16
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);   // row_bcast:15 into rows 1 and 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);   // row_bcast:31 into rows 2 and 3
    return v;
}
// the same scan on 64-bit values (three 21-bit counters packed into one word scan together)
__device__ __forceinline__ unsigned long long wave_incl_scan_u64(unsigned long long v) {
#define YGZF_SCAN64_STEP(CTRL, ROWMASK)                                                                                            \
    {                                                                                                                              \
        const unsigned lo = (unsigned) __builtin_amdgcn_update_dpp(0, (int) (unsigned) v, CTRL, ROWMASK, 0xf, false);              \
        const unsigned hi = (unsigned) __builtin_amdgcn_update_dpp(0, (int) (unsigned) (v >> 32), CTRL, ROWMASK, 0xf, false);      \
        v += ((unsigned long long) hi << 32) | lo;                                                                                 \
    }
    YGZF_SCAN64_STEP(0x111, 0xf) YGZF_SCAN64_STEP(0x112, 0xf) YGZF_SCAN64_STEP(0x114, 0xf) YGZF_SCAN64_STEP(0x118, 0xf)
    YGZF_SCAN64_STEP(0x142, 0xa) YGZF_SCAN64_STEP(0x143, 0xc)
#undef YGZF_SCAN64_STEP
    return v;
}
__device__ __forceinline__ int wave_sum(int v) {
    YGZF_ROW_REDUCE(v, wops_add)
    return (__builtin_amdgcn_readlane(v, 0) + __builtin_amdgcn_readlane(v, 16)) + (__builtin_amdgcn_readlane(v, 32) + __builtin_amdgcn_readlane(v, 48));
}
__device__ __forceinline__ unsigned wave_max_u32(unsigned u) {
    int v = (int) u;
    YGZF_ROW_REDUCE(v, wops_umax)
    return max(max((unsigned) __builtin_amdgcn_readlane(v, 0), (unsigned) __builtin_amdgcn_readlane(v, 16)),
               max((unsigned) __builtin_amdgcn_readlane(v, 32), (unsigned) __builtin_amdgcn_readlane(v, 48)));
}
__device__ __forceinline__ unsigned wave_min_u32(unsigned u) {
    int v = (int) u;
    YGZF_ROW_REDUCE(v, wops_umin)
    return min(min((unsigned) __builtin_amdgcn_readlane(v, 0), (unsigned) __builtin_amdgcn_readlane(v, 16)),
               min((unsigned) __builtin_amdgcn_readlane(v, 32), (unsigned) __builtin_amdgcn_readlane(v, 48)));
}
// 64-bit maximum: both halves travel through the same DPP steps, the comparison is on the pair
__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long v) {
#define YGZF_U64_STEP(CTRL)                                                                                                        \
    {                                                                                                                              \
        const unsigned lo = (unsigned) __builtin_amdgcn_update_dpp(0, (int) (unsigned) v, CTRL, 0xf, 0xf, false);                  \
        const unsigned hi = (unsigned) __builtin_amdgcn_update_dpp(0, (int) (unsigned) (v >> 32), CTRL, 0xf, 0xf, false);          \
        const unsigned long long t = ((unsigned long long) hi << 32) | lo;                                                         \
        v = t > v ? t : v;                                                                                                         \
    }
    YGZF_U64_STEP(0xb1) YGZF_U64_STEP(0x4e) YGZF_U64_STEP(0x124) YGZF_U64_STEP(0x128)
#undef YGZF_U64_STEP
    unsigned long long r = 0;
#pragma unroll
    for (int l = 0; l < 64; l += 16) {
        const unsigned long long t = ((unsigned long long) (unsigned) __builtin_amdgcn_readlane((int) (unsigned) (v >> 32), l) << 32) |
                                     (unsigned) __builtin_amdgcn_readlane((int) (unsigned) v, l);
        r = t > r ? t : r;
    }
    return r;
}

}  // namespace ygzf
#endif
