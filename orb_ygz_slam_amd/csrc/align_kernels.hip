// align_kernels.hip -- SVO-style sparse photometric alignment on gfx950 (product code).
//
//   k_sia_run   ygz::SparseImgAlign::run(Frame *ref, Frame *cur, SE3f &TCR)      reference src/SparseImageAlign.cc:20-49
//               precomputeReferencePatches :57-128, computeResiduals :130-231, solve :233-238, update :240-244,
//               NLLSSolver::optimizeGaussNewton include/NLSSolver_impl.hpp:17-91, Sophus SE3f exp / * / inverse
//               (Thirdparty/sophus/sophus/se3.hpp:159-171,267-271,406-428, so3.hpp:425-456)
//
// The problem is latency-bound: <= n_iter x (max_level - min_level + 1) dependent Gauss-Newton iterations
// over N x 16 pixels.  One persistent workgroup per (ref, cur) pair runs every level and every iteration on the device:
// per-thread partial normal equations (21 + 6 + 2 values), a fixed-shape reduction tree (wave shuffles, then LDS across
// waves -> bit-reproducible run to run), the 6x6 pivoted LDL^T solve and the SE3 update on lane 0, no host round trip.
// fp32 throughout, like the reference (Eigen float / Sophus float); parity with the CPU oracle is graded at 1e-5 on the
// SE3 output because the summation order differs from the reference's sequential loop.
#include <type_traits>

#include "kernels.h"
#include "se3_device.h"

namespace ygzf {

constexpr int kSiaBlock = 512;
// x = H.ldlt().solve(b): LDL^T with diagonal pivoting, 6x6 float (single lane).  Every array index is a compile-time
// constant after unrolling (the pivot swap is a chain of predicated exchanges), so A/b/perm stay in registers; with
// run-time indices they land in scratch memory and each access costs a memory round trip.
__device__ __forceinline__ void ldlt_solve6(const float Hin[36], const float bin[6], float x[6]) {
    float A[36], b[6];
    int perm[6];
#pragma unroll
    for (int i = 0; i < 36; i++) A[i] = Hin[i];
#pragma unroll
    for (int i = 0; i < 6; i++) { b[i] = bin[i]; perm[i] = i; }
#pragma unroll
    for (int k = 0; k < 6; k++) {
        int p = k;
        float best = fabsf(A[7 * k]);
#pragma unroll
        for (int i = k + 1; i < 6; i++)
            if (fabsf(A[7 * i]) > best) { best = fabsf(A[7 * i]); p = i; }
#pragma unroll
        for (int i = k + 1; i < 6; i++) {
            if (p == i) {   // exchange rows / columns k and i
#pragma unroll
                for (int j = 0; j < 6; j++) { const float t = A[6 * k + j]; A[6 * k + j] = A[6 * i + j]; A[6 * i + j] = t; }
#pragma unroll
                for (int j = 0; j < 6; j++) { const float t = A[6 * j + k]; A[6 * j + k] = A[6 * j + i]; A[6 * j + i] = t; }
                const float t = b[k]; b[k] = b[i]; b[i] = t;
                const int ti = perm[k]; perm[k] = perm[i]; perm[i] = ti;
            }
        }
        const float d = A[7 * k];
#pragma unroll
        for (int i = k + 1; i < 6; i++) {
            const float l = A[6 * i + k] / d;
#pragma unroll
            for (int j = k + 1; j < 6; j++) A[6 * i + j] -= l * A[6 * k + j];
            A[6 * i + k] = l;
        }
    }
    float y[6], z[6];
#pragma unroll
    for (int i = 0; i < 6; i++) {
        float s = b[i];
#pragma unroll
        for (int j = 0; j < i; j++) s -= A[6 * i + j] * y[j];
        y[i] = s;
    }
#pragma unroll
    for (int i = 0; i < 6; i++) y[i] = y[i] / A[7 * i];
#pragma unroll
    for (int i = 5; i >= 0; i--) {
        float s = y[i];
#pragma unroll
        for (int j = i + 1; j < 6; j++) s -= A[6 * j + i] * z[j];
        z[i] = s;
    }
#pragma unroll
    for (int i = 0; i < 6; i++) {
#pragma unroll
        for (int q = 0; q < 6; q++)
            if (perm[i] == q) x[q] = z[i];
    }
}

// Eight image bytes starting at column c0 of a row as one (unaligned) 8-byte load: scattered single-byte loads were the limit of the patch
// loops (every lane of a load instruction touches its own cache line; the texture path serialises them).  The load never leaves the row:
// it starts at min(c0, w - 8) and the result is shifted so that byte 0 is column c0 (callers need at most 8 - shift bytes).
__device__ __forceinline__ unsigned long long row_bytes8(const uint8_t *row, int c0, int w) {
    if (w < 8) {
        unsigned long long v = 0;
        for (int k = 0; k < 8 && c0 + k < w; k++) v |= (unsigned long long) row[c0 + k] << (8 * k);
        return v;
    }
    const int s0 = min(c0, w - 8);
    unsigned long long v;
    __builtin_memcpy(&v, row + s0, 8);
    // shift <= 3 bytes for every caller (patches stay 3 pixels inside the image): 32-bit byte alignment instead of a 64-bit shift
    const unsigned sh = (unsigned) (c0 - s0), lo = (unsigned) v, hi = (unsigned) (v >> 32);
    return ((unsigned long long) (hi >> (8 * sh)) << 32) | __builtin_amdgcn_alignbyte(hi, lo, sh);
}
__device__ __forceinline__ float byte_f(unsigned long long v, int k) {   // k is a compile-time constant at every call site
    const unsigned w = k < 4 ? (unsigned) v : (unsigned) (v >> 32);
    return (float) ((w >> (8 * (k & 3))) & 0xFFu);
}

// The terms of JacobXYZ2Cam (include/SparseImageAlign.h:90-111) that depend on the reference-frame point only, with the reference's
// operations: ja = (1/z, x/z^2, x*y/z^2, -(1 + x^2/z^2)), jb = (y/z, y/z^2, 1 + y^2/z^2, -x/z)
__device__ __forceinline__ void jac_terms(const float xr[3], float4 &ja, float4 &jb) {
    const float x = xr[0], yy = xr[1];
    const float z_inv = (float) (1. / (double) xr[2]);
    const float z_inv_2 = z_inv * z_inv;
    const float J2 = x * z_inv_2, J8 = yy * z_inv_2;
    ja = make_float4(z_inv, J2, yy * J2, (float) -(1.0 + (double) (x * J2)));
    jb = make_float4(yy * z_inv, J8, (float) (1.0 + (double) (yy * J8)), -x * z_inv);
}

constexpr int kAcc = 30;  // 21 upper-triangular H entries + 6 b + chi2 + n_meas + n_visible_features

// The wave totals of the 30 accumulators, written to out[0..29].  Halving tree instead of 30 full butterflies: v_permlane32_swap exchanges the
// upper half of one register with the lower half of another, so ONE add folds lanes l and l + 32 of TWO accumulators (lanes < 32 keep the
// first, lanes >= 32 the second); v_permlane16_swap does the same with rows of 16 -- after the two stages 8 registers hold, per row of 16
// lanes, the column sums of a different accumulator (rows 0..3 of register n: accumulators 4n, 4n+2, 4n+1, 4n+3), and four DPP steps
// inside the rows finish: 86 instructions instead of 210.  Fixed tree -> bit-reproducible run to run.
__device__ __forceinline__ void wave_sums_30(const float (&acc)[kAcc], float *out, int lane) {
    float A[16];
#pragma unroll
    for (int m = 0; m < 15; m++) {
        float a = acc[2 * m], b = acc[2 * m + 1];
        // (inline asm: with this compiler the two results of __builtin_amdgcn_permlane32_swap collapse into one when both feed an add;
        // the s_nop covers the VALU-write -> permlane-swap-read wait states the compiler would insert for the builtin)
        asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
        A[m] = a + b;
    }
    A[15] = 0.f;
#pragma unroll
    for (int n = 0; n < 8; n++) {
        float a = A[2 * n], b = A[2 * n + 1];
        asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
        float v = a + b;
        v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xb1, 0xf, 0xf, false));   // quad_perm [1,0,3,2]
        v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4e, 0xf, 0xf, false));   // quad_perm [2,3,0,1]
        v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xf, 0xf, false));  // row_ror:4
        v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xf, 0xf, false));  // row_ror:8
        const int row = lane >> 4;
        const int k = 4 * n + ((row & 1) << 1) + (row >> 1);
        if ((lane & 15) == 0 && k < kAcc) out[k] = v;
    }
}

// ---- the 6x6 solve and the SE3 update on the lanes of one wave --------------------------------------------------------------------
// lane_get: a value of another lane (per-lane source, LDS crossbar); lane_bcast: the value of one lane, wave-uniform (source lane uniform)
__device__ __forceinline__ float lane_get(float v, int srcLane) {
    return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(srcLane << 2, __builtin_bit_cast(int, v)));
}
__device__ __forceinline__ float lane_bcast(float v, int srcLane) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), srcLane));
}

// x = H.ldlt().solve(b) with ldlt_solve6's arithmetic -- the same operations on the same operands in the same order, element by element --
// spread over a wave: lane 8 i + j holds A[i][j] (j < 6), b[i] (j = 6) and the permutation entry perm[i] (j = 7).  Per elimination step the
// pivot search reads the remaining diagonal through readlane (uniform), the row / column exchange is ONE permutation of the lanes (the
// one-lane form spends 28 predicated moves per candidate row on it), every lane forms its own l = A[i][k] / d and updates its own element.
// The right-hand side rides along as column 6 (b[i] -= l_i b[k] at step k is the forward substitution's j = k term).  What is left -- 6
// divisions by D, the 15-term backward substitution, the un-permutation -- runs on uniform values.  Returns x[q] in lane q (q < 6).
// All 64 lanes of the wave must be active.  52 us -> see DESIGN.md for the measured gain.
template <int k>
__device__ __forceinline__ float ldlt_step_wave(float a, int i, int j) {
    if (k < 5) {
        int p = k;
        float best = fabsf(lane_bcast(a, 9 * k));
#pragma unroll
        for (int r = k + 1; r < 6; r++) {
            const float v = fabsf(lane_bcast(a, 9 * r));
            if (v > best) { best = v; p = r; }
        }
        if (p != k) {   // uniform: exchange rows / columns k and p (columns 6 and 7 follow their rows)
            const int si = i == k ? p : (i == p ? k : i);
            const int sj = j == k ? p : (j == p ? k : j);
            a = lane_get(a, 8 * si + sj);
        }
    }
    const float d = lane_bcast(a, 9 * k);
    // A[i][k]: lane k of the own group of eight -- two DPP row broadcasts (a DPP row of 16 lanes holds matrix rows 2r and 2r + 1), so the
    // division starts at once and covers the LDS-crossbar latency of the A[k][j] fetch
    const float akj = lane_get(a, 8 * k + j);
    const int ai = __builtin_bit_cast(int, a);
    const float e0 = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, ai, 0x150 + k, 0xf, 0xf, false));       // row_newbcast:k
    const float e1 = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, ai, 0x150 + 8 + k, 0xf, 0xf, false));   // row_newbcast:8+k
    const float aik = (i & 1) ? e1 : e0;
    const float l = aik / d;
    if (i > k && i < 6) {
        if (j > k && j <= 6) a -= l * akj;
        else if (j == k) a = l;
    }
    return a;
}

__device__ __forceinline__ float ldlt_solve6_wave(float a, int lane) {
    const int i = lane >> 3, j = lane & 7;
    if (j == 7) a = (float) i;   // perm[i] = i
    a = ldlt_step_wave<0>(a, i, j);
    a = ldlt_step_wave<1>(a, i, j);
    a = ldlt_step_wave<2>(a, i, j);
    a = ldlt_step_wave<3>(a, i, j);
    a = ldlt_step_wave<4>(a, i, j);
    a = ldlt_step_wave<5>(a, i, j);
    // lane (i, i): D[i]; lane (i, j < i): L[i][j]; lane (i, 6): the forward-substituted right-hand side; lane (i, 7): perm[i]
    const float q = a / lane_get(a, 9 * i);
    float y[6], z[6];
#pragma unroll
    for (int r = 0; r < 6; r++) y[r] = lane_bcast(q, 8 * r + 6);
#pragma unroll
    for (int r = 5; r >= 0; r--) {
        float sacc = y[r];
#pragma unroll
        for (int c = r + 1; c < 6; c++) sacc -= lane_bcast(a, 8 * c + r) * z[c];
        z[r] = sacc;
    }
    float x = 0.f;
#pragma unroll
    for (int r = 0; r < 6; r++)
        if ((int) lane_bcast(a, 8 * r + 7) == lane) x = z[r];
    return x;
}

// se3_exp (se3_device.h) on a wave, uniform result: the four sinf / cosf evaluations -- most of its instructions -- become two, lane 0
// working on theta / 2 and lane 1 on theta (the same functions on the same arguments as the one-lane form: the same bits).
__device__ __forceinline__ Se3 se3_exp_wave(const float a[6], int lane) {
    const float *om = a + 3;
    const float theta_sq = om[0] * om[0] + om[1] * om[1] + om[2] * om[2];
    const float theta = sqrtf(theta_sq);
    const float half_theta = 0.5f * theta;
    const bool small_angle = theta < kSophusEps;
    float sin_half = 0.f, cos_half = 0.f, sin_th = 0.f, cos_th = 0.f;
    if (!small_angle) {
        const float arg = (lane & 1) ? theta : half_theta;
        const float sv = sinf(arg), cv = cosf(arg);
        sin_half = lane_bcast(sv, 0); cos_half = lane_bcast(cv, 0);
        sin_th = lane_bcast(sv, 1); cos_th = lane_bcast(cv, 1);
    }
    float imag_factor, real_factor;
    if (small_angle) {
        const float theta_po4 = theta_sq * theta_sq;
        imag_factor = 0.5f - (float) (1.0 / 48.0) * theta_sq + (float) (1.0 / 3840.0) * theta_po4;
        real_factor = 1.f - 0.5f * theta_sq + (float) (1.0 / 384.0) * theta_po4;
    } else {
        imag_factor = sin_half / theta;
        real_factor = cos_half;
    }
    Se3 r;
    r.q[3] = real_factor;
    r.q[0] = imag_factor * om[0]; r.q[1] = imag_factor * om[1]; r.q[2] = imag_factor * om[2];
    quat_normalize(r.q);
    const float O[9] = {0, -om[2], om[1], om[2], 0, -om[0], -om[1], om[0], 0};
    float O2[9];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) O2[3 * i + j] = O[3 * i] * O[j] + O[3 * i + 1] * O[3 + j] + O[3 * i + 2] * O[6 + j];
    float V[9];
    if (small_angle) {
        quat_to_R(r.q, V);
    } else {
        const float theta_sq2 = theta * theta;   // se3.hpp:419: the square of the root, not the squared norm above
        const float c1 = (1.f - cos_th) / theta_sq2;
        const float c2 = (theta - sin_th) / (theta_sq2 * theta);
        for (int i = 0; i < 9; i++) V[i] = ((i % 4 == 0) ? 1.f : 0.f) + c1 * O[i] + c2 * O2[i];
    }
    for (int i = 0; i < 3; i++) r.t[i] = V[3 * i] * a[0] + V[3 * i + 1] * a[1] + V[3 * i + 2] * a[2];
    return r;
}

// precomputeReferencePatches for ONE feature at one level (src/SparseImageAlign.cc:57-128): false (V, mom untouched) when the patch leaves the
// level's border.  What the reference caches per patch pixel -- the interpolated value and the central differences dx, dy of the interpolated
// image -- are all differences of ONE 6 x 6 grid of interpolated values V[6 r + c] = I_ref(u - 3 + c + su, v - 3 + r + sv):
//     patch(y, x) = V[y+1][x+1],   dx(y, x) = 0.5 (V[y+1][x+2] - V[y+1][x]),   dy(y, x) = 0.5 (V[y+2][x+1] - V[y][x+1])
// (the reference writes each of them out as w_tl p[..] + w_tr p[..] + w_bl p[..] + w_br p[..] over the four bytes around the position: the same
// expression on the same operands wherever two of them name the same position, so the grid holds the same floats).  32 values (the corners are
// never used) instead of 48: what makes a thread's features fit its registers (k_sia_run).  mom = (sum dx dx, sum dx dy, sum dy dy, 0) over the 16
// pixels.  The arithmetic -- which products are fused included -- is part of the aligner's definition (oracle/oracle_align.cpp's device-order mode
// repeats it): k_sia_run and k_sia_precompute share this one body.
__device__ __forceinline__ bool sia_ref_patch(const SiaLevel &Lr, float scale, float2 kp, float (&V)[36], float4 &mom) {
    const int border = 3;                       // patch_halfsize_ + 1
    const float u_ref = kp.x * scale, v_ref = kp.y * scale;
    const int u_ref_i = (int) floorf(u_ref), v_ref_i = (int) floorf(v_ref);
    if (u_ref_i - border < 0 || v_ref_i - border < 0 || u_ref_i + border >= Lr.w || v_ref_i + border >= Lr.h) return false;
    const float su = u_ref - u_ref_i, sv = v_ref - v_ref_i;
    // the reference forms these in double and rounds to float: with u, v >= 3 the fractions are multiples of 2^-22, so 1 - su and
    // 1 - sv are exact in fp32 and the fp32 product is the same single rounding of the same exact product
    // (tests/test_kernel_models.py::test_bilinear_weights_fp32)
    const float usu = 1.f - su, usv = 1.f - sv;
    const float w_tl = usu * usv, w_tr = su * usv, w_bl = usu * sv, w_br = su * sv;
    const int st = Lr.pitch;
    // columns u-3 .. u+4 of rows v-3 .. v+3: seven 8-byte row loads, all in flight together
    const uint8_t *rt = Lr.img + (long long) (v_ref_i - 3) * st;
    unsigned long long R[7];
#pragma unroll
    for (int r = 0; r < 7; r++) R[r] = row_bytes8(rt + (long long) r * st, u_ref_i - 3, Lr.w);
#pragma unroll
    for (int r = 0; r < 6; r++)
#pragma unroll
        for (int c = 0; c < 6; c++) {
            const bool corner = (r == 0 || r == 5) && (c == 0 || c == 5);
            V[6 * r + c] = corner ? 0.f : w_tl * byte_f(R[r], c) + w_tr * byte_f(R[r], c + 1) + w_bl * byte_f(R[r + 1], c) + w_br * byte_f(R[r + 1], c + 1);
        }
    // (explicit fused multiply-adds, here and in the accumulate phase: the library is built -ffp-contract=off, and WHICH products are
    // fused is part of this kernel's definition -- the device-order oracle repeats them with fmaf and must reproduce H, b and chi2 bit
    // for bit, tests/test_gpu_align.py)
    float sxx = 0.f, sxy = 0.f, syy = 0.f;
#pragma unroll
    for (int y = 0; y < 4; y++)
#pragma unroll
        for (int x = 0; x < 4; x++) {
            const float dx = 0.5f * (V[6 * (y + 1) + x + 2] - V[6 * (y + 1) + x]), dy = 0.5f * (V[6 * (y + 2) + x + 1] - V[6 * y + x + 1]);
            sxx = __builtin_fmaf(dx, dx, sxx);
            sxy = __builtin_fmaf(dx, dy, sxy);
            syy = __builtin_fmaf(dy, dy, syy);
        }
    mom = make_float4(sxx, sxy, syy, 0.f);
    return true;
}

// The reference patches of EVERY level before the alignment starts (SiaArgs::perLevel): they depend on the reference frame only, so all
// (pair, level, feature) items are independent and run chip-wide -- inside k_sia_run they were 71 of its 355 us (seven times 1008 features
// on the one workgroup of the pair, each a chain of global loads).  Per level a cache of its own (level l at + (l - minLevel) * stride: nine
// planes of float4 = the 6 x 6 grid of sia_ref_patch, feature i at [plane * kpStride + i]; the moments as a float4 of their own) and a
// flag "visible at this level or a coarser one" (visible_fts_ is cumulative, src/SparseImageAlign.cc:41-45).  A feature that is visible from
// a coarser level but leaves THIS level's border keeps, in the reference, the patch of the last level that held it (the patch cache is
// never cleared) under a zeroed Jacobian: that level's grid is rebuilt here and the moments carry the mark "gradients are zero" (w = 1).
__global__ __launch_bounds__(256) void k_sia_precompute(SiaArgs A) {
    const int pair = blockIdx.z, level = A.maxLevel - (int) blockIdx.y;
    const int N = A.nRef ? A.nRef[pair] : A.n;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N || level < A.minLevel) return;
    const long long po = (long long) pair * A.kpStride;
    const uint8_t *mpValid = A.mpValid ? A.mpValid + po : nullptr;
    const uint8_t *outlier = A.outlier ? A.outlier + po : nullptr;
    const bool excluded = (mpValid && !mpValid[i]) || (outlier && outlier[i]);
    const ygzf_kp k = A.keys[po + i];
    const float2 kp = excluded ? make_float2(-100.f, -100.f) : make_float2(k.x, k.y);
    const size_t plane = (size_t) A.kpStride;
    const int li = level - A.minLevel;
    float4 *rc = (float4 *) (A.patchCache + (size_t) li * A.pcLevelStride + (size_t) po * 48) + i;
    float4 *mom = (float4 *) (A.momCache + (size_t) li * A.momLevelStride) + po + i;
    uint8_t *flag = A.levelFlags + (size_t) li * A.flagLevelStride + po + i;
    const SiaLevel *refLv = A.refLv + (long long) pair * A.lvStride;
    float V[36];
    float4 m = make_float4(0.f, 0.f, 0.f, 0.f);
    int src = level;
    bool ok = sia_ref_patch(refLv[level], A.invScale[level], kp, V, m);
    if (!ok)
        for (src = level + 1; src <= A.maxLevel; src++)
            if (sia_ref_patch(refLv[src], A.invScale[src], kp, V, m)) { ok = true; break; }
    *flag = ok ? 1 : 0;
    if (level == A.minLevel) A.visible[po + i] = ok ? 1 : 0;
    if (!ok) return;
    if (src != level) m = make_float4(0.f, 0.f, 0.f, 1.f);
#pragma unroll
    for (int j = 0; j < 9; j++) rc[j * plane] = make_float4(V[4 * j], V[4 * j + 1], V[4 * j + 2], V[4 * j + 3]);
    *mom = m;
}

// One feature's share of an iteration's normal equations (computeResiduals, src/SparseImageAlign.cc:130-231): projection into the current
// level, the 5 x 5 bytes under the warped patch, residuals against the reference patch, H / Jres / chi2 into the thread's accumulators.
// V: the feature's 6 x 6 reference grid (sia_ref_patch), S: its gradient moments (w != 0: gradients are zero at this level).
__device__ __forceinline__ void sia_accumulate_feature(const SiaArgs &A, const Se3 &T, const SiaLevel &Lc, float scale, bool staged, const uint8_t *s_cur,
                                                       const float4 ft, const float4 *jacp, const float (&V)[36], const float4 S, float (&acc)[kAcc]) {
    const int border = 3;                       // patch_halfsize_ + 1
        if (ft.w == 0.f) return;
    const float xr[3] = {ft.x, ft.y, ft.z};
    float xc[3];
    se3_act(T, xr, xc);
    const float ucx = A.fx * xc[0] / xc[2] + A.cx, ucy = A.fy * xc[1] / xc[2] + A.cy;   // Frame::Camera2Pixel
    const float u_cur = ucx * scale, v_cur = ucy * scale;
    const int ui = (int) floorf(u_cur), vi = (int) floorf(v_cur);
    if (ui < 0 || vi < 0 || ui - border < 0 || vi - border < 0 || ui + border >= Lc.w || vi + border >= Lc.h) return;
    acc[29] += 1.f;
    const float su = u_cur - ui, sv = v_cur - vi;
    // the reference forms these in double and rounds to float: with u, v >= 3 the fractions are multiples of 2^-22, so 1 - su and
    // 1 - sv are exact in fp32 and the fp32 product is the same single rounding of the same exact product
    // (tests/test_kernel_models.py::test_bilinear_weights_fp32)
    const float usu = 1.f - su, usv = 1.f - sv;
    const float w_tl = usu * usv, w_tr = su * usv, w_bl = usu * sv, w_br = su * sv;
    const int st = Lc.pitch;
    const uint8_t *p = Lc.img + (long long) (vi - 2) * st + (ui - 2);
    // rows vi-2 .. vi+2, columns ui-2 .. ui+2.  Each row is fetched as the two aligned dwords around its first byte (the patch
    // stays 3 pixels inside the image, so both dwords lie inside the level) and byte-aligned in registers.
    unsigned rlo[5], rhi[5];
    if (staged) {
        const unsigned o0 = (unsigned) (vi - 2) * (unsigned) st + (unsigned) (ui - 2);
#pragma unroll
        for (int r = 0; r < 5; r++) {
            const uint8_t *q = s_cur + o0 + (unsigned) r * (unsigned) st;
            const unsigned sh = (unsigned) (unsigned long long) q & 3u;   // LDS addresses: the low bits of the generic pointer are the LDS offset's
            const unsigned *qa = (const unsigned *) (q - sh);
            const unsigned v0 = qa[0], v1 = qa[1];
            rlo[r] = __builtin_amdgcn_alignbyte(v1, v0, sh);
            rhi[r] = v1 >> (8 * sh);
        }
    } else {
#pragma unroll
        for (int r = 0; r < 5; r++) {
            const uint8_t *q = p + (long long) r * st;
            const unsigned sh = (unsigned) (unsigned long long) q & 3u;
            const uint2 v = *(const uint2 *) __builtin_assume_aligned(q - sh, 4);
            rlo[r] = __builtin_amdgcn_alignbyte(v.y, v.x, sh);
            rhi[r] = v.y >> (8 * sh);
        }
    }
    float tf[5][5];
#pragma unroll
    for (int r = 0; r < 5; r++) {
#pragma unroll
        for (int k = 0; k < 4; k++) tf[r][k] = (float) ((rlo[r] >> (8 * k)) & 0xFFu);
        tf[r][4] = (float) (rhi[r] & 0xFFu);
    }
    // JacobXYZ2Cam (include/SparseImageAlign.h:90-111) of the reference-frame point: the cached quantity of the reference,
    // (dx*J.row0 + dy*J.row1) * (fx*scale), is re-evaluated with the same operations, so only patch/dx/dy (12 B per pixel
    // instead of 28 B) stream from memory every iteration.  The point-only terms come from LDS when they fit (staged once per run).
    float J[12];
    {
        float4 ja, jb;
        if (A.jacLds) { ja = jacp[0]; jb = jacp[1]; }
        else jac_terms(xr, ja, jb);
        J[0] = -ja.x; J[1] = 0.f; J[2] = ja.y; J[3] = ja.z; J[4] = ja.w; J[5] = jb.x;
        J[6] = 0.f; J[7] = -ja.x; J[8] = jb.y; J[9] = jb.z; J[10] = -ja.z; J[11] = jb.w;
    }
    const float fs = A.fx * scale;
    float Jf[12];   // J * (fx * scale): folded once per feature (the reference scales every (dx*J0 + dy*J1) row; see the note below)
#pragma unroll
    for (int k = 0; k < 12; k++) Jf[k] = J[k] * fs;
    acc[28] += 16.f;   // n_meas: one per patch pixel (exact in fp32)
    // The normal equations are sums of thousands of products: their rounding is not part of the reference's definition (its own
    // result moves in the 7th digit when features are summed in another order, tests/test_gpu_fuzz.py) and the SE3 is graded at
    // 1e-5.  The pixel Jacobian of the reference is J_p = dx_p * Jf0 + dy_p * Jf1 with the SAME two 6-vectors for the 16 pixels of
    // a feature, so the feature's share of H = sum_p J_p J_p^T is
    //     Sxx * Jf0 Jf0^T + Sxy * (Jf0 Jf1^T + Jf1 Jf0^T) + Syy * Jf1 Jf1^T,       Sxx = sum dx^2, Sxy = sum dx*dy, Syy = sum dy^2
    // -- moments of the reference patch, constant over the iterations of a level (precompute above) -- and its share of
    // Jres = -sum_p J_p res_p is -(Jf0 * sum dx*res + Jf1 * sum dy*res).  Per pixel that leaves the interpolation, the residual and
    // three multiply-adds (8 instructions instead of ~45); per feature 66 for H and 12 for Jres.  This block -- and only this
    // block -- uses fused multiply-adds (v_fma_f32, written out: the accumulate phase is issue-bound at two waves per SIMD).
    {
        float sxr = 0.f, syr = 0.f;
        // gradients marked zero (a feature that left this level's border but is visible from a coarser one): hz = 0 makes dx = dy = +-0, the sums
        // keep their +0 -- what the reference's zeroed Jacobian rows add
        const float hz = S.w != 0.f ? 0.f : 0.5f;
#pragma unroll
        for (int y = 0; y < 4; y++) {
#pragma unroll
            for (int x = 0; x < 4; x++) {
                const float I = __builtin_fmaf(w_br, tf[y + 1][x + 1], __builtin_fmaf(w_bl, tf[y + 1][x], __builtin_fmaf(w_tr, tf[y][x + 1], w_tl * tf[y][x])));
                const float res = I - V[6 * (y + 1) + x + 1];
                const float dxa = hz * (V[6 * (y + 1) + x + 2] - V[6 * (y + 1) + x]), dya = hz * (V[6 * (y + 2) + x + 1] - V[6 * y + x + 1]);
                acc[27] = __builtin_fmaf(res, res, acc[27]);
                sxr = __builtin_fmaf(dxa, res, sxr);
                syr = __builtin_fmaf(dya, res, syr);
            }
        }
        // Jf[1] and Jf[6] are structural zeros of JacobXYZ2Cam (their products are exact zeros in the reference's sums): left out
        float P[6], Q[6];
        P[0] = S.x * Jf[0]; Q[0] = S.y * Jf[0];
        P[1] = S.y * Jf[7]; Q[1] = S.z * Jf[7];
#pragma unroll
        for (int k = 2; k < 6; k++) {
            P[k] = __builtin_fmaf(S.y, Jf[6 + k], S.x * Jf[k]);
            Q[k] = __builtin_fmaf(S.z, Jf[6 + k], S.y * Jf[k]);
        }
        // H[a][b] += Jf0[a] * P[b] + Jf1[a] * Q[b], upper triangle in the accumulators' order
#pragma unroll
        for (int b2 = 0; b2 < 6; b2++) acc[b2] = __builtin_fmaf(Jf[0], P[b2], acc[b2]);
#pragma unroll
        for (int b2 = 1; b2 < 6; b2++) acc[5 + b2] = __builtin_fmaf(Jf[7], Q[b2], acc[5 + b2]);
        int t = 11;
#pragma unroll
        for (int a = 2; a < 6; a++)
#pragma unroll
            for (int b2 = a; b2 < 6; b2++, t++) {
                acc[t] = __builtin_fmaf(Jf[a], P[b2], acc[t]);
                acc[t] = __builtin_fmaf(Jf[6 + a], Q[b2], acc[t]);
            }
        acc[21] = __builtin_fmaf(-Jf[0], sxr, acc[21]);
        acc[22] = __builtin_fmaf(-Jf[7], syr, acc[22]);
#pragma unroll
        for (int k = 2; k < 6; k++) {
            acc[21 + k] = __builtin_fmaf(-Jf[k], sxr, acc[21 + k]);
            acc[21 + k] = __builtin_fmaf(-Jf[6 + k], syr, acc[21 + k]);
        }
    }
}

// DBG: phase clocks (YGZF_SIA_DEBUG) are compiled in only in the instrumented instantiation; the production kernel reads no clock
template <bool DBG>
__global__ __launch_bounds__(kSiaBlock) void k_sia_run(SiaArgs A) {
    extern __shared__ float4 s_feat[];   // per feature: xyz in the reference camera (Tref * Xw, constant over the run), w = visible flag;
                                         // behind it float2 s_uv[]: the reference keypoint (level 0), x = -100 for excluded features
    __shared__ float s_red[(kSiaBlock / 16) * kAcc];   // one partial per row of 16 lanes
    __shared__ float s_tot[kAcc];
    __shared__ Se3 s_T, s_Told, s_Tref;
    __shared__ float s_H[36];
    __shared__ float s_chi2, s_newchi2;
    __shared__ int s_stop, s_break, s_nmeas, s_iters;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int pair = blockIdx.x;
    const int N = A.nRef ? A.nRef[pair] : A.n;
    const ygzf_kp *keys = A.keys + (long long) pair * A.kpStride;
    const float *world = A.world + (long long) pair * A.kpStride * 3;
    const uint8_t *mpValid = A.mpValid ? A.mpValid + (long long) pair * A.kpStride : nullptr;
    const uint8_t *outlier = A.outlier ? A.outlier + (long long) pair * A.kpStride : nullptr;
    // reference patch cache of this pair: 12 planes of float4, plane 3 * row + {0 patch, 1 dx, 2 dy}, feature i at [plane * kpStride + i]
    // (a lane works on one feature: consecutive lanes read consecutive float4s of a plane)
    // (SiaArgs::perLevel: one such cache per level, filled by k_sia_precompute before this kernel starts; the level loop below re-points both)
    float4 *rowCache = (float4 *) (A.patchCache + (long long) pair * A.kpStride * 48);
    const size_t plane = (size_t) A.kpStride;
    uint8_t *visible = A.visible + (long long) pair * A.kpStride;
    // gradient moments of the reference patch, per feature and level: (sum dx*dx, sum dx*dy, sum dy*dy, -) over its 16 pixels
    float4 *mom = (float4 *) A.momCache + (long long) pair * A.kpStride;
    float *out = A.out + (long long) pair * 48;   // TCR[7], ret, iters, chi2, pad[2], H[36]

    if (tid == 0) {
        Se3 Tr, Tc;
        for (int i = 0; i < 4; i++) { Tr.q[i] = A.poses[pair * 14 + i]; Tc.q[i] = A.poses[pair * 14 + 7 + i]; }
        for (int i = 0; i < 3; i++) { Tr.t[i] = A.poses[pair * 14 + 4 + i]; Tc.t[i] = A.poses[pair * 14 + 11 + i]; }
        s_Tref = Tr;
        s_T = se3_mul(Tc, se3_inverse(Tr));     // T_cur_from_ref = cur.Tcw * ref.Tcw^-1  (:36)
        s_chi2 = 1e10f;                         // reset(): chi2_ = 1e10
        s_stop = 0;
        s_nmeas = 0;
        s_iters = 0;
        for (int i = 0; i < 36; i++) s_H[i] = 0;
    }
    float2 *s_uv = (float2 *) (s_feat + A.ldsFeat);
    float4 *s_jac = (float4 *) (s_uv + A.ldsFeat);   // two per feature when A.jacLds
    uint8_t *s_img = (uint8_t *) s_feat + A.stageOff;   // A.stageBytes: the current image of a level that fits (coarse levels)
    if (!A.perLevel && N > 2 * kSiaBlock)   // (only pairs beyond the register form ever read the cache)
        for (int j = 0; j < 9; j++)
            for (int i = tid; i < N; i += kSiaBlock) rowCache[j * plane + i] = make_float4(0.f, 0.f, 0.f, 0.f);
    __syncthreads();                                            // s_Tref is set
    {   // per-feature terms that do not depend on the level, once per run: the keypoint, the exclusion flags and Tref * Xw go to LDS
        // (the level loop below used to re-read keys / world / flags for every (feature, row) item: two dependent global latencies each)
        const Se3 Tref = s_Tref;
        for (int i = tid; i < N; i += kSiaBlock) {              // visible_fts_: allocated once in run(), never cleared between levels
            if (!A.perLevel) visible[i] = 0;
            const bool excluded = (mpValid && !mpValid[i]) || (outlier && outlier[i]);
            const ygzf_kp kp = keys[i];
            s_uv[i] = excluded ? make_float2(-100.f, -100.f) : make_float2(kp.x, kp.y);   // fails every level's border test
            float xyz[3];
            if (A.unitWorld) {   // the world point is the keypoint's back-projection to depth 1 (k_backproject_unit's expressions, a launch of its own until round 4)
                const float Xu[3] = {(kp.x - A.cx) / A.fx, (kp.y - A.cy) / A.fy, 1.f};
                se3_act(Tref, Xu, xyz);
            } else
                se3_act(Tref, world + 3 * (size_t) i, xyz);
            s_feat[i] = make_float4(xyz[0], xyz[1], xyz[2], 0.f);
            if (A.jacLds) jac_terms(xyz, s_jac[2 * i], s_jac[2 * i + 1]);
        }
    }
    __syncthreads();
    if (N == 0) {                               // :24-27 "no features to track"
        if (tid == 0) { for (int i = 0; i < 48; i++) out[i] = 0; out[3] = 1.f; }
        return;
    }
    // The level loop exists twice.  REG (pairs of up to 2 x kSiaBlock features -- every usual configuration): the reference grids / gradient moments
    // of features tid and tid + kSiaBlock (sia_ref_patch) live in the thread's registers across the iterations of a level, 72 registers that replace
    // 208 bytes per feature and iteration from caches in global memory.  Otherwise every feature goes through the cache of its level (nine float4
    // planes + moments).  Two instantiations of one body instead of one loop with both paths: the register form has no room for the loads of the other.
    auto run_levels = [&](auto regTag) {
    constexpr bool REG = decltype(regTag)::value;
    float Vg[2][36];
    float4 mg[2];
#pragma unroll
    for (int f = 0; f < 2; f++) {
        mg[f] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int j = 0; j < 36; j++) Vg[f][j] = 0.f;
    }
    for (int level = A.maxLevel; level >= A.minLevel; level--) {
        const SiaLevel Lr = A.refLv[(long long) pair * A.lvStride + level], Lc = A.curLv[(long long) pair * A.lvStride + level];
        const float scale = A.invScale[level];
        // ---- precomputeReferencePatches.  The reference zeroes the whole jacobian cache per level (:41); only features that
        // stay "visible" from a coarser level but fail this level's border test can still read it, so only their rows are zeroed.
        // The current image of a coarse level fits the LDS left beside the feature tables: staged once per level with coalesced loads, its
        // 5 x 5-byte patch gathers then cost LDS bank cycles instead of 64 scattered cache lines per load instruction (the texture path
        // works such a gather off one line at a time, and the eight waves of the workgroup queue behind each other).
        const unsigned imgBytes = (unsigned) Lc.h * (unsigned) Lc.pitch;
        const bool staged = imgBytes + 8 <= (unsigned) A.stageBytes;
        if (staged) {
            // every level image the library hands to this kernel starts dword-aligned and is followed by >= 3 bytes of its own allocation
            // (64-byte pitches in the pyramid buffers, 64-byte rounded slots for uploaded frames): whole dwords, the last one may over-read
            if (((unsigned long long) Lc.img & 3u) == 0) {
                const unsigned nd = (imgBytes + 3) >> 2;
                for (unsigned d = tid; d < nd; d += kSiaBlock) ((unsigned *) s_img)[d] = ((const unsigned *) Lc.img)[d];
            } else {
                for (unsigned b = tid; b < imgBytes; b += kSiaBlock) s_img[b] = Lc.img[b];
            }
        }
        const uint8_t *s_cur = s_img;
        const long long p0c = DBG ? wall_clock64() : 0;
        if (A.perLevel) {
            const int li = level - A.minLevel;
            rowCache = (float4 *) (A.patchCache + (size_t) li * A.pcLevelStride + (size_t) pair * A.kpStride * 48);
            mom = (float4 *) (A.momCache + (size_t) li * A.momLevelStride) + (long long) pair * A.kpStride;
            const uint8_t *fl = A.levelFlags + (size_t) li * A.flagLevelStride + (long long) pair * A.kpStride;
            for (int i = tid; i < N; i += kSiaBlock) s_feat[i].w = fl[i] ? 1.f : 0.f;     // visible at this level or a coarser one
#pragma unroll
            for (int f = 0; f < 2; f++) {       // this level's grids of the thread's own two features: one read per level, registers from here on
                const int i = tid + f * kSiaBlock;
                if (REG && i < N && fl[i]) {
#pragma unroll
                    for (int j = 0; j < 9; j++) {
                        const float4 v = rowCache[j * plane + i];
                        Vg[f][4 * j] = v.x; Vg[f][4 * j + 1] = v.y; Vg[f][4 * j + 2] = v.z; Vg[f][4 * j + 3] = v.w;
                    }
                    mg[f] = mom[i];
                }
            }
        } else {
            // work item = feature: the 7 x 8 image bytes under its 6 x 6 grid are loaded once (seven 8-byte row loads, all in flight together).
            // A feature that fails this level's border test but was visible at a coarser one keeps that level's grid (registers / cache are
            // simply not overwritten -- the reference's patch cache is never cleared) and gets the mark "gradients are zero".
#pragma unroll
            for (int f = 0; f < 2; f++) {
                const int i = tid + f * kSiaBlock;
                if (REG && i < N) {
                    if (sia_ref_patch(Lr, scale, s_uv[i], Vg[f], mg[f])) {
                        visible[i] = 1;
                        s_feat[i].w = 1.f;
                    } else if (s_feat[i].w != 0.f)
                        mg[f] = make_float4(0.f, 0.f, 0.f, 1.f);
                }
            }
            for (int i = tid; !REG && i < N; i += kSiaBlock) {   // through the cache in global memory
                float Vt[36];
                float4 m;
                if (sia_ref_patch(Lr, scale, s_uv[i], Vt, m)) {
                    visible[i] = 1;
                    s_feat[i].w = 1.f;
#pragma unroll
                    for (int j = 0; j < 9; j++) rowCache[j * plane + i] = make_float4(Vt[4 * j], Vt[4 * j + 1], Vt[4 * j + 2], Vt[4 * j + 3]);
                    mom[i] = m;
                } else if (s_feat[i].w != 0.f)
                    mom[i] = make_float4(0.f, 0.f, 0.f, 1.f);
            }
        }
        __syncthreads();
        if (DBG && tid == 0 && A.dbg) A.dbg[4] += wall_clock64() - p0c;
        // ---- optimizeGaussNewton ----
        if (tid == 0) { s_Told = s_T; s_break = 0; }
        __syncthreads();
        for (int iter = 0; iter < A.nIter; iter++) {
            const long long c0 = DBG ? wall_clock64() : 0;
            const Se3 T = s_T;
            float acc[kAcc];
#pragma unroll
            for (int k = 0; k < kAcc; k++) acc[k] = 0.f;
            // work item = feature: projection, interpolation weights and the point's Jacobian terms once per feature (they were rebuilt for
            // each of its four patch rows), five dword-aligned 8-byte row loads for the 5 x 5 bytes under the patch
            // A thread owns the same features in every iteration -- REG: tid and tid + kSiaBlock, from its registers (the grids built / loaded at the top
            // of the level); otherwise tid, tid + kSiaBlock, ... through the caches in global memory -- and takes them in ascending order (the order
            // is part of the definition: the device-order oracle sums the same way).
#pragma unroll
            for (int f = 0; f < 2; f++) {
                const int i = tid + f * kSiaBlock;
                if (REG && i < N) sia_accumulate_feature(A, T, Lc, scale, staged, s_cur, s_feat[i], s_jac + 2 * i, Vg[f], mg[f], acc);
            }
            for (int i = tid; !REG && i < N; i += kSiaBlock) {
                const float4 ft = s_feat[i];
                if (ft.w == 0.f) continue;
                float Vt[36];
#pragma unroll
                for (int j = 0; j < 9; j++) {
                    const float4 v = rowCache[j * plane + i];
                    Vt[4 * j] = v.x; Vt[4 * j + 1] = v.y; Vt[4 * j + 2] = v.z; Vt[4 * j + 3] = v.w;
                }
                sia_accumulate_feature(A, T, Lc, scale, staged, s_cur, ft, s_jac + 2 * i, Vt, mom[i], acc);
            }
            const long long c1 = DBG ? wall_clock64() : 0;
            // fixed-shape reduction: wave totals of the 30 accumulators (wave_sums_30) -> one LDS partial per wave (8 x 30 floats); after ONE block barrier
            // wave 0 adds the eight waves in order (lanes 0..29), publishes the totals to its own lane 0 through LDS (wave-level
            // synchronisation only) and goes straight on to the solve -- one barrier less per iteration than the row-partial form
            wave_sums_30(acc, s_red + wave * kAcc, lane);
            if (DBG && A.dbg) {
                if (tid == 0) A.dbg[5] += wall_clock64() - c1;
                if (tid == kSiaBlock - 64) A.dbg[6] += c1 - c0;
                if (lane == 0) A.dbg[8 + wave] += c1 - c0;
            }
            __syncthreads();
            long long c2 = 0;
            if (wave == 0) {   // the whole wave stays together from here to the barrier: totals, solve and update are wave-uniform
                if (lane < kAcc) {
                    float v = 0.f;
#pragma unroll
                    for (int w2 = 0; w2 < kSiaBlock / 64; w2++) v += s_red[w2 * kAcc + lane];
                    s_tot[lane] = v;
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                if (DBG) c2 = wall_clock64();
                // lane 8 i + j: H[i][j] (accumulator of the upper-triangular entry), b[i] for j = 6
                const int ei = lane >> 3, ej = lane & 7;
                const int lo = min(ei, ej), hi = max(ei, ej);
                const int tIdx = ej < 6 ? 6 * lo - ((lo * (lo - 1)) >> 1) + (hi - lo) : 21 + ei;
                float el = (ei < 6 && ej < 7) ? s_tot[tIdx] : 0.f;
                const float chiSum = s_tot[27], nMeas = s_tot[28];
                if (ei < 6 && ej < 6) s_H[6 * ei + ej] = el;
                const float new_chi2 = chiSum / nMeas;           // chi2 / n_meas_ (NaN when nothing is visible, as the reference)
                const long long q0 = DBG ? wall_clock64() : 0;
                const float xl = ldlt_solve6_wave(el, lane);
                if (DBG && lane == 0 && A.dbg) A.dbg[7] += wall_clock64() - q0;
                float x[6];
#pragma unroll
                for (int a2 = 0; a2 < 6; a2++) x[a2] = lane_bcast(xl, a2);
                const bool stop = s_stop || isnan(x[0]);         // solve() failed -> stop_ (:235-236)
                const bool rollback = (iter > 0 && (double) new_chi2 > 1.2 * (double) s_chi2) || stop;
                Se3 Tnew = T;
                bool conv = false;
                if (!rollback) {
                    float negx[6];
#pragma unroll
                    for (int a2 = 0; a2 < 6; a2++) negx[a2] = -x[a2];
                    Tnew = se3_mul(T, se3_exp_wave(negx, lane));   // update(): T_new = T_old * exp(-x)
                    float nm = 0.f;
#pragma unroll
                    for (int a2 = 0; a2 < 6; a2++) nm = fmaxf(nm, fabsf(x[a2]));
                    conv = nm <= A.eps;                          // converged
                }
                if (lane == 0) {
                    s_nmeas = (int) nMeas;
                    s_iters++;
                    if (stop) s_stop = 1;
                    if (rollback) {
                        s_T = s_Told;
                        s_break = 1;
                    } else {
                        s_Told = T;
                        s_T = Tnew;
                        s_chi2 = new_chi2;
                        if (conv) s_break = 1;
                    }
                }
            }
            if (DBG && tid == 0 && A.dbg) { const long long c3 = wall_clock64(); A.dbg[0] += c1 - c0; A.dbg[1] += c2 - c1; A.dbg[2] += c3 - c2; A.dbg[3] += 1; }
            __syncthreads();
            if (s_break) break;
        }
        __syncthreads();
    }
    };
    if (N <= 2 * kSiaBlock) run_levels(std::true_type{});
    else run_levels(std::false_type{});
    if (tid == 0) {
        for (int i = 0; i < 4; i++) out[i] = s_T.q[i];
        for (int i = 0; i < 3; i++) out[4 + i] = s_T.t[i];
        out[7] = (float) (s_nmeas / 16);      // return n_meas_ / patch_area_
        out[8] = (float) s_iters;
        out[9] = s_chi2;
        out[10] = out[11] = 0.f;
        for (int i = 0; i < 36; i++) out[12 + i] = s_H[i];
    }
}

// per feature: float4 s_feat + float2 s_uv, plus two float4 of Jacobian terms while that keeps the workgroup under 96 KB (4000 features)
bool sia_jac_in_lds(int maxFeatures) { return (size_t) maxFeatures * 56 + 16 <= 96 * 1024; }
size_t sia_lds_bytes(int maxFeatures) {
    return (size_t) maxFeatures * (sizeof(float4) + sizeof(float2) + (sia_jac_in_lds(maxFeatures) ? 2 * sizeof(float4) : 0)) + 16;
}

// dynamic LDS the kernel may use (160 KB minus its static part)
static int sia_dyn_ceiling(hipError_t *err) {
    // a constant of the compiled kernel: asked once per process (thread-safe static initialisation), not on every call
    struct Q { hipError_t e; int v; };
    static const Q q = [] {
        hipFuncAttributes fa;
        Q r{hipFuncGetAttributes(&fa, (const void *) k_sia_run<false>), 0};
        if (r.e == hipSuccess) r.v = (int) std::min<size_t>(kMaxDynLds, 160 * 1024 - ((fa.sharedSizeBytes + 255) & ~(size_t) 255));
        return r;
    }();
    *err = q.e;
    return q.v;
}

// bytes of LDS left for the staged current image behind a feature table of featBytes (0 when nothing useful fits)
size_t sia_stage_bytes(size_t featBytes, size_t largestLevelBytes) {
    hipError_t e;
    const int ceiling = sia_dyn_ceiling(&e);
    if (e != hipSuccess || (size_t) ceiling < featBytes + 4096) return 0;
    const size_t avail = ((size_t) ceiling - featBytes) & ~(size_t) 15;
    return std::min(avail, (largestLevelBytes + 8 + 15) & ~(size_t) 15);
}

hipError_t sia_prepare(size_t ldsBytes) {
    // the kernel's static LDS (reduction partials, solver state) comes out of the same 160 KB: the ceiling is a constant of the kernel,
    // so every context sets the same value
    hipError_t e;
    const int ceiling = sia_dyn_ceiling(&e);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute((const void *) k_sia_run<false>, hipFuncAttributeMaxDynamicSharedMemorySize, ceiling);
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute((const void *) k_sia_run<true>, hipFuncAttributeMaxDynamicSharedMemorySize, ceiling);
}

// A.perLevel: fills the per-level caches k_sia_run then reads (maxN = largest feature count of a pair)
void launch_sia_precompute(hipStream_t st, const SiaArgs &A, int nPairs, int maxN) {
    if (maxN <= 0 || nPairs <= 0) return;
    hipLaunchKernelGGL(k_sia_precompute, dim3((maxN + 255) / 256, A.maxLevel - A.minLevel + 1, nPairs), dim3(256), 0, st, A);
}

void launch_sia(hipStream_t st, const SiaArgs &A, int nPairs, size_t ldsBytes) {
    if (A.dbg) hipLaunchKernelGGL(k_sia_run<true>, dim3(nPairs), dim3(kSiaBlock), ldsBytes, st, A);
    else hipLaunchKernelGGL(k_sia_run<false>, dim3(nPairs), dim3(kSiaBlock), ldsBytes, st, A);
}

}  // namespace ygzf
