// direct_kernels.hip -- the direct (photometric) local-map projection on gfx950 (product code).
//
//   k_direct_projection   ORBmatcher::FindDirectProjection          reference src/ORBmatcher.cc:1574-1602
//                         GetWarpAffineMatrix :1525-1548, GetBestSearchLevel include/ORBmatcher.h:185-197,
//                         WarpAffine :1550-1572 + GetBilateralInterpUchar include/ORBmatcher.h:200-211,
//                         ygz::Align2D src/Align.cc:8-104 (8x8 inverse-compositional LK with a mean-difference term, <= 10 iterations)
//
// Tracking::SearchLocalPointsDirect calls FindDirectProjection once per (MapPoint, observing KeyFrame) candidate; candidates are
// independent, so the device form takes them as a batch, with the KeyFrames' and the current frame's pyramids resident in an HBM
// image cache (ygzf_image_cache_*).
//
// One WAVE per candidate, lane = pixel of the 8x8 patch (a frame yields a few hundred to a few thousand candidates: a wave each fills the
// chip, a thread each would leave it to a handful of serial waves):
//   * pose algebra, affine warp matrix, search level: wave-uniform, evaluated by every lane (same registers, no divergence);
//   * WarpAffine: the 10x10 bordered patch is 100 independent double-precision bilinear samples -> lanes k and k + 64, into LDS;
//   * template gradient per lane; the Hessian sums J J^T over the 64 pixels -- every term is a multiple of 1/4 below 2^14 and every
//     partial sum stays below 2^22, so float addition is EXACT in any order: a DPP tree gives the reference's sequential result;
//   * per iteration every lane interpolates its search pixel and forms res, res*dx, res*dy with the reference's operation order; the three
//     running sums Jres[0..2] are NOT order-free (float), and the 0.03-px stop test makes the outcome discontinuous in their rounding:
//     lanes 0..2 take one sum each and subtract the 64 products in raster order out of LDS (three independent dependent-add chains side
//     by side), which is exactly the reference's accumulation.  Update, stop rule and write-back are wave-uniform again.
// Results equal the CPU definition bit for bit (refined pixel bit pattern, level, success flag, warped patch).
#include "kernels.h"
#include "se3_device.h"
#include "wave_ops.h"

namespace ygzf {

constexpr int kDirWaves = 4;                   // candidates per workgroup (waves are independent: no block barrier anywhere)
constexpr int kWarpHalf = 4, kWarpPatch = 8;   // include/ORBmatcher.h:35-36

__device__ __forceinline__ void inverse3(const float m[9], float r[9]) {   // Matrix3f::inverse(): cofactors, det from the first column
#define M(i, j) m[3 * (i) + (j)]
#define COF(i, j) (M(((i) + 1) % 3, ((j) + 1) % 3) * M(((i) + 2) % 3, ((j) + 2) % 3) - M(((i) + 1) % 3, ((j) + 2) % 3) * M(((i) + 2) % 3, ((j) + 1) % 3))
    const float c00 = COF(0, 0), c10 = COF(1, 0), c20 = COF(2, 0);
    const float det = (c00 * M(0, 0) + c10 * M(1, 0)) + c20 * M(2, 0);
    const float invdet = 1.f / det;
    r[0] = c00 * invdet; r[1] = c10 * invdet; r[2] = c20 * invdet;
    r[3] = COF(0, 1) * invdet; r[4] = COF(1, 1) * invdet; r[5] = COF(2, 1) * invdet;
    r[6] = COF(0, 2) * invdet; r[7] = COF(1, 2) * invdet; r[8] = COF(2, 2) * invdet;
#undef COF
#undef M
}

// exact sum over the wave of floats whose partial sums are all representable (see above): DPP inside rows of 16, rows through readlane
__device__ __forceinline__ float wave_sum_exact(float v) {
#define FSTEP(CTRL) v = v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
    FSTEP(0xb1) FSTEP(0x4e) FSTEP(0x124) FSTEP(0x128)
#undef FSTEP
    const int iv = __builtin_bit_cast(int, v);
    return (__builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 0)) + __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 16))) +
           (__builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 32)) + __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 48)));
}
__device__ __forceinline__ float lane_bcast(float v, int lane) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane)); }

__device__ __forceinline__ void dir_lds_sync() {   // orders this wave's LDS writes before its later reads by other lanes
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__global__ __launch_bounds__(64 * kDirWaves) void k_direct_projection(DirectArgs A) {
    __shared__ uint8_t s_pwb[kDirWaves][112];          // _patch_with_border (10 x 10) of this wave's candidate
    __shared__ __attribute__((aligned(16))) float s_prod[kDirWaves][3][64];         // res*dx | res*dy | res per pixel, raster order
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int i = blockIdx.x * kDirWaves + wv;
    if (i >= A.n) return;
    uint8_t *pwb = s_pwb[wv];
    const ygzf_kp kp = A.refKp[i];
    const float px_ref[2] = {kp.x, kp.y};
    Se3 pose_ref, Tcur;
    for (int k = 0; k < 4; k++) { pose_ref.q[k] = A.refTcw7[7 * (size_t) i + k]; Tcur.q[k] = A.curTcw[k]; }
    for (int k = 0; k < 3; k++) { pose_ref.t[k] = A.refTcw7[7 * (size_t) i + 4 + k]; Tcur.t[k] = A.curTcw[4 + k]; }
    const Se3 TCR = se3_mul(Tcur, se3_inverse(pose_ref));
    const float *mp = A.mpWorld + 3 * (size_t) i;
    // ---- GetWarpAffineMatrix (wave-uniform)
    float ACR[4];
    {
        const int level = kp.octave;
        float pt_ref[3];
        const float mpw[3] = {mp[0], mp[1], mp[2]};
        se3_act(pose_ref, mpw, pt_ref);
        const float depth = pt_ref[2];
        const float du[2] = {px_ref[0] + (float) kWarpHalf * A.scale[level], px_ref[1] + 0.f * A.scale[level]};
        const float dv[2] = {px_ref[0] + 0.f * A.scale[level], px_ref[1] + (float) kWarpHalf * A.scale[level]};
        const float pt_du_ref[3] = {(du[0] - A.cx) * depth / A.fx, (du[1] - A.cy) * depth / A.fy, depth};
        const float pt_dv_ref[3] = {(dv[0] - A.cx) * depth / A.fx, (dv[1] - A.cy) * depth / A.fy, depth};
        float c[3], px_cur[2], px_du[2], px_dv[2];
        se3_act(TCR, pt_ref, c);
        px_cur[0] = A.fx * c[0] / c[2] + A.cx; px_cur[1] = A.fy * c[1] / c[2] + A.cy;
        se3_act(TCR, pt_du_ref, c);
        px_du[0] = A.fx * c[0] / c[2] + A.cx; px_du[1] = A.fy * c[1] / c[2] + A.cy;
        se3_act(TCR, pt_dv_ref, c);
        px_dv[0] = A.fx * c[0] / c[2] + A.cx; px_dv[1] = A.fy * c[1] / c[2] + A.cy;
        ACR[0] = (px_du[0] - px_cur[0]) / kWarpHalf;
        ACR[2] = (px_du[1] - px_cur[1]) / kWarpHalf;
        ACR[1] = (px_dv[0] - px_cur[0]) / kWarpHalf;
        ACR[3] = (px_dv[1] - px_cur[1]) / kWarpHalf;
    }
    // ---- GetBestSearchLevel (wave-uniform)
    int sl = 0;
    {
        float D = ACR[0] * ACR[3] - ACR[2] * ACR[1];
        const int max_level = A.nlevels - 1;
        while (D > 3.0 && sl < max_level) {
            sl += 1;
            D *= A.invLevelSigma2_1;
        }
    }
    if (lane == 0) A.searchLevel[i] = sl;
    // ---- WarpAffine, half_patch_size = 5 -> 10 x 10: pixel k = lane and lane + 64
    {
        const int half_patch_size = kWarpHalf + 1, patch_size = 2 * half_patch_size;
        const float det = ACR[0] * ACR[3] - ACR[2] * ACR[1];
        const float invdet = 1.f / det;
        const float ARC[4] = {ACR[3] * invdet, -ACR[1] * invdet, -ACR[2] * invdet, ACR[0] * invdet};
        const LevelGeom g = A.geom[kp.octave];
        int pitch;
        const uint8_t *img = level_ptr(A.cache, g, kp.octave, A.refSlot[i], &pitch);
        const float px_ref_pyr[2] = {px_ref[0] / A.scale[kp.octave], px_ref[1] / A.scale[kp.octave]};
#pragma unroll
        for (int half = 0; half < 2; half++) {
            const int k = lane + 64 * half;
            if (k < patch_size * patch_size) {
                const int y = k / patch_size, x = k - y * patch_size;
                float pp[2] = {(float) (x - half_patch_size), (float) (y - half_patch_size)};
                pp[0] *= A.scale[sl];
                pp[1] *= A.scale[sl];
                const float px[2] = {(ARC[0] * pp[0] + ARC[1] * pp[1]) + px_ref_pyr[0], (ARC[2] * pp[0] + ARC[3] * pp[1]) + px_ref_pyr[1]};
                uint8_t val = 0;
                if (!(px[0] < 0 || px[1] < 0 || px[0] >= g.w - 1 || px[1] >= g.h - 1)) {
                    const double X = (double) px[0], Y = (double) px[1];
                    const double xx = X - floor(X), yy = Y - floor(Y);
                    const uint8_t *data = img + (long long) (int) Y * pitch + (int) X;
                    val = (uint8_t) ((1 - xx) * (1 - yy) * data[0] + xx * (1 - yy) * data[1] + (1 - xx) * yy * data[pitch] + xx * yy * data[pitch + 1]);
                }
                pwb[k] = val;
                if (A.patches) A.patches[100 * (size_t) i + k] = val;
            }
        }
    }
    dir_lds_sync();
    // ---- Align2D on cur level sl: lane = pixel (py, px) of the 8 x 8 patch
    const float u0 = A.pxCurr[2 * (size_t) i] * A.invScale[sl], v0 = A.pxCurr[2 * (size_t) i + 1] * A.invScale[sl];
    float u = u0, v = v0;
    bool converged = false, isnanFail = false;
    {
        const int ref_step = kWarpPatch + 2;
        const int py = lane >> 3, pxl = lane & 7;
        const int c = (py + 1) * ref_step + 1 + pxl;
        const float ref = (float) pwb[c];
        const float dx = (float) (0.5 * ((int) pwb[c + 1] - (int) pwb[c - 1]));
        const float dy = (float) (0.5 * ((int) pwb[c + ref_step] - (int) pwb[c - ref_step]));
        float H[9];
        H[0] = wave_sum_exact(dx * dx);
        H[1] = H[3] = wave_sum_exact(dx * dy);
        H[2] = H[6] = wave_sum_exact(dx);
        H[4] = wave_sum_exact(dy * dy);
        H[5] = H[7] = wave_sum_exact(dy);
        H[8] = 64.f;
        float Hinv[9];
        inverse3(H, Hinv);
        float mean_diff = 0;
        const float min_update_squared = (float) (0.03 * 0.03);
        const LevelGeom g = A.geom[sl];
        int cur_step;
        const uint8_t *cur = level_ptr(A.cache, g, sl, A.curSlot, &cur_step);
        float *prod = &s_prod[wv][0][0];
        for (int iter = 0; iter < 10; ++iter) {
            const int u_r = (int) floorf(u), v_r = (int) floorf(v);
            if (u_r < 4 || v_r < 4 || u_r >= g.w - 4 || v_r >= g.h - 4) break;
            if (isnan(u) || isnan(v)) { isnanFail = true; break; }
            const float subpix_x = u - u_r, subpix_y = v - v_r;
            const float wTL = (float) ((1.0 - subpix_x) * (1.0 - subpix_y));
            const float wTR = (float) (subpix_x * (1.0 - subpix_y));
            const float wBL = (float) ((1.0 - subpix_x) * subpix_y);
            const float wBR = subpix_x * subpix_y;
            const uint8_t *it = cur + (long long) (v_r + py - 4) * cur_step + u_r - 4 + pxl;
            const float search_pixel = wTL * it[0] + wTR * it[1] + wBL * it[cur_step] + wBR * it[cur_step + 1];
            const float res = search_pixel - ref + mean_diff;
            prod[lane] = res * dx;
            prod[64 + lane] = res * dy;
            prod[128 + lane] = res;
            dir_lds_sync();
            // lanes 0..2: Jres[lane] -= product[k] for k = 0..63 in raster order (the reference's accumulation order)
            float acc = 0.f;
            if (lane < 3) {
                const float4 *pp = (const float4 *) (prod + 64 * lane);
#pragma unroll
                for (int k4 = 0; k4 < 16; k4++) {
                    const float4 q = pp[k4];
                    acc -= q.x; acc -= q.y; acc -= q.z; acc -= q.w;
                }
            }
            dir_lds_sync();   // the products are consumed before the next iteration overwrites them
            const float Jres[3] = {lane_bcast(acc, 0), lane_bcast(acc, 1), lane_bcast(acc, 2)};
            float update[3];
#pragma unroll
            for (int a = 0; a < 3; a++) update[a] = (Hinv[3 * a] * Jres[0] + Hinv[3 * a + 1] * Jres[1]) + Hinv[3 * a + 2] * Jres[2];
            u += update[0];
            v += update[1];
            mean_diff += update[2];
            if (update[0] * update[0] + update[1] * update[1] < min_update_squared) {
                converged = true;
                break;
            }
        }
    }
    if (isnanFail) {   // `return false` before cur_px_estimate is written back (:59-61): px_scaled keeps its entry value
        u = u0;
        v = v0;
    }
    if (lane == 0) {
        A.pxCurr[2 * (size_t) i] = u * A.scale[sl];
        A.pxCurr[2 * (size_t) i + 1] = v * A.scale[sl];
        A.success[i] = converged ? 1 : 0;
    }
}

void launch_direct_projection(hipStream_t st, const DirectArgs &A) {
    if (A.n <= 0) return;
    hipLaunchKernelGGL(k_direct_projection, dim3((A.n + kDirWaves - 1) / kDirWaves), dim3(64 * kDirWaves), 0, st, A);
}

}  // namespace ygzf
