// oracle_direct.cpp -- CPU ORACLE (test infrastructure): restatement of the direct local-map projection
//   ORBmatcher::GetWarpAffineMatrix / WarpAffine / FindDirectProjection   reference src/ORBmatcher.cc:1525-1602
//   ORBmatcher::GetBestSearchLevel / GetBilateralInterpUchar              include/ORBmatcher.h:185-211
//   ygz::Align2D                                                           src/Align.cc:8-104
//   KeyFrame::Pixel2Camera / World2Pixel                                   include/KeyFrame.h:173-194, include/Frame.h:154-175
// Eigen is not available: Matrix2f / Matrix3f determinant, inverse (cofactor form) and products are written out in natural order; this
// file DEFINES that order (unpinned) and the HIP path repeats it operation for operation, so the two agree bit for bit.
// PARITY: the control flow and every other operation are PINNED to the reference's own src/ORBmatcher.cc + src/Align.cc
// (tests/test_ref_matcher.py::test_find_direct_projection_equals_reference: those files, compiled where they lie over oracle/ref_shim/
// with this file's conventions for the small-matrix / pose algebra, return identical pixels, levels, flags and patches).
// Built with -ffp-contract=off.
#include <cmath>
#include <cstring>

#include "ygz_oracle.h"

namespace ygzo {

static const int WarpHalfPatchSize = 4, WarpPatchSize = 8;  // include/ORBmatcher.h:35-36

// Matrix3f::inverse() (Eigen compute_inverse_size3: cofactors, det from the first column)
void inverse3(const float m[9], float r[9]) {
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

// src/Align.cc:8-104
bool align2d(const Image &cur_img, const uint8_t *ref_patch_with_border, const uint8_t *ref_patch, int n_iter, float cur_px_estimate[2]) {
    const int halfpatch_size_ = 4, patch_size_ = 8, patch_area_ = 64;
    bool converged = false;
    float ref_patch_dx[patch_area_], ref_patch_dy[patch_area_];
    float H[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    const int ref_step = patch_size_ + 2;
    float *it_dx = ref_patch_dx, *it_dy = ref_patch_dy;
    for (int y = 0; y < patch_size_; ++y) {
        const uint8_t *it = ref_patch_with_border + (y + 1) * ref_step + 1;
        for (int x = 0; x < patch_size_; ++x, ++it, ++it_dx, ++it_dy) {
            float J[3];
            J[0] = (float) (0.5 * (it[1] - it[-1]));
            J[1] = (float) (0.5 * (it[ref_step] - it[-ref_step]));
            J[2] = 1;
            *it_dx = J[0];
            *it_dy = J[1];
            for (int a = 0; a < 3; a++)
                for (int b = 0; b < 3; b++) H[3 * a + b] += J[a] * J[b];
        }
    }
    float Hinv[9];
    inverse3(H, Hinv);
    float mean_diff = 0;
    float u = cur_px_estimate[0], v = cur_px_estimate[1];
    const float min_update_squared = (float) (0.03 * 0.03);
    const int cur_step = cur_img.w;
    float update[3] = {0, 0, 0};
    for (int iter = 0; iter < n_iter; ++iter) {
        int u_r = (int) std::floor(u);
        int v_r = (int) std::floor(v);
        if (u_r < halfpatch_size_ || v_r < halfpatch_size_ || u_r >= cur_img.w - halfpatch_size_ || v_r >= cur_img.h - halfpatch_size_) break;
        if (std::isnan(u) || std::isnan(v)) return false;
        float subpix_x = u - u_r;
        float subpix_y = v - v_r;
        float wTL = (float) ((1.0 - subpix_x) * (1.0 - subpix_y));
        float wTR = (float) (subpix_x * (1.0 - subpix_y));
        float wBL = (float) ((1.0 - subpix_x) * subpix_y);
        float wBR = subpix_x * subpix_y;
        const uint8_t *it_ref = ref_patch;
        const float *it_ref_dx = ref_patch_dx, *it_ref_dy = ref_patch_dy;
        float Jres[3] = {0, 0, 0};
        for (int y = 0; y < patch_size_; ++y) {
            const uint8_t *it = cur_img.d.data() + (size_t) (v_r + y - halfpatch_size_) * cur_step + u_r - halfpatch_size_;
            for (int x = 0; x < patch_size_; ++x, ++it, ++it_ref, ++it_ref_dx, ++it_ref_dy) {
                float search_pixel = wTL * it[0] + wTR * it[1] + wBL * it[cur_step] + wBR * it[cur_step + 1];
                float res = search_pixel - *it_ref + mean_diff;
                Jres[0] -= res * (*it_ref_dx);
                Jres[1] -= res * (*it_ref_dy);
                Jres[2] -= res;
            }
        }
        for (int a = 0; a < 3; a++) update[a] = (Hinv[3 * a] * Jres[0] + Hinv[3 * a + 1] * Jres[1]) + Hinv[3 * a + 2] * Jres[2];
        u += update[0];
        v += update[1];
        mean_diff += update[2];
        if (update[0] * update[0] + update[1] * update[1] < min_update_squared) {
            converged = true;
            break;
        }
    }
    cur_px_estimate[0] = u;
    cur_px_estimate[1] = v;
    return converged;
}

// include/ORBmatcher.h:200-211
static inline uint8_t bilateral_interp_uchar(double x, double y, const Image &gray) {
    const double xx = x - std::floor(x);
    const double yy = y - std::floor(y);
    const uint8_t *data = &gray.d[(size_t) int(y) * gray.w + int(x)];
    return (uint8_t) ((1 - xx) * (1 - yy) * data[0] + xx * (1 - yy) * data[1] + (1 - xx) * yy * data[gray.w] + xx * yy * data[gray.w + 1]);
}

// src/ORBmatcher.cc:1574-1602
bool find_direct_projection(const DirectRef &ref, const DirectCur &cur, const float mp_world[3], float px_curr[2], int *search_level,
                            uint8_t *patch_with_border_out) {
    const float px_ref[2] = {ref.kp.x, ref.kp.y};
    const SE3f pose_ref = ref.Tcw;
    const SE3f TCR = cur.Tcw.Mul(pose_ref.Inverse());
    // GetWarpAffineMatrix :1525-1548
    float ACR[4];  // row-major 2x2
    {
        const int level = ref.kp.octave;
        float pt_ref[3];
        pose_ref.Act(mp_world, pt_ref);
        const float depth = pt_ref[2];
        const float du[2] = {px_ref[0] + (float) WarpHalfPatchSize * ref.scaleFactors[level], px_ref[1] + 0.f * ref.scaleFactors[level]};
        const float dv[2] = {px_ref[0] + 0.f * ref.scaleFactors[level], px_ref[1] + (float) WarpHalfPatchSize * ref.scaleFactors[level]};
        const float pt_du_ref[3] = {(du[0] - ref.cx) * depth / ref.fx, (du[1] - ref.cy) * depth / ref.fy, depth};
        const float pt_dv_ref[3] = {(dv[0] - ref.cx) * depth / ref.fx, (dv[1] - ref.cy) * depth / ref.fy, depth};
        auto world2pixel = [&](const float p[3], float o[2]) {
            float c[3];
            TCR.Act(p, c);
            o[0] = cur.fx * c[0] / c[2] + cur.cx;
            o[1] = cur.fy * c[1] / c[2] + cur.cy;
        };
        float px_cur[2], px_du[2], px_dv[2];
        world2pixel(pt_ref, px_cur);
        world2pixel(pt_du_ref, px_du);
        world2pixel(pt_dv_ref, px_dv);
        ACR[0] = (px_du[0] - px_cur[0]) / WarpHalfPatchSize;
        ACR[2] = (px_du[1] - px_cur[1]) / WarpHalfPatchSize;
        ACR[1] = (px_dv[0] - px_cur[0]) / WarpHalfPatchSize;
        ACR[3] = (px_dv[1] - px_cur[1]) / WarpHalfPatchSize;
    }
    // GetBestSearchLevel include/ORBmatcher.h:185-197
    int sl = 0;
    {
        float D = ACR[0] * ACR[3] - ACR[2] * ACR[1];
        const int max_level = ref.nlevels - 1;
        while (D > 3.0 && sl < max_level) {
            sl += 1;
            D *= ref.invLevelSigma2_1;
        }
    }
    *search_level = sl;
    // WarpAffine :1550-1572 with half_patch_size = WarpHalfPatchSize + 1
    uint8_t patch_with_border[(WarpPatchSize + 2) * (WarpPatchSize + 2)], patch[WarpPatchSize * WarpPatchSize];
    {
        const int half_patch_size = WarpHalfPatchSize + 1, patch_size = half_patch_size * 2;
        const float det = ACR[0] * ACR[3] - ACR[2] * ACR[1];   // Matrix2f::inverse(): adjugate * (1 / det)
        const float invdet = 1.f / det;
        const float ARC[4] = {ACR[3] * invdet, -ACR[1] * invdet, -ACR[2] * invdet, ACR[0] * invdet};
        const Image &img_ref = *ref.level_img;
        const float px_ref_pyr[2] = {px_ref[0] / ref.scaleFactors[ref.kp.octave], px_ref[1] / ref.scaleFactors[ref.kp.octave]};
        uint8_t *patch_ptr = patch_with_border;
        for (int y = 0; y < patch_size; y++) {
            for (int x = 0; x < patch_size; x++, ++patch_ptr) {
                float pp[2] = {(float) (x - half_patch_size), (float) (y - half_patch_size)};
                pp[0] *= ref.scaleFactors[sl];
                pp[1] *= ref.scaleFactors[sl];
                const float px[2] = {(ARC[0] * pp[0] + ARC[1] * pp[1]) + px_ref_pyr[0], (ARC[2] * pp[0] + ARC[3] * pp[1]) + px_ref_pyr[1]};
                if (px[0] < 0 || px[1] < 0 || px[0] >= img_ref.w - 1 || px[1] >= img_ref.h - 1) *patch_ptr = 0;
                else *patch_ptr = bilateral_interp_uchar(px[0], px[1], img_ref);
            }
        }
    }
    for (int y = 1; y < WarpPatchSize + 1; ++y)
        for (int x = 0; x < WarpPatchSize; ++x) patch[(y - 1) * WarpPatchSize + x] = patch_with_border[y * (WarpPatchSize + 2) + 1 + x];
    if (patch_with_border_out) std::memcpy(patch_with_border_out, patch_with_border, sizeof patch_with_border);
    float px_scaled[2] = {px_curr[0] * cur.invScaleFactors[sl], px_curr[1] * cur.invScaleFactors[sl]};
    const bool success = align2d(*cur.pyramid[sl], patch_with_border, patch, 10, px_scaled);
    px_curr[0] = px_scaled[0] * cur.scaleFactors[sl];
    px_curr[1] = px_scaled[1] * cur.scaleFactors[sl];
    return success;
}

}  // namespace ygzo
