// oracle_align.cpp -- CPU ORACLE (test infrastructure): restatement of ygz::SparseImgAlign
// (reference src/SparseImageAlign.cc, include/SparseImageAlign.h:90-111), of the Gauss-Newton driver it inherits
// (include/NLSSolver_impl.hpp:17-91, reset :288-298) and of the Sophus SE3f operations it uses
// (Thirdparty/sophus/sophus/se3.hpp:159-171,267-271,406-428, so3.hpp:234-237,268-276,425-456).
// Eigen is not available: small fixed-size products are written out in natural (row, left-to-right) order and
// H.ldlt().solve(b) is a pivoted LDL^T in float.  fp32 throughout; the HIP path is graded at 1e-5 on the SE3
// output, not bit-exact.  Built with -ffp-contract=off.
// PARITY: the aligner's own logic (driver, level loop, caches, residual loop, stop / rollback rules) is PINNED to the reference's
// src/SparseImageAlign.cc + NLSSolver (tests/test_ref_matcher.py::test_sparse_img_align_equals_reference: compiled where they lie over
// oracle/ref_shim/, they return the same SE3 bit pattern, count, chi2 and Hessian); the Sophus / Eigen restatements in this file
// (quaternion algebra, exp, LDL^T, product order) are what that build uses too, and stay unpinned.
#include <cmath>
#include <cstring>

#include "ygz_oracle.h"

namespace ygzo {

static const float kSophusEps = 1e-5f;  // SophusConstants<float>::epsilon()

static inline void quat_mul(const float a[4], const float b[4], float o[4]) {  // Eigen quat product, (x,y,z,w)
    float r[4];
    r[3] = a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2];
    r[0] = a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1];
    r[1] = a[3] * b[1] + a[1] * b[3] + a[2] * b[0] - a[0] * b[2];
    r[2] = a[3] * b[2] + a[2] * b[3] + a[0] * b[1] - a[1] * b[0];
    std::memcpy(o, r, sizeof(r));
}
static inline void quat_normalize(float q[4]) {  // so3.hpp:268-276
    float n = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    for (int i = 0; i < 4; i++) q[i] /= n;
}
static inline void quat_rotate(const float q[4], const float v[3], float o[3]) {  // Eigen _transformVector
    float uv[3] = {q[1] * v[2] - q[2] * v[1], q[2] * v[0] - q[0] * v[2], q[0] * v[1] - q[1] * v[0]};
    for (int i = 0; i < 3; i++) uv[i] += uv[i];
    float c[3] = {q[1] * uv[2] - q[2] * uv[1], q[2] * uv[0] - q[0] * uv[2], q[0] * uv[1] - q[1] * uv[0]};
    for (int i = 0; i < 3; i++) o[i] = v[i] + q[3] * uv[i] + c[i];
}

void SE3f::RotationMatrix(float R[9]) const {  // Eigen toRotationMatrix
    const float tx = 2 * q[0], ty = 2 * q[1], tz = 2 * q[2];
    const float twx = tx * q[3], twy = ty * q[3], twz = tz * q[3];
    const float txx = tx * q[0], txy = ty * q[0], txz = tz * q[0];
    const float tyy = ty * q[1], tyz = tz * q[1], tzz = tz * q[2];
    R[0] = 1 - (tyy + tzz); R[1] = txy - twz;       R[2] = txz + twy;
    R[3] = txy + twz;       R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
    R[6] = txz - twy;       R[7] = tyz + twx;       R[8] = 1 - (txx + tyy);
}

void SE3f::Act(const float p[3], float out[3]) const {
    float r[3];
    quat_rotate(q, p, r);
    for (int i = 0; i < 3; i++) out[i] = r[i] + t[i];
}

SE3f SE3f::Inverse() const {  // se3.hpp:168-172
    SE3f o;
    o.q[0] = -q[0]; o.q[1] = -q[1]; o.q[2] = -q[2]; o.q[3] = q[3];
    quat_normalize(o.q);  // SO3Group(Quaternion) ctor normalises
    float nt[3] = {t[0] * -1.f, t[1] * -1.f, t[2] * -1.f};
    quat_rotate(o.q, nt, o.t);
    return o;
}

SE3f SE3f::Mul(const SE3f &other) const {  // operator*: fastMultiply (se3.hpp:159-163) + normalize
    SE3f r = *this;
    float rt[3];
    quat_rotate(r.q, other.t, rt);
    for (int i = 0; i < 3; i++) r.t[i] += rt[i];
    quat_mul(r.q, other.q, r.q);
    quat_normalize(r.q);
    return r;
}

SE3f SE3f::Exp(const float a[6]) {  // se3.hpp:406-428
    const float *omega = a + 3;
    const float theta_sq = omega[0] * omega[0] + omega[1] * omega[1] + omega[2] * omega[2];
    const float theta = std::sqrt(theta_sq);
    const float half_theta = 0.5f * theta;
    float imag_factor, real_factor;
    if (theta < kSophusEps) {  // so3.hpp:434-444
        const float theta_po4 = theta_sq * theta_sq;
        imag_factor = 0.5f - (float) (1.0 / 48.0) * theta_sq + (float) (1.0 / 3840.0) * theta_po4;
        real_factor = 1.f - 0.5f * theta_sq + (float) (1.0 / 384.0) * theta_po4;
    } else {
        const float sin_half_theta = std::sin(half_theta);
        imag_factor = sin_half_theta / theta;
        real_factor = std::cos(half_theta);
    }
    SE3f r;
    r.q[3] = real_factor;
    r.q[0] = imag_factor * omega[0];
    r.q[1] = imag_factor * omega[1];
    r.q[2] = imag_factor * omega[2];
    quat_normalize(r.q);
    // Omega = hat(omega), V = I + (1-cos)/theta^2 Omega + (theta-sin)/theta^3 Omega^2
    const float O[9] = {0, -omega[2], omega[1], omega[2], 0, -omega[0], -omega[1], omega[0], 0};
    float O2[9];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) O2[3 * i + j] = O[3 * i] * O[j] + O[3 * i + 1] * O[3 + j] + O[3 * i + 2] * O[6 + j];
    float V[9];
    if (theta < kSophusEps) {
        r.RotationMatrix(V);
    } else {
        // se3.hpp:419-422: `Scalar theta_sq = theta*theta;` -- the SQUARE OF THE ROOT, not omega.squaredNorm() (they differ in the last bit now
        // and then; found by tests/test_ref_sophus.py, which runs the reference's own header beside this function)
        const float theta_sq2 = theta * theta;
        const float c1 = (1.f - std::cos(theta)) / theta_sq2;
        const float c2 = (theta - std::sin(theta)) / (theta_sq2 * theta);
        for (int i = 0; i < 9; i++) V[i] = ((i % 4 == 0) ? 1.f : 0.f) + c1 * O[i] + c2 * O2[i];
    }
    for (int i = 0; i < 3; i++) r.t[i] = V[3 * i] * a[0] + V[3 * i + 1] * a[1] + V[3 * i + 2] * a[2];
    return r;
}

SE3f SE3f::FromRt(const float R[9], const float t_[3]) {  // Eigen Quaternion(Matrix3) + normalise
    SE3f o;
    float tr = R[0] + R[4] + R[8];
    if (tr > 0) {
        float s = std::sqrt(tr + 1.f);
        o.q[3] = 0.5f * s;
        s = 0.5f / s;
        o.q[0] = (R[7] - R[5]) * s;
        o.q[1] = (R[2] - R[6]) * s;
        o.q[2] = (R[3] - R[1]) * s;
    } else {
        int i = 0;
        if (R[4] > R[0]) i = 1;
        if (R[8] > R[4 * i]) i = 2;
        int j = (i + 1) % 3, k = (j + 1) % 3;
        float s = std::sqrt(R[4 * i] - R[4 * j] - R[4 * k] + 1.f);
        o.q[i] = 0.5f * s;
        s = 0.5f / s;
        o.q[3] = (R[3 * k + j] - R[3 * j + k]) * s;
        o.q[j] = (R[3 * j + i] + R[3 * i + j]) * s;
        o.q[k] = (R[3 * k + i] + R[3 * i + k]) * s;
    }
    quat_normalize(o.q);
    for (int i = 0; i < 3; i++) o.t[i] = t_[i];
    return o;
}

// x = H.ldlt().solve(b): LDL^T with diagonal pivoting (Eigen's LDLT picks the largest |diagonal|), float.
bool ldlt_solve6(const float Hin[36], const float bin[6], float x[6]) {
    float A[36];
    float b[6];
    int perm[6];
    std::memcpy(A, Hin, sizeof(A));
    std::memcpy(b, bin, sizeof(b));
    for (int i = 0; i < 6; i++) perm[i] = i;
    for (int k = 0; k < 6; k++) {
        int p = k;
        float best = std::fabs(A[7 * k]);
        for (int i = k + 1; i < 6; i++)
            if (std::fabs(A[7 * i]) > best) { best = std::fabs(A[7 * i]); p = i; }
        if (p != k) {
            for (int j = 0; j < 6; j++) std::swap(A[6 * k + j], A[6 * p + j]);
            for (int j = 0; j < 6; j++) std::swap(A[6 * j + k], A[6 * j + p]);
            std::swap(b[k], b[p]);
            std::swap(perm[k], perm[p]);
        }
        const float d = A[7 * k];
        for (int i = k + 1; i < 6; i++) {
            const float l = A[6 * i + k] / d;
            for (int j = k + 1; j < 6; j++) A[6 * i + j] -= l * A[6 * k + j];
            A[6 * i + k] = l;
        }
    }
    float y[6];
    for (int i = 0; i < 6; i++) {  // L y = b
        float s = b[i];
        for (int j = 0; j < i; j++) s -= A[6 * i + j] * y[j];
        y[i] = s;
    }
    for (int i = 0; i < 6; i++) y[i] = y[i] / A[7 * i];  // D z = y
    float z[6];
    for (int i = 5; i >= 0; i--) {  // L^T w = z
        float s = y[i];
        for (int j = i + 1; j < 6; j++) s -= A[6 * j + i] * z[j];
        z[i] = s;
    }
    for (int i = 0; i < 6; i++) x[perm[i]] = z[i];
    return true;
}

namespace {
struct Aligner {
    static const int patch_halfsize_ = 2, patch_size_ = 4, patch_area_ = 16;
    const AlignFrame &ref, &cur;
    int level_ = 0;
    std::vector<float> ref_patch_cache_;  // N x 16
    std::vector<float> jacobian_cache_;   // (N*16) x 6, column j of the reference's 6 x (N*16) matrix
    std::vector<uint8_t> visible_fts_;
    bool have_ref_patch_cache_ = false;
    float H_[36], Jres_[6], x_[6];
    float chi2_ = 1e10f;
    size_t n_meas_ = 0;
    int n_iter_ = 10, iter_ = 0;
    bool stop_ = false;
    float eps_ = 0.000001f;
    int iters_total = 0;
    // ---- device-order mode (test infrastructure for the HIP kernel, not a restatement of the reference) ---------------------------------------
    // The reference sums J J^T, J res and res^2 pixel by pixel, feature after feature.  k_sia_run (orb_ygz_slam_amd/csrc/align_kernels.hip) forms the
    // same sums from per-feature gradient moments (H_feature = Sxx Jf0 Jf0^T + Sxy (Jf0 Jf1^T + Jf1 Jf0^T) + Syy Jf1 Jf1^T), with fused multiply-adds
    // at stated places, 512 threads striding over the features and a fixed reduction tree.  With device_order_ this class evaluates computeResiduals
    // with exactly that arithmetic -- same formulation, same fmaf placement, same tree -- so that the kernel's H, b and chi2 can be demanded BIT FOR
    // BIT (tests/test_gpu_align.py, tests/test_gpu_fuzz.py) instead of being bounded by a re-ordering allowance.  Everything else (precompute values,
    // visibility, solve, update, stop rules) is shared with the reference-order mode.
    bool device_order_ = false;
    std::vector<float> dx_cache_, dy_cache_;   // N x 16: central-difference gradients of the reference patch (device-order mode)
    std::vector<float> mom_cache_;             // N x 3: sum dx*dx, dx*dy, dy*dy over the patch, fmaf chain in pixel order

    Aligner(const AlignFrame &r, const AlignFrame &c) : ref(r), cur(c) {}

    // SparseImageAlign.h:90-111
    static void JacobXYZ2Cam(const float xyz[3], float J[12]) {
        const float x = xyz[0], y = xyz[1];
        const float z_inv = (float) (1. / xyz[2]);
        const float z_inv_2 = z_inv * z_inv;
        J[0] = -z_inv;
        J[1] = 0.0f;
        J[2] = x * z_inv_2;
        J[3] = y * J[2];
        J[4] = (float) -(1.0 + x * J[2]);
        J[5] = y * z_inv;
        J[6] = 0.0f;
        J[7] = -z_inv;
        J[8] = y * z_inv_2;
        J[9] = (float) (1.0 + y * J[8]);
        J[10] = -J[3];
        J[11] = -x * z_inv;
    }

    // src/SparseImageAlign.cc:57-128
    void precomputeReferencePatches() {
        const int border = patch_halfsize_ + 1;
        const Image &ref_img = *ref.pyramid[level_];
        const int stride = ref_img.w;
        const float scale = ref.invScaleFactors[level_];
        const float focal_length = ref.fx;
        for (int i = 0; i < ref.N; i++) {
            if (!ref.mp_valid[i] || ref.outlier[i]) continue;
            const KeyPoint &kp = ref.keys[i];
            const float u_ref = kp.x * scale, v_ref = kp.y * scale;
            const int u_ref_i = (int) floorf(u_ref), v_ref_i = (int) floorf(v_ref);
            if (u_ref_i - border < 0 || v_ref_i - border < 0 || u_ref_i + border >= ref_img.w ||
                v_ref_i + border >= ref_img.h)
                continue;
            visible_fts_[i] = 1;
            float xyz_ref[3];
            ref.Tcw.Act(&ref.mp_world[3 * i], xyz_ref);
            float frame_jac[12];
            JacobXYZ2Cam(xyz_ref, frame_jac);
            const float subpix_u_ref = u_ref - u_ref_i, subpix_v_ref = v_ref - v_ref_i;
            const float w_ref_tl = (float) ((1.0 - subpix_u_ref) * (1.0 - subpix_v_ref));
            const float w_ref_tr = (float) (subpix_u_ref * (1.0 - subpix_v_ref));
            const float w_ref_bl = (float) ((1.0 - subpix_u_ref) * subpix_v_ref);
            const float w_ref_br = subpix_u_ref * subpix_v_ref;
            size_t pixel_counter = 0;
            float *cache_ptr = &ref_patch_cache_[(size_t) patch_area_ * i];
            for (int y = 0; y < patch_size_; ++y) {
                const uint8_t *p = &ref_img.d[(size_t) (v_ref_i + y - patch_halfsize_) * stride + (u_ref_i - patch_halfsize_)];
                for (int x = 0; x < patch_size_; ++x, ++p, ++cache_ptr, ++pixel_counter) {
                    *cache_ptr = w_ref_tl * p[0] + w_ref_tr * p[1] + w_ref_bl * p[stride] + w_ref_br * p[stride + 1];
                    float dx = 0.5f * ((w_ref_tl * p[1] + w_ref_tr * p[2] + w_ref_bl * p[stride + 1] + w_ref_br * p[stride + 2]) -
                                       (w_ref_tl * p[-1] + w_ref_tr * p[0] + w_ref_bl * p[stride - 1] + w_ref_br * p[stride]));
                    float dy = 0.5f * ((w_ref_tl * p[stride] + w_ref_tr * p[1 + stride] + w_ref_bl * p[stride * 2] +
                                        w_ref_br * p[stride * 2 + 1]) -
                                       (w_ref_tl * p[-stride] + w_ref_tr * p[1 - stride] + w_ref_bl * p[0] + w_ref_br * p[1]));
                    float *J = &jacobian_cache_[((size_t) i * patch_area_ + pixel_counter) * 6];
                    for (int k = 0; k < 6; k++) J[k] = (dx * frame_jac[k] + dy * frame_jac[6 + k]) * (focal_length * scale);
                    if (device_order_) {
                        dx_cache_[(size_t) i * patch_area_ + pixel_counter] = dx;
                        dy_cache_[(size_t) i * patch_area_ + pixel_counter] = dy;
                    }
                }
            }
            if (device_order_) {   // align_kernels.hip: sxx / sxy / syy, one fmaf per term in pixel order
                float sxx = 0.f, sxy = 0.f, syy = 0.f;
                for (int p2 = 0; p2 < patch_area_; p2++) {
                    const float dx = dx_cache_[(size_t) i * patch_area_ + p2], dy = dy_cache_[(size_t) i * patch_area_ + p2];
                    sxx = std::fmaf(dx, dx, sxx);
                    sxy = std::fmaf(dx, dy, sxy);
                    syy = std::fmaf(dy, dy, syy);
                }
                mom_cache_[3 * (size_t) i] = sxx; mom_cache_[3 * (size_t) i + 1] = sxy; mom_cache_[3 * (size_t) i + 2] = syy;
            }
        }
        have_ref_patch_cache_ = true;
    }

    // :130-231 (use_weights_ == false, display_ == false in the reference's only call site)
    // k_sia_run's accumulate phase and reduction (align_kernels.hip: the feature loop of optimizeGaussNewton, wave_sums_30, the cross-wave sum)
    float computeResidualsDeviceOrder(const SE3f &T) {
        const Image &cur_img = *cur.pyramid[level_];
        const int stride = cur_img.w, border = patch_halfsize_ + 1;
        const float scale = ref.invScaleFactors[level_];
        const int kBlock = 512, kAcc = 30;
        std::vector<float> acc((size_t) kBlock * kAcc, 0.f);     // thread t, accumulator k at [t * 30 + k]
        for (int i = 0; i < ref.N; i++) {
            if (!visible_fts_[i]) continue;
            float *a = &acc[(size_t) (i % kBlock) * kAcc];        // thread i % 512 takes features i, i + 512, ... in ascending order
            float xr[3], xc[3];
            ref.Tcw.Act(&ref.mp_world[3 * i], xr);
            T.Act(xr, xc);
            const float ucx = cur.fx * xc[0] / xc[2] + cur.cx, ucy = cur.fy * xc[1] / xc[2] + cur.cy;
            const float u_cur = ucx * scale, v_cur = ucy * scale;
            const int ui = (int) floorf(u_cur), vi = (int) floorf(v_cur);
            if (ui < 0 || vi < 0 || ui - border < 0 || vi - border < 0 || ui + border >= cur_img.w || vi + border >= cur_img.h) continue;
            a[29] += 1.f;
            const float su = u_cur - ui, sv = v_cur - vi;
            const float usu = 1.f - su, usv = 1.f - sv;
            const float w_tl = usu * usv, w_tr = su * usv, w_bl = usu * sv, w_br = su * sv;
            float J[12];
            JacobXYZ2Cam(xr, J);
            const float fs = ref.fx * scale;
            float Jf[12];
            for (int k = 0; k < 12; k++) Jf[k] = J[k] * fs;
            a[28] += 16.f;
            float sxr = 0.f, syr = 0.f;
            for (int y = 0; y < patch_size_; y++) {
                const uint8_t *p = &cur_img.d[(size_t) (vi + y - patch_halfsize_) * stride + (ui - patch_halfsize_)];
                for (int x = 0; x < patch_size_; x++, p++) {
                    const float I = std::fmaf(w_br, (float) p[stride + 1], std::fmaf(w_bl, (float) p[stride], std::fmaf(w_tr, (float) p[1], w_tl * (float) p[0])));
                    const size_t pc = (size_t) i * patch_area_ + 4 * y + x;
                    const float res = I - ref_patch_cache_[pc];
                    a[27] = std::fmaf(res, res, a[27]);
                    sxr = std::fmaf(dx_cache_[pc], res, sxr);
                    syr = std::fmaf(dy_cache_[pc], res, syr);
                }
            }
            const float Sx = mom_cache_[3 * (size_t) i], Sy = mom_cache_[3 * (size_t) i + 1], Sz = mom_cache_[3 * (size_t) i + 2];
            float P[6], Q[6];
            P[0] = Sx * Jf[0]; Q[0] = Sy * Jf[0];
            P[1] = Sy * Jf[7]; Q[1] = Sz * Jf[7];
            for (int k = 2; k < 6; k++) {
                P[k] = std::fmaf(Sy, Jf[6 + k], Sx * Jf[k]);
                Q[k] = std::fmaf(Sz, Jf[6 + k], Sy * Jf[k]);
            }
            for (int b2 = 0; b2 < 6; b2++) a[b2] = std::fmaf(Jf[0], P[b2], a[b2]);
            for (int b2 = 1; b2 < 6; b2++) a[5 + b2] = std::fmaf(Jf[7], Q[b2], a[5 + b2]);
            int t = 11;
            for (int r = 2; r < 6; r++)
                for (int b2 = r; b2 < 6; b2++, t++) {
                    a[t] = std::fmaf(Jf[r], P[b2], a[t]);
                    a[t] = std::fmaf(Jf[6 + r], Q[b2], a[t]);
                }
            a[21] = std::fmaf(-Jf[0], sxr, a[21]);
            a[22] = std::fmaf(-Jf[7], syr, a[22]);
            for (int k = 2; k < 6; k++) {
                a[21 + k] = std::fmaf(-Jf[k], sxr, a[21 + k]);
                a[21 + k] = std::fmaf(-Jf[6 + k], syr, a[21 + k]);
            }
        }
        // wave_sums_30: lanes l and l + 32, then rows of 16 lanes, then inside a row pairs, quads, the quad 12 lanes on, the half 8 lanes on (DPP
        // row_ror); then the eight waves in order
        float tot[30];
        for (int k = 0; k < kAcc; k++) {
            float v = 0.f;
            for (int w = 0; w < kBlock / 64; w++) {
                float y[16];
                for (int j = 0; j < 16; j++) {
                    auto X = [&](int l) { return acc[(size_t) (64 * w + l) * kAcc + k]; };
                    y[j] = (X(j) + X(j + 32)) + (X(j + 16) + X(j + 48));
                }
                float q[4];
                for (int g = 0; g < 4; g++) q[g] = (y[4 * g] + y[4 * g + 1]) + (y[4 * g + 2] + y[4 * g + 3]);
                v += (q[0] + q[3]) + (q[2] + q[1]);
            }
            tot[k] = v;
        }
        std::memset(H_, 0, sizeof(H_));
        int t = 0;
        for (int r = 0; r < 6; r++)
            for (int c = r; c < 6; c++, t++) H_[6 * r + c] = H_[6 * c + r] = tot[t];
        for (int k = 0; k < 6; k++) Jres_[k] = tot[21 + k];
        n_meas_ = (size_t) tot[28];
        return tot[27] / tot[28];
    }

    float computeResiduals(const SE3f &T_cur_from_ref, bool linearize_system) {
        const Image &cur_img = *cur.pyramid[level_];
        if (!have_ref_patch_cache_) precomputeReferencePatches();
        if (device_order_ && linearize_system) return computeResidualsDeviceOrder(T_cur_from_ref);
        const int stride = cur_img.w;
        const int border = patch_halfsize_ + 1;
        const float scale = ref.invScaleFactors[level_];
        float chi2 = 0.0;
        for (int i = 0; i < ref.N; i++) {
            if (!visible_fts_[i]) continue;
            float xyz_ref[3], xyz_cur[3];
            ref.Tcw.Act(&ref.mp_world[3 * i], xyz_ref);
            T_cur_from_ref.Act(xyz_ref, xyz_cur);
            const float ucx = cur.fx * xyz_cur[0] / xyz_cur[2] + cur.cx;  // Frame::Camera2Pixel (Frame.h:154-159)
            const float ucy = cur.fy * xyz_cur[1] / xyz_cur[2] + cur.cy;
            const float u_cur = ucx * scale, v_cur = ucy * scale;
            const int u_cur_i = (int) floorf(u_cur), v_cur_i = (int) floorf(v_cur);
            if (u_cur_i < 0 || v_cur_i < 0 || u_cur_i - border < 0 || v_cur_i - border < 0 ||
                u_cur_i + border >= cur_img.w || v_cur_i + border >= cur_img.h)
                continue;
            const float subpix_u_cur = u_cur - u_cur_i, subpix_v_cur = v_cur - v_cur_i;
            const float w_cur_tl = (float) ((1.0 - subpix_u_cur) * (1.0 - subpix_v_cur));
            const float w_cur_tr = (float) (subpix_u_cur * (1.0 - subpix_v_cur));
            const float w_cur_bl = (float) ((1.0 - subpix_u_cur) * subpix_v_cur);
            const float w_cur_br = subpix_u_cur * subpix_v_cur;
            const float *ref_patch_cache_ptr = &ref_patch_cache_[(size_t) patch_area_ * i];
            size_t pixel_counter = 0;
            for (int y = 0; y < patch_size_; ++y) {
                const uint8_t *p = &cur_img.d[(size_t) (v_cur_i + y - patch_halfsize_) * stride + (u_cur_i - patch_halfsize_)];
                for (int x = 0; x < patch_size_; ++x, ++pixel_counter, ++p, ++ref_patch_cache_ptr) {
                    const float intensity_cur = w_cur_tl * p[0] + w_cur_tr * p[1] + w_cur_bl * p[stride] + w_cur_br * p[stride + 1];
                    const float res = intensity_cur - (*ref_patch_cache_ptr);
                    const float weight = 1.0;
                    chi2 += res * res * weight;
                    n_meas_++;
                    if (linearize_system) {
                        const float *J = &jacobian_cache_[((size_t) i * patch_area_ + pixel_counter) * 6];
                        for (int r = 0; r < 6; r++)
                            for (int c = 0; c < 6; c++) H_[6 * r + c] += J[r] * J[c] * weight;
                        for (int r = 0; r < 6; r++) Jres_[r] -= J[r] * res * weight;
                    }
                }
            }
        }
        return chi2 / n_meas_;
    }

    // NLSSolver_impl.hpp:17-91
    void optimizeGaussNewton(SE3f &model) {
        SE3f old_model = model;
        for (iter_ = 0; iter_ < n_iter_; ++iter_) {
            std::memset(H_, 0, sizeof(H_));
            std::memset(Jres_, 0, sizeof(Jres_));
            n_meas_ = 0;
            float new_chi2 = computeResiduals(model, true);
            iters_total++;
            ldlt_solve6(H_, Jres_, x_);            // solve(): x_ = H_.ldlt().solve(Jres_)
            if (std::isnan(x_[0])) stop_ = true;   // :235-236 -> "Matrix is close to singular"
            if ((iter_ > 0 && new_chi2 > 1.2 * chi2_) || stop_) {
                model = old_model;
                break;
            }
            float negx[6];
            for (int k = 0; k < 6; k++) negx[k] = -x_[k];
            SE3f new_model = model.Mul(SE3f::Exp(negx));  // update(): T_new = T_old * exp(-x)
            old_model = model;
            model = new_model;
            chi2_ = new_chi2;
            float nm = 0;
            for (int k = 0; k < 6; k++) nm = std::max(nm, std::fabs(x_[k]));
            if (nm <= eps_) break;
        }
    }
};
}  // namespace

AlignResult sparse_img_align(const AlignFrame &ref, const AlignFrame &cur, int max_level, int min_level, int n_iter, bool device_order) {
    AlignResult out;
    std::memset(out.H, 0, sizeof(out.H));
    if (ref.N == 0) return out;  // :24-27
    Aligner A(ref, cur);
    A.device_order_ = device_order;
    if (device_order) {
        A.dx_cache_.assign((size_t) ref.N * 16, 0.f);
        A.dy_cache_.assign((size_t) ref.N * 16, 0.f);
        A.mom_cache_.assign((size_t) ref.N * 3, 0.f);
    }
    A.ref_patch_cache_.assign((size_t) ref.N * 16, 0.f);
    A.jacobian_cache_.assign((size_t) ref.N * 16 * 6, 0.f);
    A.visible_fts_.assign(ref.N, 0);
    SE3f T_cur_from_ref = cur.Tcw.Mul(ref.Tcw.Inverse());
    for (int level = max_level; level >= min_level; level -= 1) {
        A.level_ = level;
        std::fill(A.jacobian_cache_.begin(), A.jacobian_cache_.end(), 0.f);
        A.have_ref_patch_cache_ = false;
        A.n_iter_ = n_iter;  // reference indexes iterations[level_] = 10 (OOB for level >= 6; defined as n_iter)
        A.optimizeGaussNewton(T_cur_from_ref);
    }
    out.TCR = T_cur_from_ref;
    out.ret = A.n_meas_ / 16;
    out.iters_total = A.iters_total;
    out.chi2 = A.chi2_;
    std::memcpy(out.H, A.H_, sizeof(out.H));
    return out;
}


// ---- fp64 evaluation of the same Gauss-Newton: the THIRD PARTY of the tolerance argument (test infrastructure) ---------------------------------
// The device (fp32, per-feature moments, tree reduction) and the reference-order mode above (fp32, pixel by pixel) are two roundings of one
// algorithm; on ill-conditioned scenes they differ from each other by more than north_star's 1e-5 and neither is "the" answer.  This is the same
// algorithm -- src/SparseImageAlign.cc:20-244 + NLSSolver_impl.hpp:17-91, same features, same visibility / border tests, same stop and rollback
// rules, T <- T * exp(-x) -- with EVERY quantity in double: poses, projections, interpolation weights, patches, Jacobians, the sums of the normal
// equations, the pivoted LDL^T, exp.  Inputs are the same floats widened.  tests/test_gpu_fuzz.py reports |device - fp64| beside
// |reference_order - fp64| per case: the device is held to being no further from this than the reference's own arithmetic is.
namespace f64 {
struct SE3d {
    double q[4] = {0, 0, 0, 1}, t[3] = {0, 0, 0};
};
static void qmul(const double a[4], const double b[4], double o[4]) {
    double r[4];
    r[3] = a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2];
    r[0] = a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1];
    r[1] = a[3] * b[1] + a[1] * b[3] + a[2] * b[0] - a[0] * b[2];
    r[2] = a[3] * b[2] + a[2] * b[3] + a[0] * b[1] - a[1] * b[0];
    std::memcpy(o, r, sizeof r);
}
static void qnorm(double q[4]) {
    const double n = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    for (int i = 0; i < 4; i++) q[i] /= n;
}
static void qrot(const double q[4], const double v[3], double o[3]) {
    double uv[3] = {q[1] * v[2] - q[2] * v[1], q[2] * v[0] - q[0] * v[2], q[0] * v[1] - q[1] * v[0]};
    for (int i = 0; i < 3; i++) uv[i] += uv[i];
    const double c[3] = {q[1] * uv[2] - q[2] * uv[1], q[2] * uv[0] - q[0] * uv[2], q[0] * uv[1] - q[1] * uv[0]};
    for (int i = 0; i < 3; i++) o[i] = v[i] + q[3] * uv[i] + c[i];
}
static void act(const SE3d &T, const double p[3], double o[3]) {
    double r[3];
    qrot(T.q, p, r);
    for (int i = 0; i < 3; i++) o[i] = r[i] + T.t[i];
}
static SE3d inverse(const SE3d &T) {
    SE3d o;
    o.q[0] = -T.q[0]; o.q[1] = -T.q[1]; o.q[2] = -T.q[2]; o.q[3] = T.q[3];
    qnorm(o.q);
    const double nt[3] = {-T.t[0], -T.t[1], -T.t[2]};
    qrot(o.q, nt, o.t);
    return o;
}
static SE3d mul(const SE3d &a, const SE3d &b) {
    SE3d r = a;
    double rt[3];
    qrot(a.q, b.t, rt);
    for (int i = 0; i < 3; i++) r.t[i] += rt[i];
    qmul(a.q, b.q, r.q);
    qnorm(r.q);
    return r;
}
static SE3d expd(const double a[6]) {   // se3.hpp:406-428 in double (series below 1e-10 of rotation)
    const double *w = a + 3;
    const double th2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2], th = std::sqrt(th2);
    SE3d r;
    double imag, real, c1, c2;
    if (th < 1e-10) {
        imag = 0.5 - th2 / 48.0; real = 1.0 - th2 / 8.0; c1 = 0.5; c2 = 1.0 / 6.0;
    } else {
        imag = std::sin(0.5 * th) / th; real = std::cos(0.5 * th);
        c1 = (1.0 - std::cos(th)) / th2; c2 = (th - std::sin(th)) / (th2 * th);
    }
    r.q[3] = real; r.q[0] = imag * w[0]; r.q[1] = imag * w[1]; r.q[2] = imag * w[2];
    qnorm(r.q);
    const double O[9] = {0, -w[2], w[1], w[2], 0, -w[0], -w[1], w[0], 0};
    double O2[9], V[9];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) O2[3 * i + j] = O[3 * i] * O[j] + O[3 * i + 1] * O[3 + j] + O[3 * i + 2] * O[6 + j];
    for (int i = 0; i < 9; i++) V[i] = ((i % 4 == 0) ? 1.0 : 0.0) + c1 * O[i] + c2 * O2[i];
    for (int i = 0; i < 3; i++) r.t[i] = V[3 * i] * a[0] + V[3 * i + 1] * a[1] + V[3 * i + 2] * a[2];
    return r;
}
static void ldlt6(const double Hin[36], const double bin[6], double x[6]) {   // the same pivoted LDL^T as ldlt_solve6, in double
    double A[36], b[6];
    int perm[6];
    std::memcpy(A, Hin, sizeof A);
    std::memcpy(b, bin, sizeof b);
    for (int i = 0; i < 6; i++) perm[i] = i;
    for (int k = 0; k < 6; k++) {
        int p = k;
        double best = std::fabs(A[7 * k]);
        for (int i = k + 1; i < 6; i++)
            if (std::fabs(A[7 * i]) > best) { best = std::fabs(A[7 * i]); p = i; }
        if (p != k) {
            for (int j = 0; j < 6; j++) std::swap(A[6 * k + j], A[6 * p + j]);
            for (int j = 0; j < 6; j++) std::swap(A[6 * j + k], A[6 * j + p]);
            std::swap(b[k], b[p]);
            std::swap(perm[k], perm[p]);
        }
        const double d = A[7 * k];
        for (int i = k + 1; i < 6; i++) {
            const double l = A[6 * i + k] / d;
            for (int j = k + 1; j < 6; j++) A[6 * i + j] -= l * A[6 * k + j];
            A[6 * i + k] = l;
        }
    }
    double y[6], z[6];
    for (int i = 0; i < 6; i++) {
        double s = b[i];
        for (int j = 0; j < i; j++) s -= A[6 * i + j] * y[j];
        y[i] = s;
    }
    for (int i = 0; i < 6; i++) y[i] /= A[7 * i];
    for (int i = 5; i >= 0; i--) {
        double s = y[i];
        for (int j = i + 1; j < 6; j++) s -= A[6 * j + i] * z[j];
        z[i] = s;
    }
    for (int i = 0; i < 6; i++) x[perm[i]] = z[i];
}
static void jac(const double p[3], double J[12]) {   // SparseImageAlign.h:90-111
    const double x = p[0], y = p[1], zi = 1.0 / p[2], zi2 = zi * zi;
    J[0] = -zi; J[1] = 0; J[2] = x * zi2; J[3] = y * J[2]; J[4] = -(1.0 + x * J[2]); J[5] = y * zi;
    J[6] = 0; J[7] = -zi; J[8] = y * zi2; J[9] = 1.0 + y * J[8]; J[10] = -J[3]; J[11] = -x * zi;
}
}  // namespace f64

AlignResultF64 sparse_img_align_f64(const AlignFrame &ref, const AlignFrame &cur, int max_level, int min_level, int n_iter) {
    using namespace f64;
    AlignResultF64 out;
    if (ref.N == 0) return out;
    auto widen = [](const SE3f &T) { SE3d o; for (int i = 0; i < 4; i++) o.q[i] = T.q[i]; for (int i = 0; i < 3; i++) o.t[i] = T.t[i]; return o; };
    const SE3d Tref = widen(ref.Tcw);
    SE3d T = mul(widen(cur.Tcw), inverse(Tref));
    const int N = ref.N, hp = 2, border = 3;
    std::vector<double> patch((size_t) N * 16), Jc((size_t) N * 16 * 6);
    std::vector<uint8_t> vis(N, 0);                 // (visible_fts_ is never cleared between levels in the reference either)
    double chi2_ = 1e10;
    size_t n_meas = 0;
    int iters_total = 0;
    bool stop = false;
    for (int level = max_level; level >= min_level; level--) {
        const Image &ri = *ref.pyramid[level], &ci = *cur.pyramid[level];
        const double scale = ref.invScaleFactors[level];
        std::fill(Jc.begin(), Jc.end(), 0.0);
        for (int i = 0; i < N; i++) {                                   // precomputeReferencePatches, :57-128
            if (!ref.mp_valid[i] || ref.outlier[i]) continue;
            const double u = (double) ref.keys[i].x * scale, v = (double) ref.keys[i].y * scale;
            const int ui = (int) std::floor(u), vi = (int) std::floor(v);
            if (ui - border < 0 || vi - border < 0 || ui + border >= ri.w || vi + border >= ri.h) continue;
            vis[i] = 1;
            const double pw[3] = {ref.mp_world[3 * i], ref.mp_world[3 * i + 1], ref.mp_world[3 * i + 2]};
            double xr[3], J[12];
            act(Tref, pw, xr);
            jac(xr, J);
            const double su = u - ui, sv = v - vi, wtl = (1 - su) * (1 - sv), wtr = su * (1 - sv), wbl = (1 - su) * sv, wbr = su * sv;
            const int st = ri.w;
            for (int y = 0; y < 4; y++) {
                const uint8_t *p = &ri.d[(size_t) (vi + y - hp) * st + (ui - hp)];
                for (int x = 0; x < 4; x++, p++) {
                    const size_t pc = (size_t) i * 16 + 4 * y + x;
                    patch[pc] = wtl * p[0] + wtr * p[1] + wbl * p[st] + wbr * p[st + 1];
                    const double dx = 0.5 * ((wtl * p[1] + wtr * p[2] + wbl * p[st + 1] + wbr * p[st + 2]) - (wtl * p[-1] + wtr * p[0] + wbl * p[st - 1] + wbr * p[st]));
                    const double dy = 0.5 * ((wtl * p[st] + wtr * p[1 + st] + wbl * p[2 * st] + wbr * p[2 * st + 1]) - (wtl * p[-st] + wtr * p[1 - st] + wbl * p[0] + wbr * p[1]));
                    for (int k = 0; k < 6; k++) Jc[pc * 6 + k] = (dx * J[k] + dy * J[6 + k]) * ((double) ref.fx * scale);
                }
            }
        }
        SE3d old = T;                                                   // optimizeGaussNewton, NLSSolver_impl.hpp:17-91
        for (int iter = 0; iter < n_iter; iter++) {
            double H[36] = {0}, b[6] = {0}, chi2 = 0;
            n_meas = 0;
            const int st = ci.w;
            for (int i = 0; i < N; i++) {                               // computeResiduals, :130-231
                if (!vis[i]) continue;
                const double pw[3] = {ref.mp_world[3 * i], ref.mp_world[3 * i + 1], ref.mp_world[3 * i + 2]};
                double xr[3], xc[3];
                act(Tref, pw, xr);
                act(T, xr, xc);
                const double u = ((double) cur.fx * xc[0] / xc[2] + cur.cx) * scale, v = ((double) cur.fy * xc[1] / xc[2] + cur.cy) * scale;
                const int ui = (int) std::floor(u), vi = (int) std::floor(v);
                if (ui < 0 || vi < 0 || ui - border < 0 || vi - border < 0 || ui + border >= ci.w || vi + border >= ci.h) continue;
                const double su = u - ui, sv = v - vi, wtl = (1 - su) * (1 - sv), wtr = su * (1 - sv), wbl = (1 - su) * sv, wbr = su * sv;
                for (int y = 0; y < 4; y++) {
                    const uint8_t *p = &ci.d[(size_t) (vi + y - hp) * st + (ui - hp)];
                    for (int x = 0; x < 4; x++, p++) {
                        const size_t pc = (size_t) i * 16 + 4 * y + x;
                        const double res = (wtl * p[0] + wtr * p[1] + wbl * p[st] + wbr * p[st + 1]) - patch[pc];
                        chi2 += res * res;
                        n_meas++;
                        const double *J = &Jc[pc * 6];
                        for (int r = 0; r < 6; r++)
                            for (int c = 0; c < 6; c++) H[6 * r + c] += J[r] * J[c];
                        for (int r = 0; r < 6; r++) b[r] -= J[r] * res;
                    }
                }
            }
            const double new_chi2 = chi2 / (double) n_meas;
            iters_total++;
            double x[6];
            ldlt6(H, b, x);
            if (std::isnan(x[0])) stop = true;
            if ((iter > 0 && new_chi2 > 1.2 * chi2_) || stop) { T = old; break; }
            double nx[6];
            for (int k = 0; k < 6; k++) nx[k] = -x[k];
            const SE3d Tn = mul(T, expd(nx));
            old = T;
            T = Tn;
            chi2_ = new_chi2;
            double nm = 0;
            for (int k = 0; k < 6; k++) nm = std::max(nm, std::fabs(x[k]));
            if (nm <= 0.000001) break;                                  // eps_ (float 1e-6 in the reference)
        }
    }
    for (int i = 0; i < 4; i++) out.T[i] = T.q[i];
    for (int i = 0; i < 3; i++) out.T[4 + i] = T.t[i];
    out.ret = n_meas / 16;
    out.iters_total = iters_total;
    out.chi2 = chi2_;
    return out;
}

}  // namespace ygzo
