// se3_device.h -- Sophus SE3f operations restated for the device (Thirdparty/sophus/sophus/se3.hpp:159-171,267-271,406-428,
// so3.hpp:234-237,268-276,425-456; Eigen quaternion product / _transformVector / toRotationMatrix), fp32, source-order arithmetic.
// Shared by align_kernels.hip (SparseImgAlign) and direct_kernels.hip (FindDirectProjection).
#ifndef YGZF_SE3_DEVICE_H
#define YGZF_SE3_DEVICE_H
#include <hip/hip_runtime.h>

namespace ygzf {

constexpr float kSophusEps = 1e-5f;

struct Se3 { float q[4]; float t[3]; };  // quaternion x,y,z,w + translation

__device__ __forceinline__ void quat_mul(const float a[4], const float b[4], float o[4]) {
    float r[4];
    r[3] = a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2];
    r[0] = a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1];
    r[1] = a[3] * b[1] + a[1] * b[3] + a[2] * b[0] - a[0] * b[2];
    r[2] = a[3] * b[2] + a[2] * b[3] + a[0] * b[1] - a[1] * b[0];
    o[0] = r[0]; o[1] = r[1]; o[2] = r[2]; o[3] = r[3];
}
__device__ __forceinline__ void quat_normalize(float q[4]) {
    const float n = sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    q[0] /= n; q[1] /= n; q[2] /= n; q[3] /= n;
}
__device__ __forceinline__ void quat_rotate(const float q[4], const float v[3], float o[3]) {  // Eigen _transformVector
    float uv[3] = {q[1] * v[2] - q[2] * v[1], q[2] * v[0] - q[0] * v[2], q[0] * v[1] - q[1] * v[0]};
    uv[0] += uv[0]; uv[1] += uv[1]; uv[2] += uv[2];
    const float c[3] = {q[1] * uv[2] - q[2] * uv[1], q[2] * uv[0] - q[0] * uv[2], q[0] * uv[1] - q[1] * uv[0]};
    o[0] = v[0] + q[3] * uv[0] + c[0];
    o[1] = v[1] + q[3] * uv[1] + c[1];
    o[2] = v[2] + q[3] * uv[2] + c[2];
}
__device__ __forceinline__ void se3_act(const Se3 &T, const float p[3], float o[3]) {
    float r[3];
    quat_rotate(T.q, p, r);
    o[0] = r[0] + T.t[0]; o[1] = r[1] + T.t[1]; o[2] = r[2] + T.t[2];
}
__device__ __forceinline__ Se3 se3_inverse(const Se3 &T) {
    Se3 o;
    o.q[0] = -T.q[0]; o.q[1] = -T.q[1]; o.q[2] = -T.q[2]; o.q[3] = T.q[3];
    quat_normalize(o.q);
    const float nt[3] = {T.t[0] * -1.f, T.t[1] * -1.f, T.t[2] * -1.f};
    quat_rotate(o.q, nt, o.t);
    return o;
}
__device__ __forceinline__ Se3 se3_mul(const Se3 &a, const Se3 &b) {  // fastMultiply + normalize
    Se3 r = a;
    float rt[3];
    quat_rotate(r.q, b.t, rt);
    r.t[0] += rt[0]; r.t[1] += rt[1]; r.t[2] += rt[2];
    quat_mul(r.q, b.q, r.q);
    quat_normalize(r.q);
    return r;
}
__device__ __forceinline__ void quat_to_R(const float q[4], float R[9]) {
    const float tx = 2 * q[0], ty = 2 * q[1], tz = 2 * q[2];
    const float twx = tx * q[3], twy = ty * q[3], twz = tz * q[3];
    const float txx = tx * q[0], txy = ty * q[0], txz = tz * q[0];
    const float tyy = ty * q[1], tyz = tz * q[1], tzz = tz * q[2];
    R[0] = 1 - (tyy + tzz); R[1] = txy - twz;       R[2] = txz + twy;
    R[3] = txy + twz;       R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
    R[6] = txz - twy;       R[7] = tyz + twx;       R[8] = 1 - (txx + tyy);
}
__device__ Se3 se3_exp(const float a[6]) {  // se3.hpp:406-428
    const float *om = a + 3;
    const float theta_sq = om[0] * om[0] + om[1] * om[1] + om[2] * om[2];
    const float theta = sqrtf(theta_sq);
    const float half_theta = 0.5f * theta;
    float imag_factor, real_factor;
    if (theta < kSophusEps) {
        const float theta_po4 = theta_sq * theta_sq;
        imag_factor = 0.5f - (float) (1.0 / 48.0) * theta_sq + (float) (1.0 / 3840.0) * theta_po4;
        real_factor = 1.f - 0.5f * theta_sq + (float) (1.0 / 384.0) * theta_po4;
    } else {
        imag_factor = sinf(half_theta) / theta;
        real_factor = cosf(half_theta);
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
    if (theta < kSophusEps) {
        quat_to_R(r.q, V);
    } else {
        const float theta_sq2 = theta * theta;   // se3.hpp:419: the square of the root, not the squared norm above
        const float c1 = (1.f - cosf(theta)) / theta_sq2;
        const float c2 = (theta - sinf(theta)) / (theta_sq2 * theta);
        for (int i = 0; i < 9; i++) V[i] = ((i % 4 == 0) ? 1.f : 0.f) + c1 * O[i] + c2 * O2[i];
    }
    for (int i = 0; i < 3; i++) r.t[i] = V[3 * i] * a[0] + V[3 * i + 1] * a[1] + V[3 * i + 2] * a[2];
    return r;
}

}  // namespace ygzf
#endif
