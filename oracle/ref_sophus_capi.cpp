// oracle/ref_sophus_capi.cpp -- TEST INFRASTRUCTURE.  C entry points over the REFERENCE'S OWN Thirdparty/sophus/sophus/se3.hpp / so3.hpp, included
// where they lie (nothing is copied) on top of oracle/ref_shim/eigen_min (the Eigen slice they need; Eigen itself is not in the checkout).
// tests/test_ref_sophus.py holds the oracle's restatement of the pose algebra (oracle_align.cpp: SE3f::Exp, Mul, Inverse, Act -- what the aligner
// pin and the HIP kernel's se3_device.h are compared with) to these functions bit for bit.  Built by `make -C oracle ref_sophus` into
// oracle/_ref/libref_sophus.so.
#include <sophus/se3.hpp>

#include <cstring>

namespace {
typedef Sophus::SE3f SE3;
SE3 from7(const float *p) {   // (qx, qy, qz, qw, tx, ty, tz): built member-wise, no normalisation on the way in
    SE3 T;
    T.so3() = Sophus::SO3f();
    float *q = T.so3().data();
    q[0] = p[0]; q[1] = p[1]; q[2] = p[2]; q[3] = p[3];
    T.translation() = Eigen::Vector3f(p[4], p[5], p[6]);
    return T;
}
void to7(const SE3 &T, float *o) {
    const float *q = T.so3().data();
    o[0] = q[0]; o[1] = q[1]; o[2] = q[2]; o[3] = q[3];
    o[4] = T.translation()[0]; o[5] = T.translation()[1]; o[6] = T.translation()[2];
}
}  // namespace

extern "C" {
// SE3f::exp (se3.hpp:406-428 + so3.hpp:425-456)
void ref_sophus_exp(const float *a6, float *out7) {
    Sophus::SE3f::Tangent a;
    for (int i = 0; i < 6; i++) a[i] = a6[i];
    to7(Sophus::SE3f::exp(a), out7);
}
// operator* = fastMultiply + normalize (se3.hpp:159-163, 267-271)
void ref_sophus_mul(const float *a7, const float *b7, float *out7) { to7(from7(a7) * from7(b7), out7); }
// inverse (se3.hpp:168-172)
void ref_sophus_inverse(const float *a7, float *out7) { to7(from7(a7).inverse(), out7); }
// action on a point (se3.hpp: operator*(Point))
void ref_sophus_act(const float *a7, const float *p3, float *out3) {
    const Eigen::Vector3f r = from7(a7) * Eigen::Vector3f(p3[0], p3[1], p3[2]);
    out3[0] = r[0]; out3[1] = r[1]; out3[2] = r[2];
}
void ref_sophus_rotation_matrix(const float *a7, float *R9) {
    const Eigen::Matrix3f R = from7(a7).rotationMatrix();
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) R9[3 * i + j] = R(i, j);
}
}
