// oracle/ref_shim/frame_stubs.h -- TEST INFRASTRUCTURE (YGZ_REF_FRAME build): what the REAL include/Frame.h and src/Frame.cc need beyond
// matcher_stubs.h -- IMU types, double-precision pose types, the vocabulary -- as declarations with aborting bodies: those code paths
// (IMU integration, constructors, BoW, undistortion) are compiled, never executed; the pinned functions are AssignFeaturesToGrid / PosInGrid /
// GetFeaturesInArea, isInFrustum and ComputeStereoMatches.
#ifndef YGZ_ORACLE_REF_SHIM_FRAME_STUBS_H
#define YGZ_ORACLE_REF_SHIM_FRAME_STUBS_H
#include "mini_cv.h"

namespace Eigen {
struct Vector3d {
    double v[3];
    Vector3d() { v[0] = v[1] = v[2] = 0; }
    static Vector3d Zero() { return Vector3d(); }
    double norm() const { return std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]); }
};
struct Matrix3d {
    double m[9];
    Matrix3d inverse() const { yr_unsupported("Matrix3d::inverse"); }
    Matrix3d transpose() const { yr_unsupported("Matrix3d::transpose"); }
};
inline Vector3d operator*(const Matrix3d &, const Vector3d &) { yr_unsupported("Matrix3d * v"); }
inline Matrix3d operator*(const Matrix3d &, const Matrix3d &) { yr_unsupported("Matrix3d * Matrix3d"); }
inline Vector3d operator*(const Vector3d &, double) { yr_unsupported("Vector3d * s"); }
inline Vector3d operator*(double, const Vector3d &) { yr_unsupported("s * Vector3d"); }
inline Vector3d operator+(const Vector3d &, const Vector3d &) { yr_unsupported("Vector3d +"); }
inline Vector3d operator-(const Vector3d &, const Vector3d &) { yr_unsupported("Vector3d -"); }
inline Vector3d operator-(const Vector3d &) { yr_unsupported("-Vector3d"); }
}  // namespace Eigen
using Eigen::Vector3d;
using Eigen::Matrix3d;
using namespace Eigen;   // the IMU headers do that for the real include/Frame.h (Matrix<double, 15, 15> mMargCovInv)

namespace Sophus {
struct SO3d {
    SO3d inverse() const { yr_unsupported("SO3d::inverse"); }
};
inline SO3d operator*(const SO3d &, const SO3d &) { yr_unsupported("SO3d * SO3d"); }
inline Vector3d operator*(const SO3d &, const Vector3d &) { yr_unsupported("SO3d * v"); }
struct SE3d {
    SO3d r; Vector3d p;
    SE3d() {}
    SE3d(const SO3d &, const Vector3d &) {}
    template <class A, class B> SE3d(const A &, const B &) { yr_unsupported("SE3d(R, t)"); }
    SE3d inverse() const { yr_unsupported("SE3d::inverse"); }
    const SO3d &so3() const { return r; }
    const Vector3d &translation() const { return p; }
    template <class T> SE3f cast() const { yr_unsupported("SE3d::cast"); }
};
}  // namespace Sophus
using Sophus::SE3d;
using Sophus::SO3d;
#ifdef YGZ_REF_TRACKING
namespace Sophus {
template <> struct SE3f::CastResult<double> { typedef SE3d type; };
inline SE3d operator*(const SE3d &, const SE3d &) { yr_unsupported("SE3d * SE3d"); }
struct SO3 {   // Sophus::SO3 (the non-templated double form of the IMU code)
    template <class M> SO3(const M &) { yr_unsupported("Sophus::SO3"); }
    SO3() {}
    SO3 inverse() const { yr_unsupported("SO3::inverse"); }
    template <class V> static SO3 exp(const V &) { yr_unsupported("SO3::exp"); }
    Vector3d log() const { yr_unsupported("SO3::log"); }
};
}  // namespace Sophus
#endif

namespace ygz {
struct IMUData { double _t = 0; Vector3d _g, _a; };
class NavState {
public:
    Vector3d Get_BiasGyr() const { yr_unsupported("NavState"); }
    Vector3d Get_BiasAcc() const { yr_unsupported("NavState"); }
    Vector3d Get_dBias_Gyr() const { yr_unsupported("NavState"); }
    Vector3d Get_dBias_Acc() const { yr_unsupported("NavState"); }
    void Set_BiasGyr(const Vector3d &) {}
    void Set_BiasAcc(const Vector3d &) {}
    void Set_DeltaBiasGyr(const Vector3d &) {}
    void Set_DeltaBiasAcc(const Vector3d &) {}
    SO3d Get_R() const { yr_unsupported("NavState"); }
    Vector3d Get_P() const { yr_unsupported("NavState"); }
    Vector3d Get_V() const { yr_unsupported("NavState"); }
    void Set_Pos(const Vector3d &) {}
    void Set_Vel(const Vector3d &) {}
    void Set_Rot(const SO3d &) {}
};
class IMUPreintegrator {
public:
    void reset() {}
    void update(const Vector3d &, const Vector3d &, double) {}
#ifdef YGZ_REF_TRACKING
    Vector3d getDeltaP() const { yr_unsupported("IMUPreintegrator"); }
    Vector3d getDeltaV() const { yr_unsupported("IMUPreintegrator"); }
    Matrix3d getDeltaR() const { yr_unsupported("IMUPreintegrator"); }
    double getDeltaTime() const { yr_unsupported("IMUPreintegrator"); }
    Matrix3d getJPBiasg() const { yr_unsupported("IMUPreintegrator"); }
    Matrix3d getJPBiasa() const { yr_unsupported("IMUPreintegrator"); }
    Matrix3d getJVBiasg() const { yr_unsupported("IMUPreintegrator"); }
    Matrix3d getJVBiasa() const { yr_unsupported("IMUPreintegrator"); }
    Matrix3d getJRBiasg() const { yr_unsupported("IMUPreintegrator"); }
#endif
};
// include/ORBVocabulary.h: DBoW2::TemplatedVocabulary<...>; Frame::ComputeBoW calls transform()
#if defined(YGZ_REAL_DBOW2)
// the reference's real include/ORBVocabulary.h (DBoW2::TemplatedVocabulary) is used: nothing to stand in for
#elif defined(YGZ_BOUNDARY_BUILD)
// boundary build (tests/cpp/build_boundary.sh: the reference's own Frame.cc over the PRODUCT's class shells): ExtractFeatures() ends in
// ComputeBoW(), so the vocabulary must be callable; the test driver supplies the body
class ORBVocabulary {
public:
    void transform(const std::vector<cv::Mat> &features, DBoW2::BowVector &v, DBoW2::FeatureVector &fv, int levelsup) const;
};
#else
class ORBVocabulary {
public:
    template <class D> void transform(const D &, DBoW2::BowVector &, DBoW2::FeatureVector &, int) const { yr_unsupported("ORBVocabulary::transform"); }
};
#endif
}  // namespace ygz
#endif
