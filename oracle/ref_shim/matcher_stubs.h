// oracle/ref_shim/matcher_stubs.h -- TEST INFRASTRUCTURE (included by mini_cv.h when YGZ_REF_MATCHER is defined).
// What the reference's src/ORBmatcher.cc needs from Common.h (Eigen, Sophus), Frame.h, KeyFrame.h, MapPoint.h, Align.h and Converter.h,
// reduced to plain data + the handful of methods the file calls, so that the reference's own matcher source compiles where it lies.
// The real headers are switched off through their include guards.  Arithmetic stand-ins follow the oracle's conventions (3x3 products
// row by row, left to right); methods that only the out-of-scope functions (Fuse, Sim3, triangulation) call abort when reached.
#ifndef YGZ_ORACLE_REF_SHIM_MATCHER_STUBS_H
#define YGZ_ORACLE_REF_SHIM_MATCHER_STUBS_H

#define YGZ_FRAME_H_
#define YGZ_KEYFRAME_H_
#define YGZ_MAPPOINT_H
#define YGZ_ALIGN_H_
#define YGZ_CONVERTER_H_

#include <cstdlib>
#include <nmmintrin.h>   // _mm_popcnt_u64 (ORBmatcher::DescriptorDistance, src/ORBmatcher.cc:1515)

using namespace std;   // include/Common.h:19

[[noreturn]] inline void yr_unsupported(const char *what) {
    std::fprintf(stderr, "oracle/ref_shim: %s is outside the pinned path\n", what);
    std::abort();
}

namespace Eigen {
struct Vector2f {
    float v[2];
    Vector2f() { v[0] = v[1] = 0; }
    Vector2f(float a, float b) { v[0] = a; v[1] = b; }
    float &operator[](int i) { return v[i]; }
    float operator[](int i) const { return v[i]; }
    float &operator()(int i) { return v[i]; }
    float operator()(int i) const { return v[i]; }
    Vector2f &operator*=(float s) { v[0] *= s; v[1] *= s; return *this; }
};
inline Vector2f operator+(const Vector2f &a, const Vector2f &b) { return Vector2f(a[0] + b[0], a[1] + b[1]); }
inline Vector2f operator-(const Vector2f &a, const Vector2f &b) { return Vector2f(a[0] - b[0], a[1] - b[1]); }
inline Vector2f operator*(const Vector2f &a, float s) { return Vector2f(a[0] * s, a[1] * s); }
inline Vector2f operator/(const Vector2f &a, float s) { return Vector2f(a[0] / s, a[1] / s); }

struct Vector3f {
    float v[3];
    Vector3f() { v[0] = v[1] = v[2] = 0; }
    Vector3f(float a, float b, float c) { v[0] = a; v[1] = b; v[2] = c; }
    float &operator[](int i) { return v[i]; }
    float operator[](int i) const { return v[i]; }
    float &operator()(int i) { return v[i]; }
    float operator()(int i) const { return v[i]; }
    float dot(const Vector3f &o) const { return v[0] * o[0] + v[1] * o[1] + v[2] * o[2]; }
    float norm() const { return std::sqrt(dot(*this)); }
};
inline Vector3f operator+(const Vector3f &a, const Vector3f &b) { return Vector3f(a[0] + b[0], a[1] + b[1], a[2] + b[2]); }
inline Vector3f operator-(const Vector3f &a, const Vector3f &b) { return Vector3f(a[0] - b[0], a[1] - b[1], a[2] - b[2]); }

struct Matrix3f {
    float m[9];   // row major
    Matrix3f() { for (float &x : m) x = 0; }
    float &operator()(int r, int c) { return m[3 * r + c]; }
    float operator()(int r, int c) const { return m[3 * r + c]; }
    Matrix3f transpose() const { Matrix3f t; for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) t(r, c) = (*this)(c, r); return t; }
};
inline Vector3f operator*(const Matrix3f &A, const Vector3f &x) {
    return Vector3f(A(0, 0) * x[0] + A(0, 1) * x[1] + A(0, 2) * x[2], A(1, 0) * x[0] + A(1, 1) * x[1] + A(1, 2) * x[2],
                    A(2, 0) * x[0] + A(2, 1) * x[1] + A(2, 2) * x[2]);
}
inline Matrix3f operator*(float s, const Matrix3f &A) { Matrix3f r; for (int i = 0; i < 9; i++) r.m[i] = s * A.m[i]; return r; }
inline Matrix3f operator*(int s, const Matrix3f &A) { return (float) s * A; }

struct Matrix2f {   // only FindDirectProjection's warp uses it (not pinned)
    float m[4];
    Matrix2f() { for (float &x : m) x = 0; }
    struct ColRef { Matrix2f *M; int c; void operator=(const Vector2f &v) { M->m[c] = v[0]; M->m[2 + c] = v[1]; } };
    ColRef col(int c) { return ColRef{this, c}; }
    Matrix2f inverse() const { yr_unsupported("Matrix2f::inverse"); }
    float determinant() const { yr_unsupported("Matrix2f::determinant"); }
};
inline Vector2f operator*(const Matrix2f &, const Vector2f &) { yr_unsupported("Matrix2f * Vector2f"); }
}  // namespace Eigen
using Eigen::Matrix2f;
using Eigen::Matrix3f;
using Eigen::Vector2f;
using Eigen::Vector3f;

namespace Sophus {
struct SE3f {   // rotation matrix + translation, as the matcher reads them
    Matrix3f R;
    Vector3f t;
    SE3f() { R(0, 0) = R(1, 1) = R(2, 2) = 1.f; }
    Matrix3f rotationMatrix() const { return R; }
    Vector3f translation() const { return t; }
    SE3f inverse() const { yr_unsupported("SE3f::inverse"); }
};
inline SE3f operator*(const SE3f &, const SE3f &) { yr_unsupported("SE3f * SE3f"); }
inline Vector3f operator*(const SE3f &T, const Vector3f &x) { return T.R * x + T.t; }
}  // namespace Sophus
using Sophus::SE3f;

namespace ygz {

class Frame;
class KeyFrame;

// include/MapPoint.h, src/MapPoint.cc: the fields / accessors the matcher touches
class MapPoint {
public:
    Vector3f mWorldPos, mNormal;
    cv::Mat mDescriptor;                       // 1 x 32
    bool mbBad = false;
    int nObs = 0;
    float mfMaxDistance = 0;                   // PredictScale's numerator
    float minDistInv = 0, maxDistInv = 0;      // what GetMin/MaxDistanceInvariance() return (0.8 * mfMinDistance, 1.2 * mfMaxDistance)
    // set by Frame::isInFrustum (src/Frame.cc:363-422)
    float mTrackProjX = 0, mTrackProjY = 0, mTrackProjXR = 0, mTrackViewCos = 0;
    int mnTrackScaleLevel = 0;
    bool mbTrackInView = false;
    long unsigned int mnId = 0;
    long unsigned int mnFuseCandidateForKF = 0;

    Vector3f GetWorldPos() { return mWorldPos; }
    Vector3f GetNormal() { return mNormal; }
    cv::Mat GetDescriptor() { return mDescriptor; }
    bool isBad() { return mbBad; }
    int Observations() { return nObs; }
    std::map<KeyFrame *, size_t> GetObservations() { yr_unsupported("MapPoint::GetObservations"); }
    float GetMinDistanceInvariance() { return minDistInv; }               // src/MapPoint.cc:347-350
    float GetMaxDistanceInvariance() { return maxDistInv; }               // :352-355
    int PredictScale(const float &currentDist, KeyFrame *pKF);           // :359-373 (body in ref_orbmatcher_capi.cpp)
    int PredictScale(const float &currentDist, Frame *pF);
    bool IsInKeyFrame(KeyFrame *) { yr_unsupported("MapPoint::IsInKeyFrame"); }
    int GetIndexInKeyFrame(KeyFrame *) { yr_unsupported("MapPoint::GetIndexInKeyFrame"); }
    void Replace(MapPoint *) { yr_unsupported("MapPoint::Replace"); }
    void AddObservation(KeyFrame *, size_t) { yr_unsupported("MapPoint::AddObservation"); }
};

// include/Frame.h: data members the matcher reads + GetFeaturesInArea (src/Frame.cc:424-481, body = the oracle's restatement)
class Frame {
public:
    int N = 0;
    std::vector<cv::KeyPoint> mvKeys;
    std::vector<float> mvuRight;
    cv::Mat mDescriptors;                      // N x 32
    std::vector<MapPoint *> mvpMapPoints;
    std::vector<bool> mvbOutlier;
    SE3f mTcw;
    float fx = 0, fy = 0, cx = 0, cy = 0, mb = 0, mbf = 0;
    float mnMinX = 0, mnMaxX = 0, mnMinY = 0, mnMaxY = 0;
    int mnScaleLevels = 0;
    float mfLogScaleFactor = 0;
    std::vector<float> mvScaleFactors, mvInvScaleFactors, mvLevelSigma2, mvInvLevelSigma2;
    std::vector<cv::Mat> mvImagePyramid;
    DBoW2::FeatureVector mFeatVec;
    void *grid = nullptr;                      // ygzo::Grid + FrameView of this frame (ref_orbmatcher_capi.cpp)

    std::vector<size_t> GetFeaturesInArea(const float &x, const float &y, const float &r, const int minLevel = -1, const int maxLevel = -1) const;
    Vector2f World2Pixel(const Vector3f &, const SE3f &) { yr_unsupported("Frame::World2Pixel"); }
};

// include/KeyFrame.h
class KeyFrame {
public:
    long unsigned int mnId = 0;
    int N = 0;
    std::vector<cv::KeyPoint> mvKeys, mvKeysUn;
    std::vector<float> mvuRight;
    cv::Mat mDescriptors;
    DBoW2::FeatureVector mFeatVec;
    std::vector<MapPoint *> mvpMapPoints;
    float fx = 0, fy = 0, cx = 0, cy = 0, mbf = 0, mb = 0;
    int mnScaleLevels = 0;
    float mfLogScaleFactor = 0;
    std::vector<float> mvScaleFactors, mvLevelSigma2, mvInvLevelSigma2;
    std::vector<cv::Mat> mvImagePyramid;

    std::vector<MapPoint *> GetMapPointMatches() { return mvpMapPoints; }
    MapPoint *GetMapPoint(const size_t &i) { return mvpMapPoints[i]; }
    std::set<MapPoint *> GetMapPoints() { yr_unsupported("KeyFrame::GetMapPoints"); }
    void AddMapPoint(MapPoint *, const size_t &) { yr_unsupported("KeyFrame::AddMapPoint"); }
    Matrix3f GetRotation() { yr_unsupported("KeyFrame::GetRotation"); }
    Vector3f GetTranslation() { yr_unsupported("KeyFrame::GetTranslation"); }
    Vector3f GetCameraCenter() { yr_unsupported("KeyFrame::GetCameraCenter"); }
    SE3f GetPose() const { yr_unsupported("KeyFrame::GetPose"); }
    bool IsInImage(const float &, const float &) const { yr_unsupported("KeyFrame::IsInImage"); }
    std::vector<size_t> GetFeaturesInArea(const float &, const float &, const float &) const { yr_unsupported("KeyFrame::GetFeaturesInArea"); }
    Vector3f Pixel2Camera(const Vector2f &, float) const { yr_unsupported("KeyFrame::Pixel2Camera"); }
};

// include/Converter.h: only the Sim3 / fuse functions use it
class Converter {
public:
    static cv::Mat toCvMat(const Vector3f &) { yr_unsupported("Converter::toCvMat"); }
    static cv::Mat toCvMat(const Matrix3f &) { yr_unsupported("Converter::toCvMat"); }
};

// include/Align.h
inline bool Align2D(const cv::Mat &, uint8_t *, uint8_t *, const int, Vector2f &, bool = false) { yr_unsupported("Align2D"); }

}  // namespace ygz
#endif
