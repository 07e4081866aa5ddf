// oracle/ref_shim/matcher_stubs.h -- TEST INFRASTRUCTURE (included by mini_cv.h when YGZ_REF_MATCHER is defined).
// What the reference's src/ORBmatcher.cc needs from Common.h (Eigen, Sophus), Frame.h, KeyFrame.h, MapPoint.h, Align.h and Converter.h,
// reduced to plain data + the handful of methods the file calls, so that the reference's own matcher source compiles where it lies.
// The real headers are switched off through their include guards.  Arithmetic stand-ins follow the oracle's conventions (3x3 products
// row by row, left to right); methods that only the out-of-scope functions (Fuse, Sim3, triangulation) call abort when reached.
#ifndef YGZ_ORACLE_REF_SHIM_MATCHER_STUBS_H
#define YGZ_ORACLE_REF_SHIM_MATCHER_STUBS_H

#ifndef YGZ_REF_FRAME      // the Frame build (tests the reference's own src/Frame.cc) keeps the real include/Frame.h
#define YGZ_FRAME_H_
#endif
#ifndef YGZ_REAL_DBOW2
#define YGZ_ORBVOCABULARY_H_
#endif
#define YGZ_KEYFRAME_H_
#ifndef YGZ_REF_MAPPOINT   // the MapPoint build (tests the reference's own src/MapPoint.cc) keeps the real include/MapPoint.h
#define YGZ_MAPPOINT_H
#endif
#define YGZ_MAP_H_
#define YGZ_ALIGN_H_
#define YGZ_CONVERTER_H_

#include <cstdlib>

#include "ygz_oracle.h"   // ygzo::SE3f (quaternion form, Sophus semantics), ygzo::inverse3: the oracle's conventions behind the stand-ins
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
    Vector2f &operator+=(const Vector2f &o) { v[0] += o.v[0]; v[1] += o.v[1]; return *this; }
    float x() const { return v[0]; }
    float y() const { return v[1]; }
    float operator()(int i, int) const { return v[i]; }
    struct CommaInit { Vector2f *p; CommaInit operator,(float b) { p->v[1] = b; return *this; } };
    CommaInit operator<<(float a) { v[0] = a; return CommaInit{this}; }   // `vec << u, v;`
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
    float operator()(int i, int) const { return v[i]; }
    void setZero() { v[0] = v[1] = v[2] = 0; }
    Vector3f &operator*=(double s) { v[0] = (float) (v[0] * s); v[1] = (float) (v[1] * s); v[2] = (float) (v[2] * s); return *this; }
    void normalize() { const float n = norm(); v[0] /= n; v[1] /= n; v[2] /= n; }   // Eigen: *this /= norm()
    struct Row { const Vector3f *p; };
    Row transpose() const { return Row{this}; }
};
inline Vector3f operator+(const Vector3f &a, const Vector3f &b) { return Vector3f(a[0] + b[0], a[1] + b[1], a[2] + b[2]); }
inline Vector3f operator-(const Vector3f &a, const Vector3f &b) { return Vector3f(a[0] - b[0], a[1] - b[1], a[2] - b[2]); }
inline Vector3f operator-(const Vector3f &a) { return Vector3f(-a[0], -a[1], -a[2]); }
inline Vector3f operator*(const Vector3f &a, float s) { return Vector3f(a[0] * s, a[1] * s, a[2] * s); }
inline Vector3f operator/(const Vector3f &a, float s) { return Vector3f(a[0] / s, a[1] / s, a[2] / s); }

struct Matrix3f {
    float m[9];   // row major
    Matrix3f() { for (float &x : m) x = 0; }
    float &operator()(int r, int c) { return m[3 * r + c]; }
    float operator()(int r, int c) const { return m[3 * r + c]; }
    static Matrix3f Zero() { return Matrix3f(); }
    static Matrix3f Identity() { Matrix3f r; r(0, 0) = r(1, 1) = r(2, 2) = 1.f; return r; }
    void setZero() { for (float &x : m) x = 0; }
    Matrix3f &operator+=(const Matrix3f &o) { for (int i = 0; i < 9; i++) m[i] += o.m[i]; return *this; }
    Matrix3f inverse() const { Matrix3f r; ygzo::inverse3(m, r.m); return r; }
    Matrix3f transpose() const { Matrix3f t; for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) t(r, c) = (*this)(c, r); return t; }
};
inline Vector3f operator*(const Matrix3f &A, const Vector3f &x) {
    return Vector3f(A(0, 0) * x[0] + A(0, 1) * x[1] + A(0, 2) * x[2], A(1, 0) * x[0] + A(1, 1) * x[1] + A(1, 2) * x[2],
                    A(2, 0) * x[0] + A(2, 1) * x[1] + A(2, 2) * x[2]);
}
inline Matrix3f operator*(const Vector3f &a, const Vector3f::Row &b) {   // outer product J * J^T
    Matrix3f r;
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) r(i, j) = a[i] * (*b.p)[j];
    return r;
}
inline Matrix3f operator*(float s, const Matrix3f &A) { Matrix3f r; for (int i = 0; i < 9; i++) r.m[i] = s * A.m[i]; return r; }
inline Matrix3f operator*(int s, const Matrix3f &A) { return (float) s * A; }

struct Matrix2f {   // FindDirectProjection's warp; determinant / inverse written as in oracle_direct.cpp (adjugate * (1 / det))
    float m[4];   // row major
    Matrix2f() { for (float &x : m) x = 0; }
    struct ColRef { Matrix2f *M; int c; void operator=(const Vector2f &v) { M->m[c] = v[0]; M->m[2 + c] = v[1]; } };
    ColRef col(int c) { return ColRef{this, c}; }
    float determinant() const { return m[0] * m[3] - m[2] * m[1]; }
    Matrix2f inverse() const {
        const float det = m[0] * m[3] - m[2] * m[1];
        const float invdet = 1.f / det;
        Matrix2f r;
        r.m[0] = m[3] * invdet; r.m[1] = -m[1] * invdet; r.m[2] = -m[2] * invdet; r.m[3] = m[0] * invdet;
        return r;
    }
};
inline Vector2f operator*(const Matrix2f &A, const Vector2f &x) { return Vector2f(A.m[0] * x[0] + A.m[1] * x[1], A.m[2] * x[0] + A.m[3] * x[1]); }
}  // namespace Eigen
using Eigen::Matrix2f;
using Eigen::Matrix3f;
using Eigen::Vector2f;
using Eigen::Vector3f;

namespace Eigen {
struct Quaternionf {   // Eigen::Quaternionf(w, x, y, z) with coefficient accessors: what the product's class shells use to cross their C ABI
    float c[4];       // x, y, z, w (Eigen's storage order)
    Quaternionf() { c[0] = c[1] = c[2] = 0; c[3] = 1; }
    Quaternionf(float w, float x, float y, float z) { c[0] = x; c[1] = y; c[2] = z; c[3] = w; }
    float x() const { return c[0]; }
    float y() const { return c[1]; }
    float z() const { return c[2]; }
    float w() const { return c[3]; }
};
}  // namespace Eigen

namespace Sophus {
// The projection searches only read rotationMatrix() / translation() (R, t are what the harness puts there); the direct projection
// composes, inverts and applies poses: that goes through the oracle's quaternion-form SE3f (Sophus semantics, ygz_oracle.h).
struct SE3f {
    Matrix3f R;
    Vector3f t;
    ygzo::SE3f q;
    SE3f() { R(0, 0) = R(1, 1) = R(2, 2) = 1.f; }
    explicit SE3f(const ygzo::SE3f &q_) : q(q_) { q.RotationMatrix(R.m); t = Vector3f(q.t[0], q.t[1], q.t[2]); }
    SE3f(const Eigen::Quaternionf &qq, const Vector3f &tt) {   // Sophus::SE3f(unit quaternion, translation)
        for (int i = 0; i < 4; i++) q.q[i] = qq.c[i];
        for (int i = 0; i < 3; i++) q.t[i] = tt[i];
        q.RotationMatrix(R.m);
        t = tt;
    }
    Eigen::Quaternionf unit_quaternion() const { return Eigen::Quaternionf(q.q[3], q.q[0], q.q[1], q.q[2]); }
    Matrix3f rotationMatrix() const { return R; }
    Vector3f translation() const { return t; }
    SE3f inverse() const { return SE3f(q.Inverse()); }
#ifdef YGZ_REF_TRACKING
    SE3f(const Matrix3f &, const Vector3f &) { yr_unsupported("SE3f(R, t)"); }                 // only the monocular initialiser builds poses from matrices
    template <class T> struct CastResult;
    template <class T> typename CastResult<T>::type cast() const { yr_unsupported("SE3f::cast"); }   // IMU branches only
#endif
    template <class V> static SE3f exp(const V &a) { float v[6]; for (int i = 0; i < 6; i++) v[i] = a[i]; return SE3f(ygzo::SE3f::Exp(v)); }
};
inline SE3f operator*(const SE3f &a, const SE3f &b) { return SE3f(a.q.Mul(b.q)); }
inline Vector3f operator*(const SE3f &T, const Vector3f &x) { Vector3f o; T.q.Act(x.v, o.v); return o; }
}  // namespace Sophus
using Sophus::SE3f;

namespace ygz {

class Frame;
class KeyFrame;
#ifdef YGZ_REF_TRACKING
class Map;
class KeyFrameDatabase;
struct IMUData;
class NavState;
#endif

#ifndef YGZ_REF_MAPPOINT
// include/MapPoint.h, src/MapPoint.cc: the fields / accessors the matcher touches
#define YGZ_STUB_MAPPOINT 1   // (no mutexes, GetDescriptor() hands out a header: code that reaches into the real class's privates asks for this)
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
    std::map<KeyFrame *, size_t> mObservations;
    std::map<KeyFrame *, size_t> GetObservations() { return mObservations; }
    float GetMinDistanceInvariance() { return minDistInv; }               // src/MapPoint.cc:347-350
    float GetMaxDistanceInvariance() { return maxDistInv; }               // :352-355
    int PredictScale(const float &currentDist, KeyFrame *pKF);           // :359-373 (body in ref_orbmatcher_capi.cpp)
    int PredictScale(const float &currentDist, Frame *pF);
    bool IsInKeyFrame(KeyFrame *) { yr_unsupported("MapPoint::IsInKeyFrame"); }
    int GetIndexInKeyFrame(KeyFrame *) { yr_unsupported("MapPoint::GetIndexInKeyFrame"); }
    void Replace(MapPoint *) { yr_unsupported("MapPoint::Replace"); }
    void AddObservation(KeyFrame *, size_t) { yr_unsupported("MapPoint::AddObservation"); }
#ifdef YGZ_REF_TRACKING
    // what src/Tracking.cc touches beyond the matcher: plain counters for the two members SearchLocalPoints calls, declarations for the rest
    // (bodies: tests/cpp/tracking_outside.S -- every member of a class outside the hot path aborts when reached)
    MapPoint() {}
    MapPoint(const Vector3f &Pos, KeyFrame *pRefKF, Map *pMap);
    MapPoint(const Vector3f &Pos, Map *pMap, Frame *pFrame, const int &idxF);
    long unsigned int mnLastFrameSeen = 0, mnTrackReferenceForFrame = 0;
    int mnVisible = 1, mnFound = 1;
    void IncreaseVisible(int n = 1) { mnVisible += n; }      // src/MapPoint.cc:186-189 without the mutex
    void IncreaseFound(int n = 1) { mnFound += n; }
    void ComputeDistinctiveDescriptors();
    void UpdateNormalAndDepth();
    MapPoint *GetReplaced();
    void SetWorldPos(const Vector3f &Pos);
#endif
};

#else
class MapPoint;
#endif

#ifndef YGZ_REF_FRAME
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
    long unsigned int mnId = 0;
    Vector3f mOw;
    Vector3f GetCameraCenter() { return mOw; }

    std::vector<size_t> GetFeaturesInArea(const float &x, const float &y, const float &r, const int minLevel = -1, const int maxLevel = -1) const;
    Vector2f Camera2Pixel(const Vector3f &p_c) const { return Vector2f(fx * p_c(0) / p_c(2) + cx, fy * p_c(1) / p_c(2) + cy); }   // include/Frame.h:154-159
    Vector2f World2Pixel(const Vector3f &p_w, const SE3f &T_c_w) const {   // include/Frame.h:146-175: Camera2Pixel(T_c_w * p_w)
        const Vector3f p_c = T_c_w * p_w;
        return Vector2f(fx * p_c(0) / p_c(2) + cx, fy * p_c(1) / p_c(2) + cy);
    }
};

#endif

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
    bool mHasPose = false;                     // set by the SearchForTriangulation entry point; Fuse / Sim3 stay outside the pinned path
    Matrix3f mRcw;
    Vector3f mtcw;
    Matrix3f GetRotation() { if (!mHasPose) yr_unsupported("KeyFrame::GetRotation"); return mRcw; }
    Vector3f GetTranslation() { if (!mHasPose) yr_unsupported("KeyFrame::GetTranslation"); return mtcw; }
    Vector3f mOw;
    Vector3f GetCameraCenter() { return mOw; }
    long unsigned int mnFrameId = 0;
    bool mbBad = false;
    bool isBad() { return mbBad; }
    void EraseMapPointMatch(MapPoint *) {}
    void EraseMapPointMatch(const size_t &) {}
    void ReplaceMapPointMatch(const size_t &, MapPoint *) {}
    SE3f mPose;
    SE3f GetPose() const { return mPose; }
    bool IsInImage(const float &, const float &) const { yr_unsupported("KeyFrame::IsInImage"); }
    std::vector<size_t> GetFeaturesInArea(const float &, const float &, const float &) const { yr_unsupported("KeyFrame::GetFeaturesInArea"); }
    Vector3f Pixel2Camera(const Vector2f &p_p, float depth = 1) const {   // include/KeyFrame.h:181-187
        return Vector3f((p_p(0) - cx) * depth / fx, (p_p(1) - cy) * depth / fy, depth);
    }
#ifdef YGZ_REF_TRACKING
    KeyFrame() {}
    KeyFrame(Frame &F, Map *pMap, KeyFrameDatabase *pKFDB);
    KeyFrame(Frame &F, Map *pMap, KeyFrameDatabase *pKFDB, std::vector<IMUData> vIMUData, KeyFrame *pLastKF = NULL);
    static long unsigned int nNextId;
    double mTimeStamp = 0;
    long unsigned int mnTrackReferenceForFrame = 0;
    void ComputeBoW();
    void ComputePreInt(void);
    float ComputeSceneMedianDepth(const int q);
    std::vector<KeyFrame *> GetBestCovisibilityKeyFrames(const int &N);
    std::set<KeyFrame *> GetChilds();
    const NavState &GetNavState(void);
    KeyFrame *GetParent();
    KeyFrame *GetPrevKeyFrame(void);
    SE3f GetPoseInverse();
    void SetInitialNavStateAndBias(const NavState &ns);
    void SetPose(const SE3f &Tcw) { mPose = Tcw; }
    int TrackedMapPoints(const int &minObs);
    void UpdateConnections();
    DBoW2::BowVector mBowVec;
#endif
};

// include/Map.h: what src/MapPoint.cc touches
class Map {
public:
    std::mutex mMutexPointCreation;
    void EraseMapPoint(MapPoint *) {}
#ifdef YGZ_REF_TRACKING
    std::mutex mMutexMapUpdate;
    std::vector<KeyFrame *> mvpKeyFrameOrigins;
    void AddKeyFrame(KeyFrame *pKF);
    void AddMapPoint(MapPoint *pMP);
    std::vector<KeyFrame *> GetAllKeyFrames();
    std::vector<MapPoint *> GetAllMapPoints();
    long unsigned KeyFramesInMap();
    long unsigned int MapPointsInMap();
    void SetReferenceMapPoints(const std::vector<MapPoint *> &vpMPs);
    void clear();
#endif
};

// include/Converter.h: only the Sim3 / fuse functions use it
class Converter {
public:
    static cv::Mat toCvMat(const Vector3f &) { yr_unsupported("Converter::toCvMat"); }
    static cv::Mat toCvMat(const Matrix3f &) { yr_unsupported("Converter::toCvMat"); }
    template <class A, class B, class C> static void updateNS(A &, const B &, const C &) { yr_unsupported("Converter::updateNS"); }
#ifdef YGZ_BOUNDARY_BUILD
    struct SE3QuatStandIn {   // g2o::SE3Quat: only Relocalization's PnP branch names it
        Matrix3f rotation() const { yr_unsupported("SE3Quat::rotation"); }
        Vector3f translation() const { yr_unsupported("SE3Quat::translation"); }
    };
    template <class T> static SE3QuatStandIn toSE3Quat(const T &) { yr_unsupported("Converter::toSE3Quat"); }
    static std::vector<cv::Mat> toDescriptorVector(const cv::Mat &D) {   // src/Converter.cc:52-59
        std::vector<cv::Mat> v;
        for (int j = 0; j < D.rows; j++) v.push_back(D.row(j));
        return v;
    }
#else
    static std::vector<cv::Mat> toDescriptorVector(const cv::Mat &) { yr_unsupported("Converter::toDescriptorVector"); }
#endif
};

// include/Align.h
bool Align2D(const cv::Mat &cur_img, uint8_t *ref_patch_with_border, uint8_t *ref_patch, const int n_iter, Vector2f &cur_px_estimate,
             bool no_simd = false);   // body: the reference's own src/Align.cc

}  // namespace ygz
#endif
