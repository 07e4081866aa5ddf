// ygz_compat.h -- the types the three class shells are written against (product code, host side).
//
// Built inside the reference tree (-DYGZF_WITH_REFERENCE_HEADERS) this header simply pulls the
// reference's own Common.h / Frame.h / MapPoint.h / KeyFrame.h (OpenCV cv::Mat / cv::KeyPoint, Sophus SE3f, Eigen) and defines the
// few adapters below on top of them; it is included by the shells' .cc files only (Frame.h itself includes ORBextractor.h).
// Built stand-alone (this repository: no OpenCV / Eigen / Sophus installed) it provides a
// minimal stand-in for exactly the slice of those types the hot-path signatures touch, layout-compatible where layout
// matters (cv::KeyPoint = 7 x 4 bytes, continuous 8-bit Mats), so that the shells compile and are testable here.
#ifndef YGZF_COMPAT_H
#define YGZF_COMPAT_H

#ifdef YGZF_WITH_REFERENCE_HEADERS
#include "Common.h"     // reference include/Common.h: OpenCV, Eigen, Sophus, glog, typedefs (SE3f, Vector3f, ...)
#include "Frame.h"
#include "MapPoint.h"
#include "KeyFrame.h"
namespace ygz_compat {   // SE3f / Vector3f / Matrix3f are the global names include/Common.h brings in (using Sophus::SE3f, Eigen::...)
inline void se3_to7(const SE3f &T, float o[7]) {
    const Eigen::Quaternionf q = T.unit_quaternion();
    o[0] = q.x(); o[1] = q.y(); o[2] = q.z(); o[3] = q.w();
    o[4] = T.translation()[0]; o[5] = T.translation()[1]; o[6] = T.translation()[2];
}
inline SE3f se3_from7(const float i[7]) {
    return SE3f(Eigen::Quaternionf(i[3], i[0], i[1], i[2]), Vector3f(i[4], i[5], i[6]));
}
inline void se3_to_Rt(const SE3f &T, float R[9], float t[3]) {
    const Matrix3f M = T.rotationMatrix();
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) R[3 * r + c] = M(r, c);
    for (int r = 0; r < 3; r++) t[r] = T.translation()[r];
}
inline void world_pos(ygz::MapPoint *mp, float o[3]) { const Vector3f p = mp->GetWorldPos(); o[0] = p[0]; o[1] = p[1]; o[2] = p[2]; }
// how many cv::Mat headers share m's buffer (0: no buffer of its own / external data): the level pool of ORBextractor::ComputePyramid hands a
// buffer out again only when nobody but the pool refers to it
inline int mat_refcount(const cv::Mat &m) {
#if defined(YGZ_MINI_CV)
    return m.use_count();
#elif defined(CV_MAJOR_VERSION) && CV_MAJOR_VERSION >= 3
    return m.u ? (int) m.u->refcount : 0;
#else
    return m.refcount ? *m.refcount : 0;
#endif
}
}  // namespace ygz_compat
#else  // ---------------------------------------------------------------------------------------------- stand-alone shim
#include <cstdint>
#include <cmath>
#include <cstring>
#include <map>
#include <memory>
#include <vector>

#define CV_8U 0
#define CV_8UC1 0
#define CV_32F 5

#include <set>
namespace cv {
struct Point2f { float x, y; };
struct KeyPoint {  // bit-compatible with cv::KeyPoint
    Point2f pt;
    float size, angle, response;
    int octave, class_id;
};
class Mat {
public:
    int rows = 0, cols = 0;
    size_t step = 0;
    unsigned char *data = nullptr;
    Mat() {}
    Mat(int r, int c, int type) { create(r, c, type); }
    Mat(int r, int c, int type, void *ext, size_t st = 0) : rows(r), cols(c), step(st ? st : (size_t) c * esz(type)), data((unsigned char *) ext), type_(type) {}
    void create(int r, int c, int type) {
        if (r == rows && c == cols && type == type_ && data && hold_ && step == (size_t) c * esz(type)) return;
        rows = r; cols = c; type_ = type; step = (size_t) c * esz(type);
        hold_.reset(new unsigned char[(size_t) r * step + 16], std::default_delete<unsigned char[]>());
        data = hold_.get();
    }
    void release() { hold_.reset(); data = nullptr; rows = cols = 0; step = 0; }
    bool empty() const { return data == nullptr || rows == 0 || cols == 0; }
    bool isContinuous() const { return step == (size_t) cols * esz(type_); }
    int type() const { return type_; }
    template <typename T> T *ptr(int r = 0) { return (T *) (data + (size_t) r * step); }
    template <typename T> const T *ptr(int r = 0) const { return (const T *) (data + (size_t) r * step); }
    unsigned char *ptr(int r = 0) { return data + (size_t) r * step; }
    const unsigned char *ptr(int r = 0) const { return data + (size_t) r * step; }
    Mat row(int r) const { Mat m(1, cols, type_, (void *) (data + (size_t) r * step), step); m.hold_ = hold_; return m; }
    int refcount() const { return hold_ ? (int) hold_.use_count() : 0; }   // (stand-in for cv::Mat::u->refcount)
    Mat clone() const {
        Mat m(rows, cols, type_);
        for (int r = 0; r < rows; r++) std::memcpy(m.ptr(r), ptr(r), (size_t) cols * esz(type_));
        return m;
    }
private:
    static size_t esz(int type) { return type == CV_32F ? 4 : 1; }
    int type_ = CV_8U;
    std::shared_ptr<unsigned char> hold_;
};
class _InputArray {
public:
    _InputArray(const Mat &m) : m_(&m) {}
    Mat getMat() const { return *m_; }
    bool empty() const { return m_->empty(); }
private:
    const Mat *m_;
};
class _OutputArray {
public:
    _OutputArray(Mat &m) : m_(&m) {}
    void create(int r, int c, int type) const { m_->create(r, c, type); }
    void release() const { m_->release(); }
    Mat getMat() const { return *m_; }
private:
    Mat *m_;
};
typedef const _InputArray &InputArray;
typedef const _OutputArray &OutputArray;
}  // namespace cv

namespace DBoW2 {
typedef std::map<unsigned int, std::vector<unsigned int>> FeatureVector;   // node id -> feature indices (DBoW2/FeatureVector.h)
}

namespace ygz {
struct Vector3f { float v[3]; float &operator[](int i) { return v[i]; } const float &operator[](int i) const { return v[i]; } };
struct Matrix3f { float m[9]; float &operator()(int r, int c) { return m[3 * r + c]; } const float &operator()(int r, int c) const { return m[3 * r + c]; } };
struct Vector2f { float v[2]; float &operator[](int i) { return v[i]; } const float &operator[](int i) const { return v[i]; } };
class KeyFrame;
struct SE3f {  // Sophus::SE3f storage: unit quaternion (x,y,z,w) + translation
    float q[4] = {0, 0, 0, 1};
    float t[3] = {0, 0, 0};
};
class Frame;
class ORBextractor;
class MapPoint {  // the accessors the hot path calls (reference include/MapPoint.h)
public:
    Vector3f mWorldPos{};
    cv::Mat mDescriptor;
    int nObs = 1;
    bool mbBad = false;
    // set by Frame::isInFrustum (src/Frame.cc:413-419)
    bool mbTrackInView = false;
    float mTrackProjX = 0, mTrackProjY = 0, mTrackProjXR = 0, mTrackViewCos = 0;
    int mnTrackScaleLevel = 0;
    Vector3f GetWorldPos() const { return mWorldPos; }
    cv::Mat GetDescriptor() const { return mDescriptor.clone(); }
    int Observations() const { return nObs; }
    bool isBad() const { return mbBad; }
    // scale-invariance range (src/MapPoint.cc:325-373)
    float mfMinDistance = 0, mfMaxDistance = 0;
    float GetMinDistanceInvariance() const { return 0.8f * mfMinDistance; }
    float GetMaxDistanceInvariance() const { return 1.2f * mfMaxDistance; }
    inline int PredictScale(const float &currentDist, Frame *pF);
    std::map<KeyFrame *, size_t> mObservations;
    std::map<KeyFrame *, size_t> GetObservations() const { return mObservations; }
};
class Frame {  // the members the hot path reads/writes (reference include/Frame.h)
public:
    static float fx, fy, cx, cy, invfx, invfy, mnMinX, mnMaxX, mnMinY, mnMaxY;
    float mb = 0, mbf = 0;
    int N = 0;
    cv::Mat mImGray, mImRight;
    std::vector<cv::Mat> mvImagePyramid;
    std::vector<cv::KeyPoint> mvKeys, mvKeysRight;
    std::vector<float> mvuRight, mvDepth;
    cv::Mat mDescriptors, mDescriptorsRight;
    std::vector<MapPoint *> mvpMapPoints;
    std::vector<bool> mvbOutlier;
    std::vector<float> mvScaleFactors, mvInvScaleFactors;
    SE3f mTcw;
    float mfLogScaleFactor = 0;
    int mnScaleLevels = 0;
    DBoW2::FeatureVector mFeatVec;
    long unsigned int mnId = 0;
    ORBextractor *mpORBextractorLeft = nullptr;   // the extractor whose ComputePyramid built mvImagePyramid (include/Frame.h:220)
};
inline int MapPoint::PredictScale(const float &currentDist, Frame *pF) {  // src/MapPoint.cc:359-373
    float ratio = mfMaxDistance / currentDist;
    int nScale = (int) std::ceil(std::log(ratio) / pF->mfLogScaleFactor);
    if (nScale < 0) nScale = 0;
    else if (nScale >= pF->mnScaleLevels) nScale = pF->mnScaleLevels - 1;
    return nScale;
}
class KeyFrame {  // the members SearchByProjection(Cur, KF, ...) reads (reference include/KeyFrame.h)
public:
    std::vector<cv::KeyPoint> mvKeys;
    std::vector<MapPoint *> mvpMapPoints;
    std::vector<MapPoint *> GetMapPointMatches() const { return mvpMapPoints; }
    cv::Mat mDescriptors;
    DBoW2::FeatureVector mFeatVec;
    // read by FindDirectProjection (src/ORBmatcher.cc:1574-1602)
    long unsigned int mnId = 0;
    std::vector<cv::Mat> mvImagePyramid;
    SE3f mPose;
    SE3f GetPose() const { return mPose; }
    // read by SearchForTriangulation (src/ORBmatcher.cc:596-741)
    int N = 0;
    std::vector<float> mvuRight, mvScaleFactors, mvLevelSigma2;
    float fx = 0, fy = 0, cx = 0, cy = 0;
    MapPoint *GetMapPoint(const size_t &i) const { return mvpMapPoints[i]; }
    Matrix3f mRcw{};
    Vector3f mtcw{}, mOw{};
    Matrix3f GetRotation() const { return mRcw; }
    Vector3f GetTranslation() const { return mtcw; }
    Vector3f GetCameraCenter() const { return mOw; }
};
}  // namespace ygz

namespace ygz_compat {
inline int mat_refcount(const cv::Mat &m) { return m.refcount(); }
inline void se3_to7(const ygz::SE3f &T, float o[7]) { std::memcpy(o, T.q, 16); std::memcpy(o + 4, T.t, 12); }
inline ygz::SE3f se3_from7(const float i[7]) { ygz::SE3f T; std::memcpy(T.q, i, 16); std::memcpy(T.t, i + 4, 12); return T; }
inline void se3_to_Rt(const ygz::SE3f &T, float R[9], float t[3]) {  // Eigen Quaternion::toRotationMatrix
    const float *q = T.q;
    const float tx = 2 * q[0], ty = 2 * q[1], tz = 2 * q[2];
    const float twx = tx * q[3], twy = ty * q[3], twz = tz * q[3];
    const float txx = tx * q[0], txy = ty * q[0], txz = tz * q[0];
    const float tyy = ty * q[1], tyz = tz * q[1], tzz = tz * q[2];
    R[0] = 1 - (tyy + tzz); R[1] = txy - twz; R[2] = txz + twy;
    R[3] = txy + twz; R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
    R[6] = txz - twy; R[7] = tyz + twx; R[8] = 1 - (txx + tyy);
    t[0] = T.t[0]; t[1] = T.t[1]; t[2] = T.t[2];
}
inline void world_pos(ygz::MapPoint *mp, float o[3]) { const ygz::Vector3f p = mp->GetWorldPos(); o[0] = p[0]; o[1] = p[1]; o[2] = p[2]; }
}  // namespace ygz_compat
#endif
#endif
