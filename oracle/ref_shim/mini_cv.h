// oracle/ref_shim/mini_cv.h -- TEST INFRASTRUCTURE.  The smallest stand-in for the OpenCV API surface that the reference's
// src/ORBextractor.cc touches, so that THE REFERENCE'S OWN SOURCE FILE can be compiled where it lies (/root/reference/src/ORBextractor.cc,
// never copied) and run next to the oracle: oracle/Makefile target `ref_extractor` -> oracle/_ref/libref_orbextractor.so.
//
// What this pins and what it does not: every line of reference-owned logic in that file -- pyramid flow, the 30-px cell loop with its
// threshold fallback, ExtractorNode::DivideNode / DistributeOctTree, IC_Angle, computeOrbDescriptor with the reference's own
// bit_pattern_31_ table, the scale bookkeeping of operator(), ComputeKeyPointsDSOSingleLevel + ShiTomasiScore on the reference's own
// libfast -- runs as written and must agree bit for bit with the oracle's restatement (tests/test_ref_extractor.py).  The OpenCV
// PRIMITIVES underneath (cv::resize, cv::FAST, cv::GaussianBlur, cv::fastAtan2, cvRound) are the oracle's restatements
// (oracle_cvprims.cpp) on both sides, so they stay unpinned (OpenCV is not available in this environment).
//
// Also neutralises the reference's Common.h (Eigen / Sophus / g2o / pangolin / glog are absent): its include guard is defined here and
// the few names ORBextractor.{h,cc} need from it are provided instead.
#ifndef YGZ_ORACLE_MINI_CV_H
#define YGZ_ORACLE_MINI_CV_H

#define YGZ_COMMON_H_   // include guard of the reference's include/Common.h: skip it

#include <algorithm>
#include <cassert>
#include <climits>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <iostream>
#include <list>
#include <map>
#include <memory>
#include <mutex>
#include <set>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

typedef unsigned char uchar;

#define CV_8U 0
#define CV_8UC1 0
#define CV_32F 5
#define CV_PI 3.1415926535897932384626433832795
#define CV_Assert(x) assert(x)

int cvRound(double v);   // round half to even (the oracle's cv_round)
inline int cvFloor(double v) { int i = (int) v; return i - (i > v); }
inline int cvCeil(double v) { int i = (int) v; return i + (i < v); }

// glog stand-in: LOG(INFO) << ... goes nowhere
struct MiniCvNullStream {
    template <class T> MiniCvNullStream &operator<<(const T &) { return *this; }
    MiniCvNullStream &operator<<(std::ostream &(*)(std::ostream &)) { return *this; }
};
#define LOG(severity) MiniCvNullStream()

namespace cv {

template <class T> struct Point_ {
    T x, y;
    Point_() : x(0), y(0) {}
    Point_(T x_, T y_) : x(x_), y(y_) {}
    template <class U> Point_(const Point_<U> &o) : x((T) o.x), y((T) o.y) {}
    Point_ &operator*=(float s) { x = (T) (x * s); y = (T) (y * s); return *this; }   // saturate_cast<float> of a float product
};
template <class T> inline Point_<T> operator*(const Point_<T> &a, float s) { return Point_<T>((T) (a.x * s), (T) (a.y * s)); }
typedef Point_<int> Point2i;
typedef Point2i Point;
typedef Point_<float> Point2f;

struct Size {
    int width, height;
    Size() : width(0), height(0) {}
    Size(int w, int h) : width(w), height(h) {}
};
struct Rect {
    int x, y, width, height;
    Rect() : x(0), y(0), width(0), height(0) {}
    Rect(int x_, int y_, int w, int h) : x(x_), y(y_), width(w), height(h) {}
};

struct KeyPoint {
    Point2f pt;
    float size, angle, response;
    int octave, class_id;
    KeyPoint() : size(0), angle(-1), response(0), octave(0), class_id(-1) {}
    KeyPoint(float x, float y, float s, float a = -1, float r = 0, int o = 0, int c = -1) : pt(x, y), size(s), angle(a), response(r), octave(o), class_id(c) {}
    KeyPoint(Point2f p, float s, float a = -1, float r = 0, int o = 0, int c = -1) : pt(p), size(s), angle(a), response(r), octave(o), class_id(c) {}
};

struct MatStep {   // Mat::step: converts to size_t and offers .p[0]
    size_t p[2];
    MatStep() { p[0] = p[1] = 0; }
    operator size_t() const { return p[0]; }
};

// Mat::zeros returns a matrix EXPRESSION in OpenCV: assigning it to a matrix that already has that size and type fills the existing
// buffer in place (the reference's computeDescriptors relies on it: `descriptors = Mat::zeros(...)` on a rowRange view of the output
// matrix, src/ORBextractor.cc:964, must clear the view, not rebind the header)
struct MatZerosExpr { int rows, cols; };

[[noreturn]] inline void mini_cv_unsupported(const char *what) {
    std::fprintf(stderr, "oracle/ref_shim/mini_cv: %s is not provided (outside the pinned path)\n", what);
    std::abort();
}

#define YGZ_MINI_CV 1   // (lets code that must look inside cv::Mat -- the reference count -- tell this stand-in from a real OpenCV)
// 8-bit single-channel matrix header over a shared buffer (views: rowRange / colRange / operator()(Rect) keep the parent's step)
class Mat {
public:
    int rows, cols;
    uchar *data;
    MatStep step;
    Mat() : rows(0), cols(0), data(nullptr) {}
    Mat(int r, int c, int type) : rows(0), cols(0), data(nullptr) { create(r, c, type); }
    Mat(Size s, int type) : rows(0), cols(0), data(nullptr) { create(s.height, s.width, type); }
    template <class S> Mat(Size s, int type, const S &) : rows(0), cols(0), data(nullptr) { create(s.height, s.width, type); }   // (size, type, Scalar(0)): residual display image
    void create(int r, int c, int type) {
        const size_t es = type == CV_32F ? 4 : 1;     // 8-bit everywhere except the aligner's float patch cache
        if (data && rows == r && cols == c && elem == es) return;   // as cv::Mat::create: same size and type -> nothing happens
        buf = std::make_shared<std::vector<uchar>>((size_t) r * c * es + 64);
        rows = r; cols = c; elem = es; data = buf->data(); step.p[0] = (size_t) c * es; step.p[1] = es;
    }
    size_t elem = 1;
    Size size() const { return Size(cols, rows); }
    static MatZerosExpr zeros(int r, int c, int /*type*/) { return MatZerosExpr{r, c}; }
    Mat(const MatZerosExpr &e) : rows(0), cols(0), data(nullptr) { *this = e; }
    Mat &operator=(const MatZerosExpr &e) {
        create(e.rows, e.cols, CV_8UC1);
        for (int y = 0; y < rows; y++) std::memset(data + (size_t) y * step.p[0], 0, (size_t) cols);
        return *this;
    }
    void release() { buf.reset(); rows = cols = 0; data = nullptr; }
    int use_count() const { return buf ? (int) buf.use_count() : 0; }   // what cv::Mat::u->refcount (OpenCV >= 3) / *cv::Mat::refcount (2.4) tell
    bool empty() const { return data == nullptr || rows == 0 || cols == 0; }
    int type() const { return CV_8UC1; }
    bool isContinuous() const { return rows <= 1 || step.p[0] == (size_t) cols * elem; }
    size_t step1() const { return step.p[0]; }
    Mat clone() const {
        Mat m(rows, cols, elem == 4 ? CV_32F : CV_8UC1);
        for (int y = 0; y < rows; y++) std::memcpy(m.data + (size_t) y * m.step.p[0], data + (size_t) y * step.p[0], (size_t) cols * elem);
        return m;
    }
    Mat rowRange(int a, int b) const { Mat m = *this; m.data = data + (size_t) a * step.p[0]; m.rows = b - a; return m; }
    Mat colRange(int a, int b) const { Mat m = *this; m.data = data + a; m.cols = b - a; return m; }
    Mat operator()(const Rect &r) const { return rowRange(r.y, r.y + r.height).colRange(r.x, r.x + r.width); }
    Mat row(int y) const { return rowRange(y, y + 1); }
    void copyTo(Mat &dst) const { dst = clone(); }
    // 8-bit -> float conversion (Frame::ComputeStereoMatches converts its 11 x 11 windows); a new buffer, as OpenCV allocates for a new type
    void convertTo(Mat &dst, int type) const {
        assert(type == CV_32F && elem == 1);
        Mat o(rows, cols, CV_32F);
        for (int y = 0; y < rows; y++) for (int x = 0; x < cols; x++) o.at<float>(y, x) = (float) at<uchar>(y, x);
        dst = o;
    }
    static Mat ones(int r, int c, int type) { assert(type == CV_32F); Mat o(r, c, CV_32F); for (int y = 0; y < r; y++) for (int x = 0; x < c; x++) o.at<float>(y, x) = 1.f; return o; }
    // float-matrix algebra (Sim3 / fuse functions of the matcher): declared so that those functions compile, never executed
    Mat col(int) const { mini_cv_unsupported("Mat::col"); }
    Mat t() const { mini_cv_unsupported("Mat::t"); }
    double dot(const Mat &) const { mini_cv_unsupported("Mat::dot"); }
    template <class T> T &at(int i) { assert(cols == 1); return *(T *) (data + (size_t) i * step.p[0]); }          // column vectors (Tracking's DistCoef)
    template <class T> const T &at(int i) const { assert(cols == 1); return *(const T *) (data + (size_t) i * step.p[0]); }
    int channels() const { return 1; }
    void resize(size_t nrows) {   // column vector grown in place (Tracking.cc:122)
        Mat o((int) nrows, cols, elem == 4 ? CV_32F : CV_8UC1);
        for (int y = 0; y < (int) nrows; y++) std::memset(o.data + (size_t) y * o.step.p[0], 0, (size_t) cols * elem);
        for (int y = 0; y < rows && y < (int) nrows; y++) std::memcpy(o.data + (size_t) y * o.step.p[0], data + (size_t) y * step.p[0], (size_t) cols * elem);
        *this = o;
    }
    void convertTo(Mat &, int, double) const { mini_cv_unsupported("Mat::convertTo(dst, type, alpha)"); }
    template <class T> T &at(int y, int x) { return *(T *) (data + (size_t) y * step.p[0] + (size_t) x * sizeof(T)); }
    template <class T> const T &at(int y, int x) const { return *(const T *) (data + (size_t) y * step.p[0] + (size_t) x * sizeof(T)); }
    template <class T> T *ptr(int y = 0) { return (T *) (data + (size_t) y * step.p[0]); }
    template <class T> const T *ptr(int y = 0) const { return (const T *) (data + (size_t) y * step.p[0]); }
    uchar *ptr(int y = 0) { return data + (size_t) y * step.p[0]; }
    const uchar *ptr(int y = 0) const { return data + (size_t) y * step.p[0]; }

private:
    std::shared_ptr<std::vector<uchar>> buf;
};

class _InputArray {
public:
    _InputArray(const Mat &m) : m_(const_cast<Mat *>(&m)) {}
    bool empty() const { return m_->empty(); }
    Mat getMat() const { return *m_; }
protected:
    Mat *m_;
};
class _OutputArray : public _InputArray {
public:
    _OutputArray(Mat &m) : _InputArray(m) {}
    void release() const { m_->release(); }
    void create(int r, int c, int type) const { m_->create(r, c, type); }
};
typedef const _InputArray &InputArray;
typedef const _OutputArray &OutputArray;

inline Mat operator/(const Mat &, double) { mini_cv_unsupported("Mat / s"); }
inline Mat operator*(const Mat &, const Mat &) { mini_cv_unsupported("Mat * Mat"); }
inline Mat operator*(double s, const Mat &a) {   // float matrices only (the stereo windows)
    assert(a.elem == 4);
    Mat o(a.rows, a.cols, CV_32F);
    for (int y = 0; y < a.rows; y++) for (int x = 0; x < a.cols; x++) o.at<float>(y, x) = (float) (s * a.at<float>(y, x));
    return o;
}
inline Mat operator+(const Mat &, const Mat &) { mini_cv_unsupported("Mat + Mat"); }
inline Mat operator-(const Mat &a, const Mat &b) {
    assert(a.elem == 4 && b.elem == 4 && a.rows == b.rows && a.cols == b.cols);
    Mat o(a.rows, a.cols, CV_32F);
    for (int y = 0; y < a.rows; y++) for (int x = 0; x < a.cols; x++) o.at<float>(y, x) = a.at<float>(y, x) - b.at<float>(y, x);
    return o;
}
inline Mat operator-(const Mat &) { mini_cv_unsupported("-Mat"); }
inline double norm(const Mat &) { mini_cv_unsupported("cv::norm"); }
enum { NORM_L1 = 2 };
inline double norm(const Mat &a, const Mat &b, int type) {   // NORM_L1 of float matrices, accumulated in double as OpenCV does
    assert(type == NORM_L1 && a.elem == 4 && b.elem == 4);
    double acc = 0;
    for (int y = 0; y < a.rows; y++) for (int x = 0; x < a.cols; x++) acc += std::fabs((double) a.at<float>(y, x) - (double) b.at<float>(y, x));
    return acc;
}
#define CV_16SC2 11
inline void initUndistortRectifyMap(const Mat &, const Mat &, const Mat &, const Mat &, Size, int, Mat &, Mat &) { mini_cv_unsupported("initUndistortRectifyMap"); }
inline void remap(const Mat &, Mat &, const Mat &, const Mat &, int) { mini_cv_unsupported("remap"); }
inline void undistortPoints(const Mat &, Mat &, const Mat &, const Mat &, const Mat &, const Mat &) { mini_cv_unsupported("undistortPoints"); }

enum { BORDER_REFLECT_101 = 4, BORDER_ISOLATED = 16, INTER_LINEAR = 1 };

// the primitives: bodies in mini_cv.cpp on top of the oracle's restatements (oracle_cvprims.cpp)
void resize(InputArray src, OutputArray dst, Size dsize, double fx = 0, double fy = 0, int interpolation = INTER_LINEAR);
void GaussianBlur(InputArray src, OutputArray dst, Size ksize, double sigmaX, double sigmaY = 0, int borderType = BORDER_REFLECT_101);
void copyMakeBorder(InputArray src, OutputArray dst, int top, int bottom, int left, int right, int borderType);
void FAST(InputArray image, std::vector<KeyPoint> &keypoints, int threshold, bool nonmaxSuppression = true);
float fastAtan2(float y, float x);

struct KeyPointsFilter {
    static void retainBest(std::vector<KeyPoint> &keypoints, int npoints);   // only the dead ComputeKeyPointsOld uses it
};

}  // namespace cv

#ifdef YGZ_REF_MATCHER
using cv::Mat;   // include/Common.h:64
#include "Thirdparty/DBoW2/DBoW2/FeatureVector.h"
#include "matcher_stubs.h"
#include "sia_stubs.h"
#ifdef YGZ_REF_FRAME
#include "Thirdparty/DBoW2/DBoW2/BowVector.h"
#include "frame_stubs.h"
#endif
#endif
#endif
