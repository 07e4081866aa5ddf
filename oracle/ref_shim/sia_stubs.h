// oracle/ref_shim/sia_stubs.h -- TEST INFRASTRUCTURE (included by mini_cv.h after matcher_stubs.h when YGZ_REF_MATCHER is defined).
// A fixed-size matrix template with just the Eigen expressions that include/NLSSolver.h, include/NLSSolver_impl.hpp,
// include/SparseImageAlign.h and src/SparseImageAlign.cc use, so that the reference's own sparse image aligner compiles where it lies.
// Element-wise operation order is the natural one (what the oracle's restatement uses): (J * J^T) * w, (J * res) * w,
// (dx * row0 + dy * row1) * s; H.ldlt().solve(b) is the oracle's pivoted LDL^T (oracle_align.cpp), SE3f::exp the oracle's (Sophus) exp.
#ifndef YGZ_ORACLE_REF_SHIM_SIA_STUBS_H
#define YGZ_ORACLE_REF_SHIM_SIA_STUBS_H

#define EIGEN_MAKE_ALIGNED_OPERATOR_NEW

namespace ygzo { bool ldlt_solve6(const float H[36], const float b[6], float x[6]); }

namespace Eigen {
enum { Dynamic = -1, ColMajor = 0, RowMajor = 1 };
struct NoChange_t {};
static const NoChange_t NoChange = NoChange_t();

template <class T, int R, int C, int Opt = 0> struct Matrix {   // column-major, R fixed, C fixed or Dynamic
    std::vector<T> d;
    int cols_;
    Matrix() : d((size_t) R * (C == Dynamic ? 0 : C), T(0)), cols_(C == Dynamic ? 0 : C) {}
    static Matrix Zero() { return Matrix(); }
    int rows() const { return R; }
    int cols() const { return cols_; }
    int size() const { return R * cols_; }
    void resize(NoChange_t, int n) { cols_ = n; d.assign((size_t) R * n, T(0)); }
    void setZero() { std::fill(d.begin(), d.end(), T(0)); }
    T &operator()(int r, int c) { return d[(size_t) c * R + r]; }
    T operator()(int r, int c) const { return d[(size_t) c * R + r]; }
    T &operator()(int i) { return d[i]; }
    T operator()(int i) const { return d[i]; }
    T &operator[](int i) { return d[i]; }
    T operator[](int i) const { return d[i]; }
    Matrix &noalias() { return *this; }
    Matrix &operator+=(const Matrix &o) { for (size_t i = 0; i < d.size(); i++) d[i] += o.d[i]; return *this; }
    Matrix &operator-=(const Matrix &o) { for (size_t i = 0; i < d.size(); i++) d[i] -= o.d[i]; return *this; }
    Matrix<T, C, R> transpose() const { Matrix<T, C, R> t; for (int r = 0; r < R; r++) for (int c = 0; c < cols_; c++) t(c, r) = (*this)(r, c); return t; }
    Matrix<T, 1, C> row(int r) const { Matrix<T, 1, C> o; for (int c = 0; c < cols_; c++) o(0, c) = (*this)(r, c); return o; }
    struct ColRef {   // jacobian_cache_.col(k) = row-vector expression;  Vector6f J(jacobian_cache_.col(k))
        Matrix *M; int c;
        void operator=(const Matrix<T, R, 1> &v) { for (int r = 0; r < R; r++) (*M)(r, c) = v[r]; }
        void operator=(const Matrix<T, 1, R> &v) { for (int r = 0; r < R; r++) (*M)(r, c) = v[r]; }   // Eigen transposes vectors on assignment
        operator Matrix<T, R, 1>() const { Matrix<T, R, 1> v; for (int r = 0; r < R; r++) v[r] = (*M)(r, c); return v; }
    };
    ColRef col(int c) { return ColRef{this, c}; }
    // Levenberg-Marquardt damping `H_ += (H_.diagonal() * mu_).asDiagonal()` (NLSSolver_impl.hpp:144): compiles, never taken (GaussNewton)
    struct Diag { Matrix<T, R, 1> v; Diag operator*(T s) const { Diag o = *this; for (auto &x : o.v.d) x = x * s; return o; } Matrix asDiagonal() const { Matrix m; for (int i = 0; i < R; i++) m(i, i) = v[i]; return m; } };
    Diag diagonal() const { Diag o; for (int i = 0; i < R; i++) o.v[i] = (*this)(i, i); return o; }
    struct LDLT { const Matrix *M; Matrix<T, R, 1> solve(const Matrix<T, R, 1> &b) const; };
    LDLT ldlt() const { return LDLT{this}; }
};
template <class T, int R, int C, int O> Matrix<T, R, C, O> operator*(T s, const Matrix<T, R, C, O> &a) { Matrix<T, R, C, O> r = a; for (auto &x : r.d) x = s * x; return r; }
template <class T, int R, int C, int O> Matrix<T, R, C, O> operator*(const Matrix<T, R, C, O> &a, T s) { Matrix<T, R, C, O> r = a; for (auto &x : r.d) x = x * s; return r; }
template <class T, int R, int C, int O> Matrix<T, R, C, O> operator/(const Matrix<T, R, C, O> &a, T s) { Matrix<T, R, C, O> r = a; for (auto &x : r.d) x = x / s; return r; }
template <class T, int R, int C, int O> Matrix<T, R, C, O> operator+(const Matrix<T, R, C, O> &a, const Matrix<T, R, C, O> &b) { Matrix<T, R, C, O> r = a; r += b; return r; }
template <class T, int R, int C, int O> Matrix<T, R, C, O> operator-(const Matrix<T, R, C, O> &a) { Matrix<T, R, C, O> r = a; for (auto &x : r.d) x = -x; return r; }
template <class T, int R, int K, int C> Matrix<T, R, C> operator*(const Matrix<T, R, K> &a, const Matrix<T, K, C> &b) {
    Matrix<T, R, C> r;
    for (int i = 0; i < R; i++)
        for (int j = 0; j < C; j++) {
            T s = a(i, 0) * b(0, j);
            for (int k = 1; k < K; k++) s = s + a(i, k) * b(k, j);
            r(i, j) = s;
        }
    return r;
}
template <class T, int R, int C, int O> std::ostream &operator<<(std::ostream &os, const Matrix<T, R, C, O> &) { return os << "[matrix]"; }
template <class T, int R, int C, int O> Matrix<T, R, 1> Matrix<T, R, C, O>::LDLT::solve(const Matrix<T, R, 1> &b) const {
    static_assert(R == 6, "only the aligner's 6 x 6 system");
    float H[36], bb[6], x[6];
    for (int r = 0; r < 6; r++) { bb[r] = b[r]; for (int c = 0; c < 6; c++) H[6 * r + c] = (*M)(r, c); }
    ygzo::ldlt_solve6(H, bb, x);
    Matrix<T, R, 1> o;
    for (int r = 0; r < 6; r++) o[r] = x[r];
    return o;
}

struct VectorXf {   // norm_max(const Eigen::VectorXf &) in include/NLSSolver.h
    std::vector<float> d;
    template <int R> VectorXf(const Matrix<float, R, 1> &m) : d(m.d) {}
    int size() const { return (int) d.size(); }
    float operator[](int i) const { return d[i]; }
};
}  // namespace Eigen

typedef Eigen::Matrix<float, 6, 1> Vector6f;   // include/Common.h

namespace Sophus {
inline SE3f se3_exp6(const Vector6f &a) { float v[6]; for (int i = 0; i < 6; i++) v[i] = a[i]; return SE3f(ygzo::SE3f::Exp(v)); }
}

namespace cv {
struct Scalar { double v; Scalar(double v_ = 0) : v(v_) {} };
enum { CV_WINDOW_AUTOSIZE_ = 1 };
inline void namedWindow(const char *, int) {}
inline void imshow(const char *, const Mat &) {}
inline int waitKey(int) { return 0; }
inline Mat operator*(const Mat &, int) { mini_cv_unsupported("Mat * s"); }
}  // namespace cv
#define CV_WINDOW_AUTOSIZE 1
#endif
