// SparseImageAlign.h -- ygz::SparseImgAlign::run with the reference's signature (include/SparseImageAlign.h:14-51) over
// libygzf.  The reference class derives from NLLSSolver<6,SE3f>; only the members Tracking touches are kept public here
// (ctor, run, getFisherInformation); inside the reference tree the shell can equally be compiled as a member of the
// original class.
#ifndef YGZF_HOST_SPARSE_IMAGE_ALIGN_H
#define YGZF_HOST_SPARSE_IMAGE_ALIGN_H
#include <cstddef>

#include "ygz_compat.h"

struct ygzf_ctx;

namespace ygz {
class SparseImgAlign {
public:
    enum Method { GaussNewton, LevenbergMarquardt };
    SparseImgAlign(int max_level, int min_level, int n_iter = 10, Method method = GaussNewton, bool display = false, bool verbose = false);
    ~SparseImgAlign();
    // Relative motion T_cur_from_ref; returns n_meas/16 (0 = failure, src/Tracking.cc:2089).
    size_t run(Frame *ref_frame, Frame *cur_frame, SE3f &TCR);
    // H_ / (5e-4 * 255^2), row-major 6x6 (reference returns an Eigen matrix).
    void getFisherInformation(float out36[36]) const;

    int n_iter_;
    float eps_ = 0.000001f;
    static int sDevice;

protected:
    int max_level_, min_level_;
    float H_[36];
    ygzf_ctx *ctx_ = nullptr;
};
}  // namespace ygz
#endif
