// standalone/SparseImageAlign.h -- ygz::SparseImgAlign's public interface (reference include/SparseImageAlign.h:14-55 and the public knobs
// it inherits from NLLSSolver<6, SE3f>, include/NLSSolver.h:95-163) for builds WITHOUT the reference tree.  Inside the reference tree this
// file is not used: SparseImageAlign.cc is compiled against the reference's own, unchanged include/SparseImageAlign.h + NLSSolver.h and
// defines that class's members over libygzf.
#ifndef YGZ_SPARSE_IMAGE_ALIGN_
#define YGZ_SPARSE_IMAGE_ALIGN_
#include <cstddef>

#include "ygz_compat.h"

namespace ygz {
struct Matrix6f {   // stand-in for Eigen::Matrix<float, 6, 6>
    float m[36];
    float operator()(int r, int c) const { return m[6 * r + c]; }
    float &operator()(int r, int c) { return m[6 * r + c]; }
};

class SparseImgAlign {
public:
    enum Method { GaussNewton, LevenbergMarquardt };
    cv::Mat resimg_;
    SparseImgAlign(int n_levels, int min_level, int n_iter = 10, Method method = GaussNewton, bool display = false, bool verbose = false);
    // Relative motion T_cur_from_ref; returns n_meas/16 (0 = failure, src/Tracking.cc:2089).
    size_t run(Frame *ref_frame, Frame *cur_frame, SE3f &TCR);
    // H_ / (5e-4 * 255^2): the Hessian of the log-likelihood at the converged state
    Matrix6f getFisherInformation();

    // public knobs of NLLSSolver the callers may touch
    size_t n_iter_init_, n_iter_;
    size_t n_meas_ = 0;
    bool verbose_;
    float eps_;

protected:
    Matrix6f H_;
    Method method_;
    bool display_;
    int max_level_, min_level_;
};
}  // namespace ygz
#endif
