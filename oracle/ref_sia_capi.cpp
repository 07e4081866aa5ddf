// oracle/ref_sia_capi.cpp -- TEST INFRASTRUCTURE: the REFERENCE's ygz::SparseImgAlign (src/SparseImageAlign.cc with include/NLSSolver.h,
// include/NLSSolver_impl.hpp, include/SparseImageAlign.h, compiled where they lie into oracle/_ref/libref_orbmatcher.so) behind the oracle's
// flat entry point yo_sparse_img_align.  The Gauss-Newton driver, the level loop, the visibility bookkeeping, the patch / Jacobian caches,
// the residual loop and the stop / rollback rules are the reference's own code; the matrix template underneath (oracle/ref_shim/
// sia_stubs.h) evaluates its expressions element by element in natural order and delegates ldlt().solve and SE3f::exp / * / inverse to the
// oracle's restatements, so those stay unpinned.
#include <opencv2/core/core.hpp>

#include "SparseImageAlign.h"

namespace {
struct yo_align_frame {  // as in oracle_capi.cpp
    int N;
    const ygzo::KeyPoint *keys;
    const uint8_t *mp_valid, *outlier;
    const float *mp_world;
    float Tcw[7];
    int nlevels;
    const uint8_t *const *levels;
    const int *level_w, *level_h;
    const float *invScaleFactors;
    float fx, fy, cx, cy;
};

struct CountingAlign : ygz::SparseImgAlign {
    int linearisations = 0;
    CountingAlign(int max_level, int min_level, int n_iter) : ygz::SparseImgAlign(max_level, min_level, n_iter, GaussNewton, false, false) {}
    float computeResiduals(const SE3f &model, bool linearize_system, bool compute_weight_scale = false) override {
        if (linearize_system) linearisations++;
        return ygz::SparseImgAlign::computeResiduals(model, linearize_system, compute_weight_scale);
    }
    float chi2() const { return chi2_; }
    float H(int r, int c) const { return H_(r, c); }
};

void fill(ygz::Frame &F, std::vector<ygz::MapPoint> &mps, const yo_align_frame *f) {
    F.N = f->N;
    mps.resize((size_t) std::max(f->N, 1));
    F.mvpMapPoints.assign(f->N, (ygz::MapPoint *) nullptr);
    F.mvbOutlier.assign(f->N, false);
    for (int i = 0; i < f->N; i++) {
        F.mvKeys.push_back(cv::KeyPoint(f->keys[i].x, f->keys[i].y, f->keys[i].size, f->keys[i].angle, f->keys[i].response, f->keys[i].octave, f->keys[i].class_id));
        mps[i].mWorldPos = Vector3f(f->mp_world[3 * i], f->mp_world[3 * i + 1], f->mp_world[3 * i + 2]);
        if (f->mp_valid[i]) F.mvpMapPoints[i] = &mps[i];     // mp_valid = non-null and not bad
        F.mvbOutlier[i] = f->outlier[i] != 0;
    }
    ygzo::SE3f q;
    std::memcpy(q.q, f->Tcw, 16);
    std::memcpy(q.t, f->Tcw + 4, 12);
    F.mTcw = SE3f(q);
    for (int l = 0; l < f->nlevels; l++) {
        cv::Mat m(f->level_h[l], f->level_w[l], CV_8U);
        std::memcpy(m.data, f->levels[l], (size_t) f->level_w[l] * f->level_h[l]);
        F.mvImagePyramid.push_back(m);
        F.mvInvScaleFactors.push_back(f->invScaleFactors[l]);
    }
    F.fx = f->fx; F.fy = f->fy; F.cx = f->cx; F.cy = f->cy;
}
}  // namespace

extern "C" size_t yo_sparse_img_align(const yo_align_frame *ref, const yo_align_frame *cur, int max_level, int min_level, int n_iter, float out7[7],
                                      float info[2], float H36[36]) {
    ygz::Frame R, C;
    std::vector<ygz::MapPoint> rm, cm;
    fill(R, rm, ref);
    fill(C, cm, cur);
    CountingAlign A(max_level, min_level, n_iter);
    SE3f TCR;
    const size_t ret = A.run(&R, &C, TCR);
    std::memcpy(out7, TCR.q.q, 16);
    std::memcpy(out7 + 4, TCR.q.t, 12);
    if (info) { info[0] = (float) A.linearisations; info[1] = A.chi2(); }
    if (H36)
        for (int r = 0; r < 6; r++)
            for (int c = 0; c < 6; c++) H36[6 * r + c] = A.H(r, c);
    return ret;
}
