// SparseImageAlign.cc -- ygz::SparseImgAlign over libygzf's C ABI (product code, host side): the replacement of the reference's
// src/SparseImageAlign.cc.  Inside the reference tree it defines the members of the reference's OWN class (include/SparseImageAlign.h,
// derived from NLLSSolver<6, SE3f>, unchanged): constructor, run, getFisherInformation, and the solver's virtual hooks, which are never
// entered because the whole Gauss-Newton loop (all levels x iterations, include/NLSSolver_impl.hpp:17-91) runs inside one kernel.
// Stand-alone it defines the same members of standalone/SparseImageAlign.h.
#include "ORBextractor.h"        // first: inside the reference tree this is the replacement header (same include guard)
#include "SparseImageAlign.h"    // the reference's own header (reference tree) or standalone/SparseImageAlign.h, by include path
#include "ygz_compat.h"

#include <cstdio>
#include <vector>

#include "../../../include/ygzf.h"
#include "ygzf_pool.h"

namespace ygz {

SparseImgAlign::SparseImgAlign(int max_level, int min_level, int n_iter, Method method, bool display, bool verbose)
    : display_(display), max_level_(max_level), min_level_(min_level) {   // src/SparseImageAlign.cc:7-18
    n_iter_ = n_iter;
    n_iter_init_ = n_iter_;
    method_ = method;
    verbose_ = verbose;
    eps_ = 0.000001;
    if (method != GaussNewton) fprintf(stderr, "ygz::SparseImgAlign: only GaussNewton (the reference's only use, src/Tracking.cc:207) runs on the device\n");
}

size_t SparseImgAlign::run(Frame *ref, Frame *cur, SE3f &TCR) {
    n_meas_ = 0;
    if (ref->mvKeys.empty()) {
        fprintf(stderr, "SparseImgAlign: no features to track!\n");   // the reference logs the same and returns 0 (:24-27)
        return 0;
    }
    const int N = ref->N;
    std::vector<uint8_t> valid(N), outl(N);
    std::vector<float> world((size_t) N * 3);
    for (int i = 0; i < N; i++) {
        MapPoint *mp = ref->mvpMapPoints[i];
        valid[i] = mp != nullptr && !mp->isBad();
        outl[i] = ref->mvbOutlier[i];
        if (mp) ygz_compat::world_pos(mp, &world[3 * (size_t) i]);
    }
    ygzf_sia_frame R{}, C{};
    R.n = N;
    R.keys = (const ygzf_kp *) ref->mvKeys.data();
    R.mp_valid = valid.data();
    R.outlier = outl.data();
    R.mp_world = world.data();
    ygz_compat::se3_to7(ref->mTcw, R.Tcw);
    ygz_compat::se3_to7(cur->mTcw, C.Tcw);
    ygzf_camera cam = {Frame::fx, Frame::fy, Frame::cx, Frame::cy, 0, 0, Frame::mnMinX, Frame::mnMinY, Frame::mnMaxX, Frame::mnMaxY};
    float T7[7], H36[36];
    size_t ret = 0;
    const int kIterations = 10;   // run() overrides the constructor's n_iter with iterations[level] = 10 on every level (:38-43)
    static const char *who = "ygz::SparseImgAlign::run";
    // The two frames' images live in the device-resident image cache the direct matcher uses (ygzf_host::ImageCache, keyed by (kind, id,
    // content fingerprint)): the reference frame of this call is a COPY of the previous call's current frame (mLastFrame = Frame(mCurrentFrame),
    // a deep clone of the pyramid: same id, same pixels, other address) and hits the slot that frame filled, so one level-0 upload per new
    // frame replaces two pyramid uploads per call
    // (their pyramids are rebuilt on the device by the extractor's resize kernel: the bytes the Frames' host pyramids hold).  Frames whose
    // pyramids cannot come from the cache (no level 0, unequal sizes) go up from host memory.
    const int L = (int) ref->mvImagePyramid.size();
    bool done = false;
    if (L >= 1 && (int) cur->mvImagePyramid.size() == L && (int) ref->mvScaleFactors.size() >= L && max_level_ < L &&
        ref->mvImagePyramid[0].cols == cur->mvImagePyramid[0].cols && ref->mvImagePyramid[0].rows == cur->mvImagePyramid[0].rows) {
        const cv::Mat &r0 = ref->mvImagePyramid[0], &c0 = cur->mvImagePyramid[0];
        ygzf_host::ImageCache &ic = ygzf_host::ImageCache::instance();
        ygzf_host::ImageCache::Guard lk(ic);
        if (ic.prepare(ORBextractor::sDevice, r0.cols, r0.rows, L, L > 1 ? ref->mvScaleFactors[1] : 1.2f, who)) {
            // (a miss is served device to device when the frame's extractor still holds the image it built the pyramid from -- the current
            // frame of a Tracking iteration, as a rule: src/Frame.cc:807 ran a moment ago)
            const int rs = ic.slot(ygzf_host::ImageCache::kFrame, ref->mnId, r0.data, r0.cols, r0.rows, (int) r0.step, who,
                                   ref->mpORBextractorLeft ? ref->mpORBextractorLeft->ResidentContext(r0) : nullptr);
            const int cs = ic.slot(ygzf_host::ImageCache::kFrame, cur->mnId, c0.data, c0.cols, c0.rows, (int) c0.step, who,
                                   cur->mpORBextractorLeft ? cur->mpORBextractorLeft->ResidentContext(c0) : nullptr);
            if (rs >= 0 && cs >= 0 && rs != cs) {
                if (ygzf_sia_run_cached(ic.ctx(), rs, cs, &R, C.Tcw, &cam, ref->mvInvScaleFactors.data(), max_level_, min_level_, kIterations, T7, &ret,
                                        nullptr, H36) != YGZF_OK) {
                    ygzf_host::report_failure(who, ygzf_last_error(ic.ctx()));
                    return 0;
                }
                done = true;
            }
        }
    }
    if (!done) {
        ygzf_host::Lease lease(ORBextractor::sDevice);
        if (!lease) return 0;
        ygzf_ctx *ctx = lease.get();
        auto fill = [](Frame *f, ygzf_sia_frame &o, std::vector<const uint8_t *> &lv, std::vector<int> &w, std::vector<int> &h) {
            const int n = (int) f->mvImagePyramid.size();
            lv.resize(n); w.resize(n); h.resize(n);
            for (int l = 0; l < n; l++) { lv[l] = f->mvImagePyramid[l].data; w[l] = f->mvImagePyramid[l].cols; h[l] = f->mvImagePyramid[l].rows; }
            o.nlevels = n; o.levels = lv.data(); o.level_w = w.data(); o.level_h = h.data();
        };
        std::vector<const uint8_t *> lr, lc;
        std::vector<int> wr, hr, wc, hc;
        fill(ref, R, lr, wr, hr);
        fill(cur, C, lc, wc, hc);
        if (ygzf_sia_run(ctx, &R, &C, &cam, ref->mvInvScaleFactors.data(), max_level_, min_level_, kIterations, T7, &ret, nullptr, H36) != YGZF_OK) {
            ygzf_host::report_failure(who, ygzf_last_error(ctx));
            return 0;
        }
    }
    n_iter_ = kIterations;
    for (int r = 0; r < 6; r++)
        for (int c = 0; c < 6; c++) H_(r, c) = H36[6 * r + c];
    n_meas_ = ret * 16;   // patch_area_
    TCR = ygz_compat::se3_from7(T7);   // the reference assigns T_cur_from_ref whenever it got past the feature check (:46)
    return ret;
}

#ifdef YGZF_WITH_REFERENCE_HEADERS
typedef Matrix<float, 6, 6> FisherMatrix;
#else
typedef Matrix6f FisherMatrix;
#endif

FisherMatrix SparseImgAlign::getFisherInformation() {   // :51-55
    const float sigma_i_sq = 5e-4 * 255 * 255;   // image noise
    FisherMatrix I;
    for (int r = 0; r < 6; r++)
        for (int c = 0; c < 6; c++) I(r, c) = H_(r, c) / sigma_i_sq;
    return I;
}

#ifdef YGZF_WITH_REFERENCE_HEADERS
// The solver hooks NLLSSolver<6, SE3f> declares pure virtual (include/NLSSolver.h:60-84): required by the class's vtable, not entered --
// run() above does not go through optimize().
void SparseImgAlign::precomputeReferencePatches() {}
float SparseImgAlign::computeResiduals(const SE3f &, bool, bool) { return 0.f; }
int SparseImgAlign::solve() { return 0; }
void SparseImgAlign::update(const ModelType &old_model, ModelType &new_model) { new_model = old_model; }
void SparseImgAlign::startIteration() {}
void SparseImgAlign::finishIteration() {}
#endif

}  // namespace ygz
