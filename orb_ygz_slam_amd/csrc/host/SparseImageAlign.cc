// SparseImageAlign.cc -- shell of ygz::SparseImgAlign::run over libygzf's C ABI (product code, host side).
#include "SparseImageAlign.h"

#include <cstdio>
#include <vector>

#include "../../../include/ygzf.h"

namespace ygz {

int SparseImgAlign::sDevice = 0;

SparseImgAlign::SparseImgAlign(int max_level, int min_level, int n_iter, Method method, bool, bool)
    : n_iter_(n_iter), max_level_(max_level), min_level_(min_level) {
    for (float &v : H_) v = 0.f;
    if (method != GaussNewton) fprintf(stderr, "ygz::SparseImgAlign: only GaussNewton (the reference's only use) is implemented\n");
}

SparseImgAlign::~SparseImgAlign() { ygzf_destroy(ctx_); }

size_t SparseImgAlign::run(Frame *ref, Frame *cur, SE3f &TCR) {
    if (ref->mvKeys.empty()) {
        fprintf(stderr, "SparseImgAlign: no features to track!\n");
        return 0;
    }
    if (!ctx_) {
        ygzf_extractor_cfg cfg = {1000, 1.2f, 8, 20, 7};   // only the stream and scratch buffers of the context are used
        if (ygzf_create(sDevice, &cfg, 64, 64, 1, &ctx_) != YGZF_OK) {
            fprintf(stderr, "ygz::SparseImgAlign: %s\n", ygzf_last_error(nullptr));
            ctx_ = nullptr;
            return 0;
        }
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
    auto fill = [](Frame *f, ygzf_sia_frame &o, std::vector<const uint8_t *> &lv, std::vector<int> &w, std::vector<int> &h) {
        const int L = (int) f->mvImagePyramid.size();
        lv.resize(L); w.resize(L); h.resize(L);
        for (int l = 0; l < L; l++) { lv[l] = f->mvImagePyramid[l].ptr<uint8_t>(0); w[l] = f->mvImagePyramid[l].cols; h[l] = f->mvImagePyramid[l].rows; }
        o.nlevels = L; o.levels = lv.data(); o.level_w = w.data(); o.level_h = h.data();
        ygz_compat::se3_to7(f->mTcw, o.Tcw);
    };
    ygzf_sia_frame R{}, C{};
    std::vector<const uint8_t *> lr, lc;
    std::vector<int> wr, hr, wc, hc;
    fill(ref, R, lr, wr, hr);
    fill(cur, C, lc, wc, hc);
    R.n = N;
    R.keys = (const ygzf_kp *) ref->mvKeys.data();
    R.mp_valid = valid.data();
    R.outlier = outl.data();
    R.mp_world = world.data();
    ygzf_camera cam = {Frame::fx, Frame::fy, Frame::cx, Frame::cy, 0, 0, Frame::mnMinX, Frame::mnMinY, Frame::mnMaxX, Frame::mnMaxY};
    float T7[7];
    size_t ret = 0;
    if (ygzf_sia_run(ctx_, &R, &C, &cam, ref->mvInvScaleFactors.data(), max_level_, min_level_, n_iter_, T7, &ret, nullptr, H_) != YGZF_OK) {
        fprintf(stderr, "ygz::SparseImgAlign::run: %s\n", ygzf_last_error(ctx_));
        return 0;
    }
    TCR = ygz_compat::se3_from7(T7);
    return ret;
}

void SparseImgAlign::getFisherInformation(float out36[36]) const {
    const float sigma_i_sq = 5e-4f * 255 * 255;   // image noise, src/SparseImageAlign.cc:52
    for (int i = 0; i < 36; i++) out36[i] = H_[i] / sigma_i_sq;
}

}  // namespace ygz
