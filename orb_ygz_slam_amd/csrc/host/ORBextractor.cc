// ORBextractor.cc -- shell of ygz::ORBextractor over libygzf's C ABI (product code, host side).
// Error convention follows the reference: void returns, diagnostics on stderr (the reference uses glog), empty image =>
// silent return (src/ORBextractor.cc:972-973), zero keypoints => descriptors.release() (:990-991).
#include "ORBextractor.h"

#include <cstdio>

#include "../../../include/ygzf.h"

namespace ygz {

int ORBextractor::sDevice = 0;

ORBextractor::ORBextractor(int _nfeatures, float _scaleFactor, int _nlevels, int _iniThFAST, int _minThFAST)
    : nfeatures(_nfeatures), scaleFactor(_scaleFactor), nlevels(_nlevels), iniThFAST(_iniThFAST), minThFAST(_minThFAST) {
    mvImagePyramid.resize(nlevels);
}

ORBextractor::~ORBextractor() { ygzf_destroy(mCtx); }

// Contexts are sized for the first image seen and re-created if a larger one arrives (cameras do not change size mid-run).
ygzf_ctx *ORBextractor::ensureContext(int w, int h) {
    if (mCtx && w <= mCtxW && h <= mCtxH) return mCtx;
    ygzf_destroy(mCtx);
    mCtx = nullptr;
    ygzf_extractor_cfg cfg = {nfeatures, (float) scaleFactor, nlevels, iniThFAST, minThFAST};
    if (ygzf_create(sDevice, &cfg, w, h, 1, &mCtx) != YGZF_OK) {
        fprintf(stderr, "ygz::ORBextractor: %s\n", ygzf_last_error(nullptr));
        mCtx = nullptr;
        return nullptr;
    }
    mCtxW = w;
    mCtxH = h;
    mvScaleFactor.resize(nlevels); mvInvScaleFactor.resize(nlevels); mvLevelSigma2.resize(nlevels); mvInvLevelSigma2.resize(nlevels);
    mnFeaturesPerLevel.resize(nlevels);
    ygzf_get_scale_tables(mCtx, mvScaleFactor.data(), mvInvScaleFactor.data(), mvLevelSigma2.data(), mvInvLevelSigma2.data());
    ygzf_get_features_per_level(mCtx, mnFeaturesPerLevel.data());
    return mCtx;
}

void ORBextractor::ComputePyramid(cv::Mat image) {
    if (image.empty()) return;
    ygzf_ctx *c = ensureContext(image.cols, image.rows);
    if (!c) return;
    std::vector<uint8_t *> out(nlevels);
    for (int l = 0; l < nlevels; l++) {
        int lw, lh;
        ygzf_level_size(c, image.cols, image.rows, l, &lw, &lh);
        mvImagePyramid[l] = cv::Mat(lh, lw, CV_8UC1);   // fresh buffer: Frames keep (shared) references to earlier levels
        out[l] = mvImagePyramid[l].ptr<uint8_t>(0);
    }
    if (ygzf_compute_pyramid(c, image.ptr<uint8_t>(0), image.cols, image.rows, (int) image.step, out.data()) != YGZF_OK)
        fprintf(stderr, "ygz::ORBextractor::ComputePyramid: %s\n", ygzf_last_error(c));
}

void ORBextractor::operator()(cv::InputArray _image, cv::InputArray, std::vector<cv::KeyPoint> &_keypoints, cv::OutputArray _descriptors) {
    if (_image.empty()) return;
    cv::Mat image = _image.getMat();
    ygzf_ctx *c = ensureContext(image.cols, image.rows);
    if (!c) return;
    const int cap = ygzf_max_keypoints(c, image.cols, image.rows);
    static_assert(sizeof(cv::KeyPoint) == sizeof(ygzf_kp), "cv::KeyPoint layout");
    _keypoints.resize(cap > 0 ? cap : 0);
    std::vector<uint8_t> desc((size_t) (cap > 0 ? cap : 0) * 32);
    int n = 0;
    if (ygzf_extract(c, image.ptr<uint8_t>(0), image.cols, image.rows, (int) image.step, (ygzf_kp *) _keypoints.data(), desc.data(), cap, &n) !=
        YGZF_OK) {
        fprintf(stderr, "ygz::ORBextractor::operator(): %s\n", ygzf_last_error(c));
        n = 0;
    }
    _keypoints.resize(n);
    if (n == 0) {
        _descriptors.release();
        return;
    }
    _descriptors.create(n, 32, CV_8U);
    cv::Mat d = _descriptors.getMat();
    for (int i = 0; i < n; i++) std::memcpy(d.ptr<uint8_t>(i), &desc[(size_t) i * 32], 32);
}

// src/ORBextractor.cc:1031-1127.  ORBSLAM_KEYPOINT on a frame without pre-existing keys (the path taken for initial
// frames, relocalisation, re-extraction after a direct-tracking failure and every right-eye image) runs on the device.
// FAST_KEYPOINT / DSO_KEYPOINT (libfast FAST-10 grid paths) and descriptors of pre-existing direct-tracked keys are the
// next rows of the scope table; they are reported, not silently emulated on the CPU.
void ORBextractor::operator()(Frame *frame, std::vector<cv::KeyPoint> &_keypoints, cv::OutputArray _descriptors, KeyPointMethod method,
                              bool leftEye) {
    if (method != ORBSLAM_KEYPOINT || (leftEye && frame->N > 0)) {
        fprintf(stderr, "ygz::ORBextractor: KeyPointMethod %d with %d pre-existing keys is not implemented on the device yet\n", (int) method,
                frame->N);
        return;
    }
    const cv::Mat &img = leftEye ? (frame->mvImagePyramid.empty() ? frame->mImGray : frame->mvImagePyramid[0]) : frame->mImRight;
    if (!leftEye) ComputePyramid(img);          // right eye: the extractor's own pyramid is read by ComputeStereoMatches
    else mvImagePyramid = frame->mvImagePyramid;
    cv::Mat mask;
    std::vector<cv::KeyPoint> kps;
    (*this)(cv::_InputArray(img), cv::_InputArray(mask), kps, _descriptors);
    _keypoints.insert(_keypoints.end(), kps.begin(), kps.end());
}

}  // namespace ygz
