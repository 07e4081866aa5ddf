// ORBextractor.cc -- shell of ygz::ORBextractor over libygzf's C ABI (product code, host side).
// Error convention follows the reference: void returns, diagnostics on stderr (the reference uses glog), empty image =>
// silent return (src/ORBextractor.cc:972-973), zero keypoints => descriptors.release() (:990-991).
#include "ORBextractor.h"

#include "ygz_compat.h"   // reference tree: Frame.h / MapPoint.h / KeyFrame.h; stand-alone: the minimal types

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>

#include "../../../include/ygzf.h"
#include "ygzf_pool.h"
#if defined(YGZF_WITH_REFERENCE_HEADERS) && !defined(YGZ_MINI_CV)
#define YGZF_BLUR_PROBE_WITH_CV 1   // a real OpenCV is in scope: the probe may call cv::GaussianBlur
#endif
#include "cv_blur_probe.h"

namespace ygz {

int ORBextractor::sDevice = 0;
bool ORBextractor::sExtractAhead = true;

// Which generation of cv::GaussianBlur's 8-bit arithmetic the descriptors follow (ygzf_cv_mode).  Built against a real OpenCV the default is
// "detect": the first extractor of the process blurs a 16 x 12 probe image with the OpenCV it is linked against and takes the generation whose
// integers come back (cv_blur_probe.h) -- a user whose OpenCV is >= 3.4.11 gets the Q8.8 kernel's descriptors without knowing that there was
// anything to set; an OpenCV that matches none of the three (an IPP / OpenCL dispatch of GaussianBlur) is reported once and generation 0 is used.
// Over the stand-in headers (this repository's tests) there is nothing to probe: generation 0, the versions the reference names.
#if defined(YGZF_WITH_REFERENCE_HEADERS) && !defined(YGZ_MINI_CV)
int ORBextractor::sCvMode = -1;
static int detect_cv_blur_generation() {
    static int cached = -2;
    if (cached != -2) return cached;
    cached = ygzf_host::blur_probe_run_opencv();
    if (cached < 0) {
        ygzf_host::report_failure("ygz::ORBextractor", "this OpenCV's 8-bit GaussianBlur matches none of the three known generations (IPP / OpenCL dispatch?): "
                                                      "descriptors follow OpenCV 2.4 / 3.2 on x86 and may differ from the CPU path's in their last bit; set ORBextractor::sCvMode to choose");
        cached = 0;
    }
    return cached;
}
static int resolve_cv_mode(int m) { return m < 0 ? detect_cv_blur_generation() : m; }
#else
int ORBextractor::sCvMode = 0;
static int resolve_cv_mode(int m) { return m < 0 ? 0 : m; }
#endif

// src/ORBextractor.cc:412-470: the scale / sigma / per-level quota tables exist as soon as the object does -- Frame's constructors read
// them before the first image is processed (src/Frame.cc:119-125 vs :148) -- so they are computed here on the host (ygzf_scale_tables_host:
// the same arithmetic the device context uses), no device needed.
ORBextractor::ORBextractor(int _nfeatures, float _scaleFactor, int _nlevels, int _iniThFAST, int _minThFAST)
    : nfeatures(_nfeatures), scaleFactor(_scaleFactor), nlevels(_nlevels), iniThFAST(_iniThFAST), minThFAST(_minThFAST),
      mDevice(sDevice), mCvMode(resolve_cv_mode(sCvMode)), mExtractAhead(sExtractAhead) {
    const int L = std::max(nlevels, 0);
    mvScaleFactor.resize(L); mvInvScaleFactor.resize(L); mvLevelSigma2.resize(L); mvInvLevelSigma2.resize(L);
    mnFeaturesPerLevel.resize(L);
    mvImagePyramid.resize(L);
    ygzf_extractor_cfg cfg = {nfeatures, (float) scaleFactor, nlevels, iniThFAST, minThFAST, mCvMode};
    if (L > 0 && ygzf_scale_tables_host(&cfg, mvScaleFactor.data(), mvInvScaleFactor.data(), mvLevelSigma2.data(), mvInvLevelSigma2.data(),
                                        mnFeaturesPerLevel.data()) != YGZF_OK)
    {
        char msg[128];
        snprintf(msg, sizeof msg, "bad configuration (nfeatures %d, scaleFactor %g, nlevels %d)", nfeatures, (double) scaleFactor, nlevels);
        ygzf_host::report_failure("ygz::ORBextractor", msg);
    }
    // row ends of the 31-px circular patch (:455-469), kept for parity with the reference's member; the device holds its own copy
    const int HALF_PATCH_SIZE = 15;
    umax.assign(HALF_PATCH_SIZE + 1, 0);
    const int vmax = (int) std::floor(HALF_PATCH_SIZE * std::sqrt(2.f) / 2 + 1), vmin = (int) std::ceil(HALF_PATCH_SIZE * std::sqrt(2.f) / 2);
    for (int v = 0; v <= vmax; ++v) umax[v] = (int) std::nearbyint(std::sqrt((double) HALF_PATCH_SIZE * HALF_PATCH_SIZE - v * v));
    for (int v = HALF_PATCH_SIZE, v0 = 0; v >= vmin; --v) {
        while (umax[v0] == umax[v0 + 1]) ++v0;
        umax[v] = v0;
        ++v0;
    }
}

ORBextractor::~ORBextractor() { ygzf_destroy(mCtx); }

// Contexts are sized for the first image seen and re-created if a larger one arrives (cameras do not change size mid-run).
ygzf_ctx *ORBextractor::ensureContext(int w, int h) {
    if (mCtx && w <= mCtxW && h <= mCtxH) return mCtx;
    ygzf_destroy(mCtx);
    mCtx = nullptr;
    ygzf_extractor_cfg cfg = {nfeatures, (float) scaleFactor, nlevels, iniThFAST, minThFAST, mCvMode};
    if (ygzf_create(mDevice, &cfg, w, h, 2, &mCtx) != YGZF_OK) {   // 2 frames: the stereo matcher stages both eyes
        ygzf_host::report_failure("ygz::ORBextractor", ygzf_last_error(nullptr));
        mCtx = nullptr;
        return nullptr;
    }
    mAheadOn = false;
    mCtxW = w;
    mCtxH = h;
    return mCtx;
}

ygzf_ctx *ORBextractor::ResidentContext(const cv::Mat &level0) const {
    if (!mCtx || mHeldImage.empty() || level0.empty() || level0.cols != mHeldImage.cols || level0.rows != mHeldImage.rows ||
        !ygzf_has_resident_image(mCtx, level0.cols, level0.rows))
        return nullptr;
    if (level0.data == mVerifiedData || level0.data == mHeldImage.data) return mCtx;     // asked before (or the very buffer that went up)
    // nothing but the content says that this Frame's level 0 is the image the context holds: every pixel is compared (a Frame's level 0 is a
    // deep clone of the extractor's, src/Frame.cc:812)
    for (int y = 0; y < level0.rows; y++)
        if (std::memcmp(level0.ptr(y), mHeldImage.ptr(y), (size_t) level0.cols) != 0) return nullptr;
    mVerifiedData = level0.data;
    return mCtx;
}

cv::Mat ORBextractor::acquireLevel(int level, int rows, int cols) {
    constexpr size_t kGenerations = 4;   // the pyramid in mvImagePyramid, the one a caller may still be cloning from, and slack
    if ((int) mLevelPool.size() < nlevels) mLevelPool.resize(nlevels);
    std::vector<cv::Mat> &pool = mLevelPool[level];
    for (size_t i = 0; i < pool.size(); i++) {
        if (pool[i].rows != rows || pool[i].cols != cols) {                 // another image size: drop the entry (its buffer lives on with whoever holds it)
            pool.erase(pool.begin() + i--);
            continue;
        }
        if (ygz_compat::mat_refcount(pool[i]) == 1) return pool[i];      // only the pool refers to it
    }
    cv::Mat m(rows, cols, CV_8UC1);
    if (pool.size() < kGenerations) pool.push_back(m);
    return m;
}

void ORBextractor::ComputePyramid(cv::Mat image) {
    if (image.empty()) return;
    ygzf_ctx *c = ensureContext(image.cols, image.rows);
    if (!c) return;
    std::vector<uint8_t *> out(nlevels);
    mResidentLevel0 = cv::Mat();
    for (int l = 0; l < nlevels; l++) {
        int lw, lh;
        ygzf_level_size(c, image.cols, image.rows, l, &lw, &lh);
        mvImagePyramid[l] = cv::Mat();                  // this pyramid's own hold on its previous level buffer goes first
        mvImagePyramid[l] = acquireLevel(l, lh, lw);    // a buffer nobody else refers to: Frames keep (shared) references to earlier levels
        out[l] = mvImagePyramid[l].data;
    }
    mHeldImage = image;
    mVerifiedData = nullptr;
    {   // extract-ahead follows the tracker's habit: on when the previous pyramid's image was extracted, off when it was not
        const bool want = mExtractAhead && mExtractedSincePyramid;
        if (want != mAheadOn) {
            if (ygzf_set_extract_ahead(c, want ? 1 : 0) != YGZF_OK) ygzf_host::report_failure("ygz::ORBextractor", ygzf_last_error(c));
            else mAheadOn = want;
        }
        mExtractedSincePyramid = false;
    }
    if (ygzf_compute_pyramid(c, image.data, image.cols, image.rows, (int) image.step, out.data()) != YGZF_OK) {
        ygzf_host::report_failure("ygz::ORBextractor::ComputePyramid", ygzf_last_error(c));
        return;
    }
    mResidentLevel0 = mvImagePyramid[0];   // the context still holds this image and its pyramid (see operator()(Frame*, ...))
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
    mResidentLevel0 = cv::Mat();
    mHeldImage = image;
    mVerifiedData = nullptr;
    if (ygzf_extract(c, image.data, image.cols, image.rows, (int) image.step, (ygzf_kp *) _keypoints.data(), desc.data(), cap, &n) !=
        YGZF_OK) {
        ygzf_host::report_failure("ygz::ORBextractor::operator()", ygzf_last_error(c));
        n = 0;
    }
    _keypoints.resize(n);
    if (n == 0) {
        _descriptors.release();
        return;
    }
    _descriptors.create(n, 32, CV_8U);
    cv::Mat d = _descriptors.getMat();
    for (int i = 0; i < n; i++) std::memcpy(d.ptr(i), &desc[(size_t) i * 32], 32);
}

// src/ORBextractor.cc:1031-1127, the overload Frame::ExtractORB calls (src/Frame.cc:332-348):
//   ORBSLAM_KEYPOINT  octree keypoints on all levels; keys the frame already holds (direct-tracked) keep their angle and get
//                     descriptors first (:1093-1106)
//   DSO_KEYPOINT      FAST-10 grid keypoints on level 0 beside the frame's existing keys (ComputeKeyPointsDSOSingleLevel)
//   FAST_KEYPOINT     ComputeKeyPointsFast (:1189-1273): one Shi-Tomasi winner per 5-px cell over the libfast corners of all levels; never
//                     requested by the reference's own callers and marked "has a bug ... don't call" by its author (:1191) -- provided with
//                     the undefined parts defined (include/ygzf.h, ygzf_extract_fast_keypoint)
// There is no CPU fallback: on a device error the outputs stay untouched and the error is printed.
void ORBextractor::operator()(Frame *frame, std::vector<cv::KeyPoint> &_keypoints, cv::OutputArray _descriptors, KeyPointMethod method,
                              bool leftEye) {
    static_assert(sizeof(cv::KeyPoint) == sizeof(ygzf_kp), "cv::KeyPoint layout");
    const cv::Mat &img = leftEye ? (frame->mvImagePyramid.empty() ? frame->mImGray : frame->mvImagePyramid[0]) : frame->mImRight;
    if (!leftEye) ComputePyramid(img);          // right eye: the extractor's own pyramid is read by ComputeStereoMatches
    else mvImagePyramid = frame->mvImagePyramid;
    const cv::Mat resident_ = mResidentLevel0;   // (ComputePyramid above, or the Frame constructor's call, left it)
    const int N = leftEye ? frame->N : 0;       // existing keys of the frame (:1090-1092)
    if (img.empty()) return;
    ygzf_ctx *c = ensureContext(img.cols, img.rows);
    if (!c) return;
    std::vector<cv::KeyPoint> fresh;            // the new keypoints
    std::vector<uint8_t> descExisting((size_t) N * 32), descNew;
    if (method == FAST_KEYPOINT) {
        const int cap = N + (img.cols / 5) * (img.rows / 5) + 16;
        std::vector<cv::KeyPoint> all(cap);
        std::vector<uint8_t> d((size_t) cap * 32);
        for (int i = 0; i < N; i++) all[i] = frame->mvKeys[i];   // right eye: ComputeKeyPointsFast gets an empty list (:1049-1050)
        int total = 0;
        mHeldImage = cv::Mat();   // (the grid detectors keep no complete pyramid on the device)
        if (ygzf_extract_fast_keypoint(c, img.data, img.cols, img.rows, (int) img.step, (ygzf_kp *) all.data(), N, cap, d.data(), &total) != YGZF_OK) {
            ygzf_host::report_failure("ygz::ORBextractor (FAST_KEYPOINT)", ygzf_last_error(c));
            return;
        }
        for (int i = 0; i < N; i++) frame->mvKeys[i].angle = all[i].angle;   // ComputeKeyPointsFast re-orients them (:1268-1271)
        fresh.assign(all.begin() + N, all.begin() + total);
        std::memcpy(descExisting.data(), d.data(), (size_t) N * 32);
        descNew.assign(d.begin() + (size_t) N * 32, d.begin() + (size_t) total * 32);
        mResidentLevel0 = cv::Mat();
    } else if (method == DSO_KEYPOINT) {
        const int g0 = mnGridSize > 0 ? mnGridSize : std::max(1, (int) std::sqrt(1.0 * img.rows * img.cols / std::max(nfeatures, 1)));
        const int gm = std::max(1, std::min(7, g0));   // smallest grid the retry loop can reach
        const int cap = N + 3 * (img.cols / gm) * (img.rows / gm) + 16;
        std::vector<cv::KeyPoint> all(cap);
        std::vector<uint8_t> d((size_t) cap * 32);
        for (int i = 0; i < N; i++) all[i] = frame->mvKeys[i];
        int total = 0;
        mHeldImage = cv::Mat();
        if (ygzf_extract_dso(c, img.data, img.cols, img.rows, (int) img.step, (ygzf_kp *) all.data(), N, cap, d.data(), &mnGridSize, &total) !=
            YGZF_OK) {
            ygzf_host::report_failure("ygz::ORBextractor (DSO_KEYPOINT)", ygzf_last_error(c));
            return;
        }
        for (int i = 0; i < N; i++) frame->mvKeys[i].angle = all[i].angle;   // ComputeKeyPointsDSOSingleLevel re-orients them (:1380-1383)
        fresh.assign(all.begin() + N, all.begin() + total);
        std::memcpy(descExisting.data(), d.data(), (size_t) N * 32);
        descNew.assign(d.begin() + (size_t) N * 32, d.begin() + (size_t) total * 32);
    } else {
        const int cap = ygzf_max_keypoints(c, img.cols, img.rows);
        fresh.resize(cap > 0 ? cap : 0);
        descNew.resize((size_t) (cap > 0 ? cap : 0) * 32);
        int n = 0;
        // Frame's constructors call ComputePyramid(image) and then this overload on a clone of the very pyramid it produced
        // (src/Frame.cc:807-813, :332-348): when the frame's level 0 still equals the image whose pyramid the context holds (one
        // 360 KB comparison), FAST / octree / descriptors run on the resident pyramid -- no second upload, no second pyramid.
        bool resident = false;
        if (!resident_.empty() && resident_.cols == img.cols && resident_.rows == img.rows) {
            resident = true;
            for (int y = 0; y < img.rows && resident; y++) resident = std::memcmp(img.ptr(y), resident_.ptr(y), (size_t) img.cols) == 0;
        }
        int rcE = YGZF_ERR_STATE;
        if (resident) rcE = ygzf_extract_resident(c, (ygzf_kp *) fresh.data(), descNew.data(), cap, &n);
        if (resident && rcE == YGZF_OK) mExtractedSincePyramid = true;
        if (rcE == YGZF_ERR_STATE) {   // nothing resident (another image operation came in between): the image goes up again
            mHeldImage = img;
            mVerifiedData = nullptr;
            rcE = ygzf_extract(c, img.data, img.cols, img.rows, (int) img.step, (ygzf_kp *) fresh.data(), descNew.data(), cap, &n);
        }
        mResidentLevel0 = cv::Mat();   // the extraction reuses the buffers: nothing is resident afterwards
        if (rcE != YGZF_OK) {
            ygzf_host::report_failure("ygz::ORBextractor (ORBSLAM_KEYPOINT)", ygzf_last_error(c));
            return;
        }
        fresh.resize(n);
        if (N > 0 && ygzf_describe_keys(c, 0, (const ygzf_kp *) frame->mvKeys.data(), N, 0, nullptr, descExisting.data()) != YGZF_OK) {
            ygzf_host::report_failure("ygz::ORBextractor (existing keys)", ygzf_last_error(c));
            return;
        }
    }
    const int nkeypoints = (int) fresh.size() + (int) _keypoints.size();   // :1065-1068
    if (nkeypoints == 0) {
        _descriptors.release();
        return;
    }
    _descriptors.create(nkeypoints, 32, CV_8U);
    cv::Mat d = _descriptors.getMat();
    for (int i = 0; i < N && i < nkeypoints; i++) std::memcpy(d.ptr(i), &descExisting[(size_t) i * 32], 32);
    for (int i = 0; i < (int) fresh.size() && N + i < nkeypoints; i++) std::memcpy(d.ptr<uint8_t>(N + i), &descNew[(size_t) i * 32], 32);   // offset = frame->N
    _keypoints.insert(_keypoints.end(), fresh.begin(), fresh.end());
}

// Body for ygz::Frame::ComputeStereoMatches (src/Frame.cc:509-682), see INTEGRATION.md: the left extractor's context stages both
// eyes' level-0 images, rebuilds their pyramids on the device and runs the row-band Hamming search + SAD refinement + median cut.
void ORBextractor::ComputeStereoMatches(Frame &F) {
    F.mvuRight = std::vector<float>(F.N, -1.0f);
    F.mvDepth = std::vector<float>(F.N, -1.0f);
    if (F.N <= 0) return;
    const cv::Mat &imL = F.mvImagePyramid.empty() ? F.mImGray : F.mvImagePyramid[0];
    ygzf_ctx *c = ensureContext(imL.cols, imL.rows);
    if (!c) return;
    const int Nr = (int) F.mvKeysRight.size();
    std::vector<uint8_t> dl((size_t) F.N * 32), dr((size_t) Nr * 32);
    for (int i = 0; i < F.N; i++) std::memcpy(&dl[(size_t) i * 32], F.mDescriptors.ptr(i), 32);
    for (int i = 0; i < Nr; i++) std::memcpy(&dr[(size_t) i * 32], F.mDescriptorsRight.ptr(i), 32);
    if (imL.step != F.mImRight.step) { ygzf_host::report_failure("ygz::ORBextractor::ComputeStereoMatches", "left/right row steps differ"); return; }
    mHeldImage = cv::Mat();
    if (ygzf_compute_stereo_matches(c, imL.data, F.mImRight.data, imL.cols, imL.rows, (int) imL.step, F.N,
                                    (const ygzf_kp *) F.mvKeys.data(), dl.data(), Nr, (const ygzf_kp *) F.mvKeysRight.data(), dr.data(), F.mb, F.mbf,
                                    F.mvuRight.data(), F.mvDepth.data()) != YGZF_OK)
        ygzf_host::report_failure("ygz::ORBextractor::ComputeStereoMatches", ygzf_last_error(c));
}

}  // namespace ygz
