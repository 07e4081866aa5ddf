// ORBextractor.h -- ygz::ORBextractor over libygzf: the drop-in replacement of the reference's include/ORBextractor.h.
//
// It IS that header for the reference tree: same include guard (YGZ_ORBEXTRACTOR_H_), same public interface
// (include/ORBextractor.h:45-109: constructor, both operator() overloads, the six getters, the public mvImagePyramid, ComputePyramid,
// the two enums) and the reference's protected data members, so that Frame.h / Frame.cc / Tracking.cc compile against it unchanged.
// In the reference tree this file replaces include/ORBextractor.h and ORBextractor.cc replaces src/ORBextractor.cc (the
// boundary test builds exactly that: tests/cpp/build_boundary.sh compiles the reference's own src/Frame.cc against this header).
// What is not carried over are the protected CPU helpers (ComputeKeyPointsOctTree, DistributeOctTree, ... and class ExtractorNode): they
// are implementation details of the file this one replaces, nothing outside src/ORBextractor.cc refers to them.
//
// Stand-alone (this repository, no OpenCV / Eigen installed) the same class compiles over the minimal types of ygz_compat.h.
#ifndef YGZ_ORBEXTRACTOR_H_
#define YGZ_ORBEXTRACTOR_H_

#ifdef YGZF_WITH_REFERENCE_HEADERS
#include "Common.h"   // the reference's include/Common.h (OpenCV, Eigen, Sophus, glog), as its own ORBextractor.h does
#else
#include "ygz_compat.h"
#endif

#include <vector>

struct ygzf_ctx;

namespace ygz {

class Frame;

class ORBextractor {
public:
    enum { HARRIS_SCORE = 0, FAST_SCORE = 1 };

    // which detector operator()(Frame*, ...) runs: the ORB-SLAM octree detector, SVO's grid FAST (never requested by the reference and
    // marked "has a bug ... don't call" by its author, src/ORBextractor.cc:1191) or the DSO-like dynamic grid FAST
    typedef enum { ORBSLAM_KEYPOINT, FAST_KEYPOINT, DSO_KEYPOINT } KeyPointMethod;

    ORBextractor(int nfeatures, float scaleFactor, int nlevels, int iniThFAST, int minThFAST);

    ~ORBextractor();
    ORBextractor(const ORBextractor &) = delete;              // owns a device context
    ORBextractor &operator=(const ORBextractor &) = delete;

    // Compute the ORB features and descriptors on an image (mask ignored, as in the reference).
    void operator()(cv::InputArray image, cv::InputArray mask, std::vector<cv::KeyPoint> &keypoints, cv::OutputArray descriptors);

    // detect features for a frame: the overload Frame::ExtractORB uses
    void operator()(Frame *frame, std::vector<cv::KeyPoint> &keypoints, cv::OutputArray descriptors, KeyPointMethod method,
                    bool leftEye = true);

    int inline GetLevels() { return nlevels; }

    float inline GetScaleFactor() { return scaleFactor; }

    std::vector<float> inline GetScaleFactors() { return mvScaleFactor; }

    std::vector<float> inline GetInverseScaleFactors() { return mvInvScaleFactor; }

    std::vector<float> inline GetScaleSigmaSquares() { return mvLevelSigma2; }

    std::vector<float> inline GetInverseScaleSigmaSquares() { return mvInvLevelSigma2; }

    // Host-readable 8-bit levels (Frame clones them, src/Frame.cc:810-813; ComputeStereoMatches reads them directly, :515-623).
    std::vector<cv::Mat> mvImagePyramid;

    void ComputePyramid(cv::Mat image);

    // ---- additions (not in the reference class) -------------------------------------------------------------------------------------
    // Device form of Frame::ComputeStereoMatches (src/Frame.cc:509-682): same outputs (mvuRight, mvDepth); a one-line binding in
    // Frame.cc makes the reference use it, the reference's own CPU body keeps working on the pyramids above otherwise.
    void ComputeStereoMatches(Frame &F);
    // HIP device on which extractors constructed from now on create their context (default 0).
    static int sDevice;
    // ygzf_cv_mode of extractors constructed from now on: which OpenCV generation's 8-bit GaussianBlur the descriptors follow
    // (include/ygzf.h: 0 = OpenCV 2.4 / 3.2 on x86, the versions the reference names; 1 = the same without SSE2; 2 = >= 3.4.11 / 4.x).
    // -1 = detect: built against a real OpenCV (the default there) the first extractor blurs a probe image with the linked OpenCV and takes the
    // generation that comes back (host/cv_blur_probe.h); over the stand-in headers -1 means 0.
    static int sCvMode;
    // Extract-ahead (include/ygzf.h, ygzf_set_extract_ahead; default on): ComputePyramid queues the ORBSLAM_KEYPOINT extraction of the same image
    // behind the pyramid, so that it runs while the levels return and the Frame constructor clones them; operator()(Frame *, ...) on that image
    // then only collects keypoints and descriptors.  The extractor does this only while it pays: it queues ahead when the PREVIOUS pyramid
    // was followed by an extraction (a tracker that extracts on every frame), and not when it was not (direct tracking, where most frames
    // never extract and would pay ~45 us of launch time per ComputePyramid for device work nobody collects).  false: never.
    static bool sExtractAhead;
    // The context, when it still holds `level0` (the image of this extractor's last operation, compared by content fingerprint) together with
    // its pyramid on the device; nullptr otherwise.  The shells that read Frame images (SparseImgAlign::run, FindDirectProjection) pass it to
    // the image cache, which then copies device to device instead of uploading the image a second time.
    ygzf_ctx *ResidentContext(const cv::Mat &level0) const;

protected:
    int nfeatures = 0;
    double scaleFactor = 0;
    int nlevels = 0;
    int iniThFAST = 0;
    int minThFAST = 0;

    std::vector<int> mnFeaturesPerLevel;

    std::vector<int> umax;
    std::vector<float> mvScaleFactor;
    std::vector<float> mvInvScaleFactor;
    std::vector<float> mvLevelSigma2;
    std::vector<float> mvInvLevelSigma2;

    int mnGridSize = -1;   // dynamic grid size of the DSO_KEYPOINT detector (include/ORBextractor.h:171); -1: not yet set (:1298-1299)

private:
    ygzf_ctx *ensureContext(int w, int h);
    ygzf_ctx *mCtx = nullptr;
    cv::Mat mResidentLevel0;     // level 0 of the pyramid the context still holds on the device (set by ComputePyramid), or empty
    int mCtxW = 0, mCtxH = 0;
    int mDevice = 0, mCvMode = 0;
    bool mExtractAhead = true;
    bool mAheadOn = false;            // what the context is set to
    bool mExtractedSincePyramid = true;   // the previous ComputePyramid was followed by an extraction of its image (optimistic start)
    // The image of the last operation that sent ONE image to the context (a header sharing the caller's / the pyramid's buffer; empty: none), for
    // ResidentContext: "is this Frame's level 0 exactly what the device holds?" is answered by comparing the pixels themselves -- every one of
    // them, no hash, nothing to collide -- the first time a buffer is asked about, and by that buffer's address afterwards.
    cv::Mat mHeldImage;
    mutable const void *mVerifiedData = nullptr;
    // Level buffers of earlier pyramids, handed out again by ComputePyramid once nobody but this pool refers to them (cv::Mat's own reference
    // count decides: a Frame or KeyFrame that kept a header keeps the buffer).  A fresh 752x480 pyramid is 1.1 MB of never-touched pages:
    // ~270 page faults, more than the pyramid kernel takes -- a recycled buffer has none.
    cv::Mat acquireLevel(int level, int rows, int cols);
    std::vector<std::vector<cv::Mat>> mLevelPool;   // [level] -> a few generations
};

}  // namespace ygz

#endif
