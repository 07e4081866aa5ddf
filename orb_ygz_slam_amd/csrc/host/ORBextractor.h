// ORBextractor.h -- ygz::ORBextractor with the reference's public interface (include/ORBextractor.h:45-109), implemented
// as a thin shell over the C ABI of libygzf (include/ygzf.h).  Tracking.cc / Frame.cc of the reference compile and link
// against this class unchanged; see INTEGRATION.md.
#ifndef YGZF_HOST_ORBEXTRACTOR_H
#define YGZF_HOST_ORBEXTRACTOR_H
#include <vector>

#include "ygz_compat.h"

struct ygzf_ctx;

namespace ygz {
class Frame;

class ORBextractor {
public:
    enum { HARRIS_SCORE = 0, FAST_SCORE = 1 };
    typedef enum { ORBSLAM_KEYPOINT, FAST_KEYPOINT, DSO_KEYPOINT } KeyPointMethod;

    ORBextractor(int nfeatures, float scaleFactor, int nlevels, int iniThFAST, int minThFAST);
    ~ORBextractor();
    ORBextractor(const ORBextractor &) = delete;
    ORBextractor &operator=(const ORBextractor &) = delete;

    // Compute the ORB features and descriptors on an image (mask ignored, as in the reference).
    void operator()(cv::InputArray image, cv::InputArray mask, std::vector<cv::KeyPoint> &keypoints, cv::OutputArray descriptors);
    // The overload Frame::ExtractORB uses.
    void operator()(Frame *frame, std::vector<cv::KeyPoint> &keypoints, cv::OutputArray descriptors, KeyPointMethod method,
                    bool leftEye = true);

    int inline GetLevels() { return nlevels; }
    float inline GetScaleFactor() { return scaleFactor; }
    std::vector<float> inline GetScaleFactors() { return mvScaleFactor; }
    std::vector<float> inline GetInverseScaleFactors() { return mvInvScaleFactor; }
    std::vector<float> inline GetScaleSigmaSquares() { return mvLevelSigma2; }
    std::vector<float> inline GetInverseScaleSigmaSquares() { return mvInvLevelSigma2; }

    // Host-readable 8-bit levels (Frame clones them; ComputeStereoMatches reads them directly).
    std::vector<cv::Mat> mvImagePyramid;
    void ComputePyramid(cv::Mat image);

    // Not in the reference class: the body Frame::ComputeStereoMatches (src/Frame.cc:509-682) is bound to.
    void ComputeStereoMatches(Frame &F);

    // Device on which new extractors create their context (default 0); set before constructing.
    static int sDevice;

protected:
    int nfeatures;
    double scaleFactor;
    int nlevels;
    int iniThFAST, minThFAST;
    std::vector<int> mnFeaturesPerLevel;
    std::vector<float> mvScaleFactor, mvInvScaleFactor, mvLevelSigma2, mvInvLevelSigma2;
    int mnGridSize = -1;   // dynamic grid size of the DSO_KEYPOINT path (include/ORBextractor.h:171), persists across frames

private:
    ygzf_ctx *ensureContext(int w, int h);
    ygzf_ctx *mCtx = nullptr;
    int mCtxW = 0, mCtxH = 0;
};
}  // namespace ygz
#endif
