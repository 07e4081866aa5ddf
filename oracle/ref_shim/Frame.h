// oracle/ref_shim/Frame.h -- TEST INFRASTRUCTURE: the four members of ygz::Frame that src/ORBextractor.cc reads (include/Frame.h
// itself needs Eigen / Sophus / DBoW2).  Found before the reference's own header through the include path of oracle/Makefile.
#ifndef YGZ_ORACLE_REF_SHIM_FRAME_H
#define YGZ_ORACLE_REF_SHIM_FRAME_H
#include "mini_cv.h"
#ifndef YGZ_REF_MATCHER   // the matcher / aligner builds get their Frame from matcher_stubs.h
namespace ygz {
class Frame {
public:
    cv::Mat mImRight;                        // include/Frame.h: right image of a stereo frame
    std::vector<cv::Mat> mvImagePyramid;     // pyramid of the left image (built by Frame::ComputeImagePyramid)
    std::vector<cv::KeyPoint> mvKeys;        // keypoints the frame already holds
    int N = 0;                               // their number
};
}  // namespace ygz
#endif
#ifdef YGZ_REF_FRAME     // src/ORBextractor.cc says #include "Frame.h" and finds this file first: hand over to the real header
#include_next "Frame.h"
#endif
#endif
