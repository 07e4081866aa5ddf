// tests/cpp/mini_cv_nocompute.cpp -- TEST INFRASTRUCTURE for the boundary build (tests/cpp/build_boundary.sh).
// oracle/ref_shim/mini_cv.h stands in for the OpenCV headers (OpenCV is not installed here).  In the oracle's builds of the reference its
// compute primitives (cv::resize, cv::GaussianBlur, cv::FAST, cv::fastAtan2, cv::copyMakeBorder) are bodies on top of the oracle's
// restatements; in the BOUNDARY build the reference's src/ORBextractor.cc is replaced by the product's class shell over libygzf, so none
// of them may ever run: every one aborts.  A green boundary test therefore proves that pyramids, keypoints and descriptors came out of
// the HIP library and not out of any CPU code path.
#include "mini_cv.h"

#include <cmath>

int cvRound(double v) { return (int) std::nearbyint(v); }

namespace cv {
[[noreturn]] static void reached(const char *what) {
    std::fprintf(stderr, "boundary build: cv::%s was called -- a CPU compute primitive ran where only the HIP path may\n", what);
    std::abort();
}
void resize(InputArray, OutputArray, Size, double, double, int) { reached("resize"); }
void GaussianBlur(InputArray, OutputArray, Size, double, double, int) { reached("GaussianBlur"); }
void copyMakeBorder(InputArray, OutputArray, int, int, int, int, int) { reached("copyMakeBorder"); }
void FAST(InputArray, std::vector<KeyPoint> &, int, bool) { reached("FAST"); }
float fastAtan2(float, float) { reached("fastAtan2"); }
void KeyPointsFilter::retainBest(std::vector<KeyPoint> &, int) { reached("KeyPointsFilter::retainBest"); }
}  // namespace cv
