// oracle/ref_shim/mini_cv.cpp -- TEST INFRASTRUCTURE: bodies of the OpenCV stand-in on top of the oracle's primitive restatements
// (oracle_cvprims.cpp).  See mini_cv.h for what this arrangement pins.
#include "mini_cv.h"

#include "../ygz_oracle.h"

int cvRound(double v) { return ygzo::cv_round(v); }

namespace cv {

static ygzo::Image to_image(const Mat &m) {   // tight copy of a (possibly strided) view
    ygzo::Image im(m.cols, m.rows);
    for (int y = 0; y < m.rows; y++) std::memcpy(&im.d[(size_t) y * m.cols], m.ptr(y), (size_t) m.cols);
    return im;
}
static void from_image(const ygzo::Image &im, Mat &m) {
    for (int y = 0; y < im.h; y++) std::memcpy(m.ptr(y), &im.d[(size_t) y * im.w], (size_t) im.w);
}

// cv::resize, INTER_LINEAR, 8UC1.  dst keeps its buffer when it already has dsize (the reference resizes into a view of the bordered
// level buffer, src/ORBextractor.cc:1135-1139).
void resize(InputArray src_, OutputArray dst_, Size dsize, double, double, int interpolation) {
    assert(interpolation == INTER_LINEAR);
    const Mat src = src_.getMat();
    dst_.create(dsize.height, dsize.width, CV_8UC1);
    Mat dst = dst_.getMat();
    const ygzo::Image s = to_image(src);
    ygzo::Image d(dsize.width, dsize.height);
    ygzo::resize_linear_u8(s, d);
    from_image(d, dst);
}

// cv::GaussianBlur, 7x7, sigma 2, BORDER_REFLECT_101 (the only form the reference uses, :1010, :1083)
void GaussianBlur(InputArray src_, OutputArray dst_, Size ksize, double sigmaX, double sigmaY, int borderType) {
    assert(ksize.width == 7 && ksize.height == 7 && sigmaX == 2 && sigmaY == 2 && borderType == BORDER_REFLECT_101);
    const Mat src = src_.getMat();
    const ygzo::Image s = to_image(src);
    ygzo::Image d(s.w, s.h);
    ygzo::gaussian_blur7_s2_u8(s, d);
    dst_.create(s.h, s.w, CV_8UC1);
    Mat dst = dst_.getMat();
    from_image(d, dst);
}

static inline int reflect101(int i, int n) {
    if (n == 1) return 0;
    while (i < 0 || i >= n) i = i < 0 ? -i : 2 * n - 2 - i;
    return i;
}
// cv::copyMakeBorder, BORDER_REFLECT_101 (+ BORDER_ISOLATED).  src may be the interior view of dst (:1141-1146): the interior is then
// rewritten with its own values and every border pixel is read from the interior.
void copyMakeBorder(InputArray src_, OutputArray dst_, int top, int bottom, int left, int right, int borderType) {
    assert((borderType & ~BORDER_ISOLATED) == BORDER_REFLECT_101);
    const Mat src = src_.getMat();
    dst_.create(src.rows + top + bottom, src.cols + left + right, CV_8UC1);
    Mat dst = dst_.getMat();
    for (int y = 0; y < dst.rows; y++) {
        const uchar *srow = src.ptr(reflect101(y - top, src.rows));
        uchar *drow = dst.ptr(y);
        for (int x = 0; x < dst.cols; x++) drow[x] = srow[reflect101(x - left, src.cols)];
    }
}

// cv::FAST (FAST-9/16): KeyPoint(x, y, 7.f, -1, score), as modules/features2d/src/fast.cpp builds them
void FAST(InputArray image, std::vector<KeyPoint> &keypoints, int threshold, bool nonmaxSuppression) {
    const Mat img = image.getMat();
    std::vector<ygzo::FastPt> pts;
    ygzo::fast9(img.data, (int) (size_t) img.step, img.cols, img.rows, threshold, nonmaxSuppression, pts);
    keypoints.clear();
    for (const ygzo::FastPt &p : pts) keypoints.push_back(KeyPoint((float) p.x, (float) p.y, 7.f, -1, (float) p.score));
}

float fastAtan2(float y, float x) { return ygzo::fast_atan2_deg(y, x); }

// KeyPointsFilter::retainBest: keep the npoints strongest and everything tied with the weakest kept one (only ComputeKeyPointsOld,
// which nothing calls, uses it)
void KeyPointsFilter::retainBest(std::vector<KeyPoint> &keypoints, int npoints) {
    if (npoints < 0 || (int) keypoints.size() <= npoints) return;
    if (npoints == 0) { keypoints.clear(); return; }
    std::nth_element(keypoints.begin(), keypoints.begin() + npoints - 1, keypoints.end(),
                     [](const KeyPoint &a, const KeyPoint &b) { return a.response > b.response; });
    const float ambiguous = keypoints[npoints - 1].response;
    auto mid = std::partition(keypoints.begin() + npoints, keypoints.end(), [ambiguous](const KeyPoint &k) { return k.response >= ambiguous; });
    keypoints.resize(mid - keypoints.begin());
}

}  // namespace cv
