// shell_latency.cc -- what one Tracking iteration costs through the class shells (reference signatures, host cv::Mat / std::vector in and
// out, one frame at a time): ORBextractor::operator()(image), SparseImgAlign::run, ORBmatcher::SearchByProjection(cur, last).
// usage: shell_latency <dir> [iterations]   reads <dir>/a.u8, <dir>/b.u8 (752x480 u8); prints "name median_us p90_us" lines.
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "ORBextractor.h"
#include "ORBmatcher.h"
#include "SparseImageAlign.h"

namespace ygz {
float Frame::fx, Frame::fy, Frame::cx, Frame::cy, Frame::invfx, Frame::invfy, Frame::mnMinX, Frame::mnMaxX, Frame::mnMinY, Frame::mnMaxY;
}

static std::vector<unsigned char> slurp(const std::string &p) {
    FILE *f = fopen(p.c_str(), "rb");
    if (!f) { perror(p.c_str()); exit(2); }
    fseek(f, 0, SEEK_END);
    const long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    std::vector<unsigned char> b(n);
    if (fread(b.data(), 1, n, f) != (size_t) n) exit(2);
    fclose(f);
    return b;
}

template <class F> static void timeit(const char *name, int iters, F f) {
    for (int i = 0; i < 10; i++) f();
    std::vector<double> us;
    for (int i = 0; i < iters; i++) {
        const auto t0 = std::chrono::steady_clock::now();
        f();
        us.push_back(std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count());
    }
    std::sort(us.begin(), us.end());
    printf("%s %.1f %.1f\n", name, us[us.size() / 2], us[us.size() * 9 / 10]);
}

int main(int argc, char **argv) {
    using namespace ygz;
    if (argc < 2) return 2;
    const std::string dir = argv[1];
    const int iters = argc > 2 ? atoi(argv[2]) : 200;
    const int W = 752, H = 480, L = 8;
    const float depth = 4.f;
    Frame::fx = 458.654f; Frame::fy = 457.296f; Frame::cx = 367.215f; Frame::cy = 248.375f;
    Frame::mnMinX = 0; Frame::mnMinY = 0; Frame::mnMaxX = (float) W; Frame::mnMaxY = (float) H;
    std::vector<unsigned char> ia = slurp(dir + "/a.u8"), ib = slurp(dir + "/b.u8");
    ORBextractor ex(1000, 1.2f, L, 20, 7);
    cv::Mat imA(H, W, CV_8UC1, ia.data()), imB(H, W, CV_8UC1, ib.data());

    // the monocular image overload (src/ORBextractor.cc:962-1029): image in, keypoints + descriptors out, pyramid mirrored to the host
    std::vector<cv::KeyPoint> keys;
    cv::Mat desc;
    timeit("extract_image", iters, [&] { ex(imA, cv::Mat(), keys, cv::_OutputArray(desc)); });

    Frame A, B;
    A.mnId = 1;
    B.mnId = 2;
    Frame *fr[2] = {&A, &B};
    cv::Mat *im[2] = {&imA, &imB};
    for (int k = 0; k < 2; k++) {
        Frame &F = *fr[k];
        F.mImGray = *im[k];
        ex.ComputePyramid(F.mImGray);
        for (int l = 0; l < L; l++) F.mvImagePyramid.push_back(ex.mvImagePyramid[l].clone());
        F.mvScaleFactors = ex.GetScaleFactors();
        F.mvInvScaleFactors = ex.GetInverseScaleFactors();
        ex(&F, F.mvKeys, cv::_OutputArray(F.mDescriptors), ORBextractor::ORBSLAM_KEYPOINT, true);
        F.N = (int) F.mvKeys.size();
        F.mvpMapPoints.assign(F.N, nullptr);
        F.mvbOutlier.assign(F.N, false);
        F.mvuRight.assign(F.N, -1.f);
    }
    timeit("compute_pyramid", iters, [&] { ex.ComputePyramid(imA); });
    {
        Frame F;
        F.mImGray = imA;
        ex.ComputePyramid(imA);
        for (int l = 0; l < L; l++) F.mvImagePyramid.push_back(ex.mvImagePyramid[l].clone());
        timeit("frame_extract_overload", iters, [&] {
            F.mvKeys.clear();
            F.N = 0;
            ex(&F, F.mvKeys, cv::_OutputArray(F.mDescriptors), ORBextractor::ORBSLAM_KEYPOINT, true);
        });
    }
    // Frame::ExtractORB: pyramid + the Frame* overload (src/Frame.cc:332-348), as every new Frame does
    {
        Frame F;
        F.mImGray = imA;
        timeit("frame_pyramid_plus_extract", iters, [&] {
            ex.ComputePyramid(F.mImGray);
            F.mvKeys.clear();
            F.N = 0;
            F.mvImagePyramid.clear();
            for (int l = 0; l < L; l++) F.mvImagePyramid.push_back(ex.mvImagePyramid[l].clone());
            ex(&F, F.mvKeys, cv::_OutputArray(F.mDescriptors), ORBextractor::ORBSLAM_KEYPOINT, true);
        });
    }
    // the same without extract-ahead
    {
        ORBextractor::sExtractAhead = false;   // the C ABI's default: nothing is computed before it is asked for
        ORBextractor exLazy(1000, 1.2f, L, 20, 7);
        ORBextractor::sExtractAhead = true;
        Frame F;
        F.mImGray = imA;
        timeit("frame_pyramid_plus_extract_lazy", iters, [&] {
            exLazy.ComputePyramid(F.mImGray);
            F.mvKeys.clear();
            F.N = 0;
            F.mvImagePyramid.clear();
            for (int l = 0; l < L; l++) F.mvImagePyramid.push_back(exLazy.mvImagePyramid[l].clone());
            exLazy(&F, F.mvKeys, cv::_OutputArray(F.mDescriptors), ORBextractor::ORBSLAM_KEYPOINT, true);
        });
    }
    std::vector<MapPoint> mps(A.N);
    for (int i = 0; i < A.N; i++) {
        mps[i].mWorldPos[0] = (A.mvKeys[i].pt.x - Frame::cx) / Frame::fx * depth;
        mps[i].mWorldPos[1] = (A.mvKeys[i].pt.y - Frame::cy) / Frame::fy * depth;
        mps[i].mWorldPos[2] = depth;
        mps[i].mDescriptor = A.mDescriptors.row(i).clone();
        A.mvpMapPoints[i] = &mps[i];
    }
    SparseImgAlign align(L - 1, 1);
    SE3f TCR;
    size_t ret = 0;
    // every call sees a NEW current frame (fresh id: one level-0 upload + device pyramid) and the previous call's frame as reference
    timeit("sparse_img_align_run", iters, [&] { TCR = SE3f(); B.mnId += 2; ret = align.run(&A, &B, TCR); });
    // the same when the current frame's extractor still holds the image on the device -- what Tracking sees: the Frame constructor's
    // ComputePyramid (src/Frame.cc:807) ran just before -- so that the cache takes level 0 and the pyramid device to device
    ex.ComputePyramid(imB);
    B.mpORBextractorLeft = &ex;
    timeit("sparse_img_align_run_resident", iters, [&] { TCR = SE3f(); B.mnId += 2; ret = align.run(&A, &B, TCR); });
    // one direct-tracking iteration as Tracking runs it: Frame construction (ComputePyramid + clones, src/Frame.cc:807-813), then
    // TrackWithSparseAlignment's SparseImgAlign::run against the previous frame (src/Tracking.cc:2061-2105)
    timeit("direct_frame_pyramid_plus_align", iters, [&] {
        ex.ComputePyramid(imB);
        B.mvImagePyramid.clear();
        for (int l = 0; l < L; l++) B.mvImagePyramid.push_back(ex.mvImagePyramid[l].clone());
        TCR = SE3f();
        B.mnId += 2;
        ret = align.run(&A, &B, TCR);
    });
    B.mpORBextractorLeft = nullptr;
    B.mTcw = TCR;
    ORBmatcher matcher(0.9f, true);
    int nm = 0;
    timeit("search_by_projection_last", iters, [&] {
        B.mvpMapPoints.assign(B.N, nullptr);
        nm = matcher.SearchByProjection(B, A, 15.f, true);
    });
    printf("info keypoints %d aligned %zu matches %d\n", A.N, ret, nm);
    return 0;
}
