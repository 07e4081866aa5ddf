// tests/cpp/boundary_frame.cc -- TEST DRIVER of the drop-in boundary: the REFERENCE's own src/Frame.cc (compiled where it lies, real
// include/Frame.h) runs on top of the PRODUCT's class shells (orb_ygz_slam_amd/csrc/host: ygz::ORBextractor replaces include/ORBextractor.h +
// src/ORBextractor.cc, ygz::ORBmatcher's hot-path members replace those of src/ORBmatcher.cc) and libygzf.  Built by
// tests/cpp/build_boundary.sh; OpenCV / Eigen / Sophus are the stand-ins of oracle/ref_shim (not installed here), whose compute primitives
// abort in this build (tests/cpp/mini_cv_nocompute.cpp).  tests/test_gpu_boundary.py compares everything dumped here with the oracle.
//
// What runs, in the reference's own code:
//   ygz::Frame::Frame(imLeft, imRight, t, &exL, &exR, voc, K, dist, bf, thDepth)   src/Frame.cc:190-230  -> GetLevels / GetScaleFactor(s) ... of
//        the shell BEFORE any image, then Frame::ComputeImagePyramid -> shell ComputePyramid (HIP) + clones of mvImagePyramid
//   Frame::ExtractFeatures()                                                   :716-795  -> two std::threads, Frame::ExtractORB(0 / 1) -> shell
//        operator()(Frame*, keys, desc, ORBSLAM_KEYPOINT, leftEye) on two contexts concurrently (HIP); then the reference's CPU
//        Frame::ComputeStereoMatches (:509-682) reading mpORBextractor{Left,Right}->mvImagePyramid (host copies of the device pyramids) and
//        calling ORBmatcher::DescriptorDistance (shell); AssignFeaturesToGrid; ComputeBoW
//   ygz::Frame::Frame(imGray, t, &ex, voc, K, dist, bf, thDepth) + ExtractFeatures()   monocular, on a second image
//   a direct-tracked frame: N > 0 existing keys && !mbFeatureExtracted -> ExtractORB takes the DSO_KEYPOINT branch (:335-337)
//   ORBmatcher(0.9, true).SearchByProjection(cur, last, 15, mono) on those reference-built Frames (Tracking::TrackWithMotionModel's call),
//   Frame::isInFrustum (reference, CPU) feeding ORBmatcher(0.8).SearchByProjection(F, mappoints, th) (Tracking::SearchLocalPoints),
//   SparseImgAlign(nLevels-1, 1).run(&last, &cur, TCR) (Tracking.cc:207, :2087) through the reference's own class declaration,
//   and the shell's device ComputeStereoMatches(F) beside the reference's CPU one.
// Round 4: the BATCH bindings (orb_ygz_slam_amd/csrc/host/TrackingBatched.cc, FrameStereo.cc) are linked as the strong definitions of
//   Tracking::SearchLocalPoints, Tracking::SearchLocalPointsDirect and Frame::ComputeStereoMatches; the reference's own bodies of the three stay
//   callable under other names (ygz_ref_*: a renamed, all-weak second copy of the objects, tests/cpp/build_boundary.sh).  Both forms run here on the
//   same objects -- identical mvpMapPoints / MapPoint tracking fields / appended keys / mvMatchedFrom / cache / mvuRight / mvDepth are demanded in
//   this process, and the medians of both forms are printed ("latency ..." lines).
// usage: boundary_frame <dir>   reads <dir>/left.u8 right.u8 next.u8 (W x H u8, sizes in <dir>/size.i32), writes *.bin
#include <opencv2/core/core.hpp>

#define private public
#define protected public
#include "ORBmatcher.h"          // the reference's: pulls the real Frame.h (whose ORBextractor.h is the product's, same guard)
#include "SparseImageAlign.h"    // the reference's
#include "Tracking.h"            // the reference's: src/Tracking.cc itself is in the link, compiled where it lies (tests/cpp/build_boundary.sh)
#undef private
#undef protected

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <string>

static std::vector<unsigned char> slurp(const std::string &p) {
    FILE *f = fopen(p.c_str(), "rb");
    if (!f) { perror(p.c_str()); exit(2); }
    fseek(f, 0, SEEK_END);
    long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    std::vector<unsigned char> b(n);
    if (n && fread(b.data(), 1, n, f) != (size_t) n) exit(2);
    fclose(f);
    return b;
}
static void dump(const std::string &p, const void *d, size_t n) {
    FILE *f = fopen(p.c_str(), "wb");
    if (n) fwrite(d, 1, n, f);
    fclose(f);
}
static void dump_desc(const std::string &p, const cv::Mat &D, int n) {
    std::vector<unsigned char> b((size_t) n * 32);
    for (int i = 0; i < n; i++) std::memcpy(&b[(size_t) i * 32], D.ptr(i), 32);
    dump(p, b.data(), b.size());
}

namespace ygz {
// src/MapPoint.cc:359-373 for the plain-data MapPoint of oracle/ref_shim (the reference's MapPoint.cc needs Map / KeyFrame machinery)
int MapPoint::PredictScale(const float &currentDist, Frame *pF) {
    const float ratio = mfMaxDistance / currentDist;
    int nScale = (int) std::ceil(std::log(ratio) / pF->mfLogScaleFactor);
    if (nScale < 0) nScale = 0;
    else if (nScale >= pF->mnScaleLevels) nScale = pF->mnScaleLevels - 1;
    return nScale;
}
int MapPoint::PredictScale(const float &, KeyFrame *) { yr_unsupported("MapPoint::PredictScale(KeyFrame*)"); }
long unsigned int KeyFrame::nNextId = 0;
void Map::SetReferenceMapPoints(const std::vector<MapPoint *> &) {}   // "This is for visualization" (src/Tracking.cc:1603): the first line of UpdateLocalMap
}  // namespace ygz
// the reference's OWN bodies of the three members the batch bindings replace (renamed copies, tests/cpp/build_boundary.sh)
extern "C" void ygz_ref_Tracking_SearchLocalPoints(ygz::Tracking *);
extern int ygzf_host_device_frustum_min;   // orb_ygz_slam_amd/csrc/host/TrackingBatched.cc
extern "C" void ygz_ref_Tracking_SearchLocalPointsDirect(ygz::Tracking *);
extern "C" void ygz_ref_Frame_ComputeStereoMatches(ygz::Frame *);
extern "C" int ygz_ref_ORBmatcher_SearchForTriangulation(ygz::ORBmatcher *, ygz::KeyFrame *, ygz::KeyFrame *, Matrix3f &,
                                                         std::vector<std::pair<size_t, size_t> > &, bool);
#include "ORBVocabularyDevice.h"   // ygz::DeviceORBVocabulary over the reference's real DBoW2 (this build: -DYGZ_REAL_DBOW2)

int main(int argc, char **argv) {
    using namespace ygz;
    if (argc < 2) return 2;
    const std::string dir = argv[1];
    std::vector<unsigned char> sz = slurp(dir + "/size.i32");
    const int W = ((const int *) sz.data())[0], H = ((const int *) sz.data())[1], NF = ((const int *) sz.data())[2], L = ((const int *) sz.data())[3];
    std::vector<unsigned char> il = slurp(dir + "/left.u8"), ir = slurp(dir + "/right.u8"), in = slurp(dir + "/next.u8");
    auto wrap = [&](std::vector<unsigned char> &b) { cv::Mat m(H, W, CV_8UC1); std::memcpy(m.data, b.data(), (size_t) W * H); return m; };
    const cv::Mat imL = wrap(il), imR = wrap(ir), imN = wrap(in);
    Matrix3f K;
    K(0, 0) = 458.654f; K(1, 1) = 457.296f; K(0, 2) = 367.215f; K(1, 2) = 248.375f; K(2, 2) = 1.f;
    cv::Mat dist(4, 1, CV_32F);
    std::memset(dist.data, 0, 4 * sizeof(float));
    const float bf = 47.9f, thDepth = 35.f;
    // src/System.cc builds the vocabulary object and loads it with the reference's own loader; as DeviceORBVocabulary its (virtual)
    // transform runs on the GPU when Frame::ComputeBoW calls it
    DeviceORBVocabulary voc;
    if (!voc.loadFromTextFile(dir + "/voc.txt")) { fprintf(stderr, "vocabulary %s/voc.txt did not load\n", dir.c_str()); return 2; }

    // Tracking::Tracking (src/Tracking.cc:179-185): the extractors exist before the first frame arrives
    ORBextractor exL(NF, 1.2f, L, 20, 7), exR(NF, 1.2f, L, 20, 7);

    // ---- stereo frame through the reference's constructor + ExtractFeatures ------------------------------------------------------------
    Frame S(imL, imR, 0.0, &exL, &exR, &voc, K, dist, bf, thDepth);
    if ((int) S.mvScaleFactors.size() != L || (int) S.mvInvLevelSigma2.size() != L || (int) S.mvImagePyramid.size() != L) {
        fprintf(stderr, "scale tables / pyramid missing after the Frame constructor\n");
        return 3;
    }
    dump(dir + "/s_scale.bin", S.mvScaleFactors.data(), L * sizeof(float));
    for (int l = 0; l < L; l++) {
        const cv::Mat &m = S.mvImagePyramid[l];
        std::vector<unsigned char> b((size_t) m.rows * m.cols);
        for (int y = 0; y < m.rows; y++) std::memcpy(&b[(size_t) y * m.cols], m.ptr(y), (size_t) m.cols);
        dump(dir + "/s_pyr" + std::to_string(l) + ".bin", b.data(), b.size());
    }
    S.ExtractFeatures();
    dump(dir + "/s_keys.bin", S.mvKeys.data(), S.mvKeys.size() * sizeof(cv::KeyPoint));
    dump_desc(dir + "/s_desc.bin", S.mDescriptors, S.N);
    dump(dir + "/s_keysr.bin", S.mvKeysRight.data(), S.mvKeysRight.size() * sizeof(cv::KeyPoint));
    dump_desc(dir + "/s_descr.bin", S.mDescriptorsRight, (int) S.mvKeysRight.size());
    dump(dir + "/s_uright.bin", S.mvuRight.data(), S.mvuRight.size() * sizeof(float));     // the REFERENCE's CPU ComputeStereoMatches on the HIP pyramids
    dump(dir + "/s_depth.bin", S.mvDepth.data(), S.mvDepth.size() * sizeof(float));
    {   // grid built by the reference's AssignFeaturesToGrid: a few GetFeaturesInArea queries as a fingerprint
        std::vector<int> q;
        for (int k = 0; k < 12; k++) {
            const std::vector<size_t> v = S.GetFeaturesInArea(60.f + 55.f * k, 40.f + 33.f * k, 25.f, k % 3 ? -1 : 0, k % 3 ? -1 : 2);
            q.push_back((int) v.size());
            for (size_t i : v) q.push_back((int) i);
        }
        dump(dir + "/s_grid.bin", q.data(), q.size() * sizeof(int));
    }
    {   // Frame::ComputeBoW ran at the end of ExtractFeatures through the device vocabulary; beside it the CPU base class on the same descriptors
        std::vector<double> bow;
        for (auto &e : S.mBowVec) { bow.push_back((double) e.first); bow.push_back(e.second); }
        dump(dir + "/s_bow.bin", bow.data(), bow.size() * sizeof(double));
        std::vector<int> fvv;
        for (auto &e : S.mFeatVec) { fvv.push_back((int) e.first); fvv.push_back((int) e.second.size()); for (unsigned f : e.second) fvv.push_back((int) f); }
        dump(dir + "/s_featvec.bin", fvv.data(), fvv.size() * sizeof(int));
        DBoW2::BowVector cb;
        DBoW2::FeatureVector cf;
        std::vector<cv::Mat> vd = Converter::toDescriptorVector(S.mDescriptors);
        voc.ORBVocabulary::transform(vd, cb, cf, 4);          // the reference's CPU transform
        if (cb.size() != S.mBowVec.size() || cf.size() != S.mFeatVec.size()) { fprintf(stderr, "BoW sizes differ from the CPU class\n"); return 5; }
        auto a = cb.begin();
        for (auto b = S.mBowVec.begin(); b != S.mBowVec.end(); ++a, ++b)
            if (a->first != b->first || std::memcmp(&a->second, &b->second, sizeof(double))) { fprintf(stderr, "BowVector differs from the CPU class at word %u\n", a->first); return 5; }
        auto c = cf.begin();
        for (auto d = S.mFeatVec.begin(); d != S.mFeatVec.end(); ++c, ++d)
            if (c->first != d->first || c->second != d->second) { fprintf(stderr, "FeatureVector differs from the CPU class at node %u\n", c->first); return 5; }
        if (S.mBowVec.empty()) { fprintf(stderr, "empty BowVector\n"); return 5; }
    }
    {   // Frame::ComputeStereoMatches is the product's strong definition in this binary (FrameStereo.cc): what ExtractFeatures left in S came from the
        // device.  Beside it the reference's own CPU body (same Frame contents) -- and the extractor shell's entry point called directly
        Frame S2(S);
        S2.mpORBextractorLeft = &exL;
        ygz_ref_Frame_ComputeStereoMatches(&S2);
        dump(dir + "/s_uright_ref.bin", S2.mvuRight.data(), S2.mvuRight.size() * sizeof(float));
        dump(dir + "/s_depth_ref.bin", S2.mvDepth.data(), S2.mvDepth.size() * sizeof(float));
        if (S2.mvuRight.size() != S.mvuRight.size() || std::memcmp(S2.mvuRight.data(), S.mvuRight.data(), S.mvuRight.size() * sizeof(float)) ||
            std::memcmp(S2.mvDepth.data(), S.mvDepth.data(), S.mvDepth.size() * sizeof(float))) {
            fprintf(stderr, "Frame::ComputeStereoMatches: the device binding differs from the reference's CPU body\n");
            return 7;
        }
        Frame S3(S);
        S3.mpORBextractorLeft = &exL;
        exL.ComputeStereoMatches(S3);
        dump(dir + "/s_uright_dev.bin", S3.mvuRight.data(), S3.mvuRight.size() * sizeof(float));
        dump(dir + "/s_depth_dev.bin", S3.mvDepth.data(), S3.mvDepth.size() * sizeof(float));
    }

    // ---- monocular frames: last = left image, cur = next image ---------------------------------------------------------------------------
    ORBextractor ex(NF, 1.2f, L, 20, 7);
    Frame last(imL, 0.0, &ex, &voc, K, dist, bf, thDepth), cur(imN, 0.1, &ex, &voc, K, dist, bf, thDepth);
    last.ExtractFeatures();
    cur.ExtractFeatures();
    dump(dir + "/m_keys_last.bin", last.mvKeys.data(), last.mvKeys.size() * sizeof(cv::KeyPoint));
    dump(dir + "/m_keys_cur.bin", cur.mvKeys.data(), cur.mvKeys.size() * sizeof(cv::KeyPoint));
    dump_desc(dir + "/m_desc_cur.bin", cur.mDescriptors, cur.N);
    // MapPoints of `last` on the plane z = depth
    const float depth = 4.0f;
    std::vector<MapPoint> mps(last.N);
    for (int i = 0; i < last.N; i++) {
        mps[i].mWorldPos = Vector3f((last.mvKeys[i].pt.x - Frame::cx) / Frame::fx * depth, (last.mvKeys[i].pt.y - Frame::cy) / Frame::fy * depth, depth);
        mps[i].mNormal = Vector3f(0, 0, 1);
        mps[i].mDescriptor = last.mDescriptors.row(i).clone();
        mps[i].nObs = 1;
        mps[i].mfMaxDistance = depth * last.mvScaleFactors[last.mvKeys[i].octave];
        mps[i].maxDistInv = 1.2f * mps[i].mfMaxDistance;
        mps[i].minDistInv = 0.8f * mps[i].mfMaxDistance / last.mvScaleFactors[L - 1];
        last.mvpMapPoints[i] = &mps[i];
    }
    last.SetPose(SE3f());
    // TrackWithSparseAlignment (src/Tracking.cc:2061-2105)
    SparseImgAlign align(L - 1, 1);
    SE3f TCR;
    cur.SetPose(SE3f());
    const size_t ret = align.run(&last, &cur, TCR);
    float t7[8];
    { const Eigen::Quaternionf q = TCR.unit_quaternion(); t7[0] = q.x(); t7[1] = q.y(); t7[2] = q.z(); t7[3] = q.w(); }
    for (int i = 0; i < 3; i++) t7[4 + i] = TCR.translation()[i];
    t7[7] = (float) ret;
    dump(dir + "/m_tcr.bin", t7, sizeof t7);
    { const Matrix<float, 6, 6> I = align.getFisherInformation(); float f36[36]; for (int r = 0; r < 6; r++) for (int c = 0; c < 6; c++) f36[6 * r + c] = I(r, c); dump(dir + "/m_fisher.bin", f36, sizeof f36); }
    cur.SetPose(TCR * last.mTcw);                                        // :2095
    {
        float Rt[12];
        for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) Rt[3 * r + c] = cur.mRcw(r, c);
        for (int r = 0; r < 3; r++) Rt[9 + r] = cur.mtcw[r];
        dump(dir + "/m_pose.bin", Rt, sizeof Rt);
    }
    // TrackWithMotionModel (:1072-1093)
    ORBmatcher matcher(0.9, true);
    const int nm = matcher.SearchByProjection(cur, last, 15, true);
    std::vector<int> assigned(cur.N, -1);
    for (int i = 0; i < cur.N; i++)
        if (cur.mvpMapPoints[i]) assigned[i] = (int) (cur.mvpMapPoints[i] - mps.data());
    dump(dir + "/m_match.bin", assigned.data(), assigned.size() * sizeof(int));
    dump(dir + "/m_nmatch.bin", &nm, sizeof nm);
    // SearchLocalPointsDirect (:2174-2326): FindDirectProjection(KF, &cur, mp, px, level) once per candidate, through the reference's class
    {
        KeyFrame KF;
        KF.mnId = 3;
        KF.mvKeys = last.mvKeys;
        KF.mvImagePyramid = last.mvImagePyramid;
        KF.mPose = last.mTcw;
        const int nd = std::min(last.N, 120);
        std::vector<float> out((size_t) nd * 4);
        for (int i = 0; i < nd; i++) {
            mps[i].mObservations[&KF] = (size_t) i;
            Vector2f px(last.mvKeys[i].pt.x + ((i % 5) - 2) * 0.75f, last.mvKeys[i].pt.y + ((i % 3) - 1) * 0.5f);
            int level = -1;
            const bool ok = matcher.FindDirectProjection(&KF, &cur, &mps[i], px, level);
            out[4 * i] = px[0]; out[4 * i + 1] = px[1]; out[4 * i + 2] = (float) level; out[4 * i + 3] = ok ? 1.f : 0.f;
        }
        dump(dir + "/m_direct.bin", out.data(), out.size() * sizeof(float));
        const Eigen::Quaternionf q = cur.mTcw.unit_quaternion();
        const float c7[7] = {q.x(), q.y(), q.z(), q.w(), cur.mTcw.translation()[0], cur.mTcw.translation()[1], cur.mTcw.translation()[2]};
        dump(dir + "/m_pose7.bin", c7, sizeof c7);
    }
    // SearchLocalPoints (:1544-1593): the reference's isInFrustum marks the points, the shell searches
    {
        Frame cur2(cur);
        cur2.mvpMapPoints.assign(cur2.N, (MapPoint *) nullptr);
        cur2.SetPose(cur.mTcw);
        std::vector<MapPoint *> local;
        std::vector<float> frustum;
        for (int i = 0; i < last.N; i++) {
            const bool in = cur2.isInFrustum(&mps[i], 0.5f);
            frustum.push_back(in ? 1.f : 0.f);
            frustum.push_back(mps[i].mTrackProjX); frustum.push_back(mps[i].mTrackProjY); frustum.push_back((float) mps[i].mnTrackScaleLevel);
            frustum.push_back(mps[i].mTrackViewCos);
            local.push_back(&mps[i]);
        }
        dump(dir + "/m_frustum.bin", frustum.data(), frustum.size() * sizeof(float));
        ORBmatcher m2(0.8, true);
        const int nm2 = m2.SearchByProjection(cur2, local, 3, false);
        std::vector<int> a2(cur2.N, -1);
        for (int i = 0; i < cur2.N; i++)
            if (cur2.mvpMapPoints[i]) a2[i] = (int) (cur2.mvpMapPoints[i] - mps.data());
        dump(dir + "/m_match2.bin", a2.data(), a2.size() * sizeof(int));
        dump(dir + "/m_nmatch2.bin", &nm2, sizeof nm2);
    }
    // LocalMapping::CreateNewMapPoints (src/LocalMapping.cc:1038-1042): SearchForTriangulation(KF1, KF2, F12, pairs, false) between the two
    // monocular frames as KeyFrames, with the FeatureVectors the device vocabulary gave them (REAL DBoW2 node ids) and the pose the aligner
    // found.  ORBmatcher::SearchForTriangulation is the product's strong definition in this binary; the reference's own body runs beside it.
    {
        KeyFrame K1, K2;
        Frame *src[2] = {&last, &cur};
        KeyFrame *kf[2] = {&K1, &K2};
        for (int s = 0; s < 2; s++) {
            KeyFrame &Kf = *kf[s];
            const Frame &F = *src[s];
            Kf.N = F.N;
            Kf.mvKeys = F.mvKeys;
            Kf.mvuRight.assign(F.N, -1.f);
            for (int i = 0; i < F.N; i += 3) Kf.mvuRight[i] = F.mvKeys[i].pt.x - 4.f;     // every third keypoint has a stereo match
            Kf.mDescriptors = F.mDescriptors;
            Kf.mFeatVec = F.mFeatVec;
            Kf.mvpMapPoints.assign(F.N, (MapPoint *) nullptr);
            for (int i = s; i < F.N; i += 5) Kf.mvpMapPoints[i] = &mps[0];                  // slots that already hold a MapPoint
            Kf.fx = Frame::fx; Kf.fy = Frame::fy; Kf.cx = Frame::cx; Kf.cy = Frame::cy;
            Kf.mvScaleFactors = F.mvScaleFactors;
            Kf.mvLevelSigma2 = F.mvLevelSigma2;
            Kf.mnScaleLevels = L;
            Kf.mHasPose = true;
            for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) Kf.mRcw(r, c) = F.mRcw(r, c); Kf.mtcw[r] = F.mtcw[r]; }
            Kf.mOw = F.mOw;
        }
        if (K1.mFeatVec.empty() || K2.mFeatVec.empty()) { fprintf(stderr, "SearchForTriangulation: empty FeatureVector\n"); return 7; }
        // ComputeF12 (src/LocalMapping.cc:1329-1344) in double, rounded once: K1^-T [t12]x R12 K2^-1
        Matrix3f F12;
        {
            double R1[9], R2[9], t1[3], t2[3], R12[9], t12[3], Ki[9] = {1.0 / Frame::fx, 0, -Frame::cx / (double) Frame::fx, 0, 1.0 / Frame::fy, -Frame::cy / (double) Frame::fy, 0, 0, 1};
            for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) { R1[3 * r + c] = K1.mRcw(r, c); R2[3 * r + c] = K2.mRcw(r, c); } t1[r] = K1.mtcw[r]; t2[r] = K2.mtcw[r]; }
            for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) { double a = 0; for (int k = 0; k < 3; k++) a += R1[3 * r + k] * R2[3 * c + k]; R12[3 * r + c] = a; }
            for (int r = 0; r < 3; r++) { double a = 0; for (int k = 0; k < 3; k++) a += R12[3 * r + k] * t2[k]; t12[r] = -a + t1[r]; }
            const double tx[9] = {0, -t12[2], t12[1], t12[2], 0, -t12[0], -t12[1], t12[0], 0};
            double A[9], B[9], C[9];
            for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) { double a = 0; for (int k = 0; k < 3; k++) a += Ki[3 * k + r] * tx[3 * k + c]; A[3 * r + c] = a; }   // K^-T [t]x
            for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) { double a = 0; for (int k = 0; k < 3; k++) a += A[3 * r + k] * R12[3 * k + c]; B[3 * r + c] = a; }
            for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) { double a = 0; for (int k = 0; k < 3; k++) a += B[3 * r + k] * Ki[3 * k + c]; C[3 * r + c] = a; }
            for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) F12(r, c) = (float) C[3 * r + c];
        }
        for (int only = 0; only < 2; only++) {
            std::vector<std::pair<size_t, size_t> > got, want;
            ORBmatcher mt(0.6f, true);
            const int ng = mt.SearchForTriangulation(&K1, &K2, F12, got, only != 0);
            const int nw = ygz_ref_ORBmatcher_SearchForTriangulation(&mt, &K1, &K2, F12, want, only != 0);
            if (ng != nw || got != want) { fprintf(stderr, "SearchForTriangulation(onlyStereo=%d): device %d pairs, the reference's own body %d\n", only, ng, nw); return 7; }
            if (only == 0 && ng < 50) { fprintf(stderr, "SearchForTriangulation: only %d pairs between two views of one plane\n", ng); return 7; }
            printf("info search_for_triangulation only_stereo %d pairs %d nodes %zu / %zu\n", only, ng, K1.mFeatVec.size(), K2.mFeatVec.size());
        }
        {
            auto med = [&](bool device) {
                std::vector<double> us;
                for (int it = 0; it < 40; it++) {
                    std::vector<std::pair<size_t, size_t> > pr;
                    ORBmatcher mt(0.6f, true);
                    const auto t0 = std::chrono::steady_clock::now();
                    if (device) mt.SearchForTriangulation(&K1, &K2, F12, pr, false); else ygz_ref_ORBmatcher_SearchForTriangulation(&mt, &K1, &K2, F12, pr, false);
                    us.push_back(std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count());
                }
                std::sort(us.begin(), us.end());
                return us[us.size() / 2];
            };
            const double dev = med(true), ref = med(false);
            printf("latency search_for_triangulation reference_us %.1f device_us %.1f keys %d / %d\n", ref, dev, K1.N, K2.N);
        }
    }
    // ---- the reference's own src/Tracking.cc (compiled unchanged, linked against the product's strong ORBmatcher / SparseImgAlign / ORBextractor
    // symbols) drives the hot path: Tracking::Tracking builds its extractors and its aligner from the settings file (:83-213),
    // TrackWithSparseAlignment (:2061-2105) calls SparseImgAlign::run, SearchLocalPoints (:1544-1593) calls Frame::isInFrustum and
    // ORBmatcher::SearchByProjection(F, MapPoints).  Every class outside the hot path is a declaration whose members abort when reached.
    {
        System sys;
        Map map;
        KeyFrameDatabase kfdb;
        FrameDrawer fdraw;
        MapDrawer mdraw;
        ConfigParam params;
        Tracking trk(&sys, &voc, &fdraw, &mdraw, &map, &kfdb, dir + "/settings.yaml", System::MONOCULAR, &params);
        if (!trk.mpORBextractorLeft || !trk.mpIniORBextractor || trk.mpORBextractorLeft->GetLevels() != L || !trk.mpAlign) {
            fprintf(stderr, "Tracking::Tracking did not build its extractors / aligner from the settings\n");
            return 6;
        }
        KeyFrame refKF;
        refKF.mPose = SE3f();
        for (auto &mp : mps) { mp.mnLastFrameSeen = 0; mp.mbTrackInView = false; }
        trk.mLastFrame = Frame(last);                       // the reference's copy constructor: deep-cloned pyramid, same id
        trk.mLastFrame.mpReferenceKF = &refKF;
        trk.mlRelativeFramePoses.push_back(SE3f());          // Tlr of UpdateLastFrame (:981-986)
        trk.mVelocity = SE3f();
        Frame fresh(imN, 0.1, &ex, &voc, K, dist, bf, thDepth);
        fresh.ExtractFeatures();
        trk.mCurrentFrame = Frame(fresh);
        const bool okA = trk.TrackWithSparseAlignment(false);
        float p7[8];
        { const Eigen::Quaternionf q = trk.mCurrentFrame.mTcw.unit_quaternion(); p7[0] = q.x(); p7[1] = q.y(); p7[2] = q.z(); p7[3] = q.w(); }
        for (int i = 0; i < 3; i++) p7[4 + i] = trk.mCurrentFrame.mTcw.translation()[i];
        p7[7] = okA ? 1.f : 0.f;
        dump(dir + "/t_pose7.bin", p7, sizeof p7);
        // SearchLocalPoints on the aligned frame: nothing matched yet, the local map = the last frame's points
        trk.mCurrentFrame.mvpMapPoints.assign(trk.mCurrentFrame.N, (MapPoint *) nullptr);
        trk.mvpLocalMapPoints.clear();
        for (auto &mp : mps) trk.mvpLocalMapPoints.push_back(&mp);
        trk.mnLastRelocFrameId = 0;
        trk.mbDirectFailed = false;
        // ---- the reference's own body of SearchLocalPoints (per point isInFrustum on the CPU, then the per-call matcher member) on this state ...
        struct MpState { bool inView; float x, y, xr, vc; int lvl, visible; unsigned long seen; };
        auto snapshot = [&]() {
            std::vector<MpState> v;
            for (auto &mp : mps) v.push_back(MpState{mp.mbTrackInView, mp.mTrackProjX, mp.mTrackProjY, mp.mTrackProjXR, mp.mTrackViewCos, mp.mnTrackScaleLevel, mp.mnVisible, mp.mnLastFrameSeen});
            return v;
        };
        auto same_state = [](const std::vector<MpState> &a, const std::vector<MpState> &b) {
            if (a.size() != b.size()) return false;
            for (size_t i = 0; i < a.size(); i++) {
                if (a[i].inView != b[i].inView || a[i].visible != b[i].visible || a[i].seen != b[i].seen) return false;
                if (a[i].inView && (std::memcmp(&a[i].x, &b[i].x, 4 * sizeof(float)) || a[i].lvl != b[i].lvl)) return false;
            }
            return true;
        };
        const std::vector<MpState> before = snapshot();
        ygz_ref_Tracking_SearchLocalPoints(&trk);
        const std::vector<MapPoint *> refAssigned = trk.mCurrentFrame.mvpMapPoints;
        const std::vector<MpState> refState = snapshot();
        // ... and, from the same starting state, the batch binding (TrackingBatched.cc: one ygzf_search_local_points call)
        trk.mCurrentFrame.mvpMapPoints.assign(trk.mCurrentFrame.N, (MapPoint *) nullptr);
        for (size_t i = 0; i < mps.size(); i++) {
            mps[i].mbTrackInView = before[i].inView; mps[i].mTrackProjX = before[i].x; mps[i].mTrackProjY = before[i].y; mps[i].mTrackProjXR = before[i].xr;
            mps[i].mTrackViewCos = before[i].vc; mps[i].mnTrackScaleLevel = before[i].lvl; mps[i].mnVisible = before[i].visible; mps[i].mnLastFrameSeen = before[i].seen;
        }
        trk.SearchLocalPoints();
        if (trk.mCurrentFrame.mvpMapPoints != refAssigned || !same_state(snapshot(), refState)) {
            fprintf(stderr, "Tracking::SearchLocalPoints: the batch binding differs from the reference's per-call body\n");
            return 7;
        }
        {
            auto restore = [&]() {
                trk.mCurrentFrame.mvpMapPoints.assign(trk.mCurrentFrame.N, (MapPoint *) nullptr);
                for (size_t i = 0; i < mps.size(); i++) { mps[i].mnVisible = before[i].visible; mps[i].mnLastFrameSeen = before[i].seen; mps[i].mbTrackInView = before[i].inView; }
            };
            auto med = [&](bool batch) {
                std::vector<double> us;
                for (int it = 0; it < 9; it++) {
                    restore();
                    const auto t0 = std::chrono::steady_clock::now();
                    if (batch) trk.SearchLocalPoints(); else ygz_ref_Tracking_SearchLocalPoints(&trk);
                    us.push_back(std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count());
                }
                std::sort(us.begin(), us.end());
                return us[4];
            };
            const double a = med(false), b = med(true);
            const int keepMin = ygzf_host_device_frustum_min;
            ygzf_host_device_frustum_min = 0;                 // the fused device frustum + matcher call whatever the size of the local map
            restore();
            trk.SearchLocalPoints();
            if (trk.mCurrentFrame.mvpMapPoints != refAssigned || !same_state(snapshot(), refState)) {
                fprintf(stderr, "Tracking::SearchLocalPoints: the fused device form differs from the reference's per-call body\n");
                return 7;
            }
            const double f = med(true);
            ygzf_host_device_frustum_min = keepMin;
            printf("latency search_local_points percall_us %.1f batch_us %.1f local_points %zu fused_device_us %.1f\n", a, b, mps.size(), f);
            restore();
            trk.SearchLocalPoints();      // leave the state the dumps below expect
        }
        {
            // ---- a local map ABOVE the binding's threshold (ygzf_host_device_frustum_min, 1500 by default): the last frame's points plus a second
            // point beside every one of them (~2000 candidates, what a well-mapped scene offers).  Nothing is forced here: the default
            // binding takes the fused device frustum + matcher call by itself, and must leave every MapPoint and every slot of the frame as the
            // reference's own body (per-point Frame::isInFrustum, per-call matcher) does.
            std::vector<MapPoint> extra;
            extra.reserve(mps.size());
            for (size_t i = 0; i < mps.size(); i++) {
                // a second point per keypoint of the last frame: ~35 px to the side at the same depth, carrying ANOTHER point's descriptor -- most
                // of them find keypoints in their window and no acceptable match, as the points of a local map that left the view's appearance do
                MapPoint q = mps[i];
                q.mWorldPos = mps[i].mWorldPos + Vector3f((i & 1) ? 0.3f : -0.3f, (i & 2) ? 0.2f : -0.2f, 0.f);
                q.mDescriptor = mps[(i * 7 + 13) % mps.size()].mDescriptor.clone();
                q.mnVisible = 1; q.mnLastFrameSeen = 0; q.mbTrackInView = false;
                extra.push_back(q);
            }
            std::vector<MapPoint *> big;
            for (auto &mp : mps) big.push_back(&mp);
            for (auto &mp : extra) big.push_back(&mp);
            struct St { bool inView; float x, y, xr, vc; int lvl, visible; unsigned long seen; };
            auto snap = [&]() { std::vector<St> v; for (MapPoint *m : big) v.push_back(St{m->mbTrackInView, m->mTrackProjX, m->mTrackProjY, m->mTrackProjXR, m->mTrackViewCos, m->mnTrackScaleLevel, m->mnVisible, m->mnLastFrameSeen}); return v; };
            auto reset = [&]() {
                trk.mCurrentFrame.mvpMapPoints.assign(trk.mCurrentFrame.N, (MapPoint *) nullptr);
                for (MapPoint *m : big) { m->mnVisible = 1; m->mnLastFrameSeen = 0; m->mbTrackInView = false; }
            };
            auto same = [](const std::vector<St> &a, const std::vector<St> &b2) {
                for (size_t i = 0; i < a.size(); i++) {
                    if (a[i].inView != b2[i].inView || a[i].visible != b2[i].visible || a[i].seen != b2[i].seen) return false;
                    if (a[i].inView && (std::memcmp(&a[i].x, &b2[i].x, 4 * sizeof(float)) || a[i].lvl != b2[i].lvl)) return false;
                }
                return true;
            };
            const std::vector<MapPoint *> keepLocal = trk.mvpLocalMapPoints;
            trk.mvpLocalMapPoints = big;
            if ((int) big.size() < ygzf_host_device_frustum_min) { fprintf(stderr, "large local map: %zu points do not reach the threshold %d\n", big.size(), ygzf_host_device_frustum_min); return 7; }
            reset();
            ygz_ref_Tracking_SearchLocalPoints(&trk);
            const std::vector<MapPoint *> refA = trk.mCurrentFrame.mvpMapPoints;
            const std::vector<St> refS = snap();
            reset();
            trk.SearchLocalPoints();                                            // default threshold: the fused device call
            if (trk.mCurrentFrame.mvpMapPoints != refA || !same(snap(), refS)) {
                fprintf(stderr, "Tracking::SearchLocalPoints (large local map, default threshold): the binding differs from the reference's body\n");
                return 7;
            }
            int matched = 0;
            for (MapPoint *m : refA) matched += m != nullptr;
            auto med2 = [&](bool batch) {
                std::vector<double> us;
                for (int it = 0; it < 9; it++) {
                    reset();
                    const auto t0 = std::chrono::steady_clock::now();
                    if (batch) trk.SearchLocalPoints(); else ygz_ref_Tracking_SearchLocalPoints(&trk);
                    us.push_back(std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count());
                }
                std::sort(us.begin(), us.end());
                return us[4];
            };
            const double a2 = med2(false), b2 = med2(true);
            printf("latency search_local_points_large percall_us %.1f binding_default_us %.1f local_points %zu matched %d\n", a2, b2, big.size(), matched);
            // back to the state the dumps below expect
            trk.mvpLocalMapPoints = keepLocal;
            trk.mCurrentFrame.mvpMapPoints.assign(trk.mCurrentFrame.N, (MapPoint *) nullptr);
            for (size_t i = 0; i < mps.size(); i++) { mps[i].mnVisible = before[i].visible; mps[i].mnLastFrameSeen = before[i].seen; mps[i].mbTrackInView = before[i].inView; }
            trk.SearchLocalPoints();
        }
        std::vector<int> a3(trk.mCurrentFrame.N, -1);
        int visible = 0;
        for (int i = 0; i < trk.mCurrentFrame.N; i++)
            if (trk.mCurrentFrame.mvpMapPoints[i]) a3[i] = (int) (trk.mCurrentFrame.mvpMapPoints[i] - mps.data());
        for (auto &mp : mps) visible += mp.mnVisible - 1;
        dump(dir + "/t_match.bin", a3.data(), a3.size() * sizeof(int));
        const int info[3] = {(int) trk.mCurrentFrame.mnId, visible, trk.mCurrentFrame.N};
        dump(dir + "/t_info.bin", info, sizeof info);
        dump(dir + "/t_keys.bin", trk.mCurrentFrame.mvKeys.data(), trk.mCurrentFrame.mvKeys.size() * sizeof(cv::KeyPoint));
        dump_desc(dir + "/t_desc.bin", trk.mCurrentFrame.mDescriptors, trk.mCurrentFrame.N);

        // ---- Tracking::SearchLocalPointsDirect (:2174-2326): the reference's body (FindDirectProjection once per candidate through the per-call member)
        // against the batch binding (one ygzf_is_in_frustum_batch + one ygzf_find_direct_projection_batch per half), both halves of the function:
        //   phase 0  empty cache: UpdateLocalMap, then every local point;   phase 1  the cache phase 0 filled, enough hits to return early
        KeyFrame KF1, KF0;                      // KF1 observes every point (the last frame as a KeyFrame); KF0 is mpLastKeyFrame (SelectNearestKeyframe skips it)
        KF1.mnId = 3; KF0.mnId = 1;
        KF1.mvKeys = last.mvKeys;
        KF1.mvImagePyramid = last.mvImagePyramid;
        KF1.mPose = last.mTcw;
        KF0.mvImagePyramid = last.mvImagePyramid;
        for (size_t i = 0; i < mps.size(); i++) {
            mps[i].mObservations.clear();
            mps[i].mObservations[&KF1] = i;
            KF1.mvpMapPoints.push_back(&mps[i]);
        }
        const Frame aligned(trk.mCurrentFrame);           // pose from TrackWithSparseAlignment, keys of the extraction
        struct DirectOut { std::vector<cv::KeyPoint> keys; std::vector<MapPoint *> mp; std::vector<int> from; std::set<MapPoint *> cache; int N; };
        auto run_direct = [&](bool batch, int phase, const std::set<MapPoint *> &cache0) {
            trk.mCurrentFrame = Frame(aligned);
            trk.mCurrentFrame.mvpMapPoints.assign(trk.mCurrentFrame.N, (MapPoint *) nullptr);   // nothing matched yet: UpdateLocalKeyFrames keeps the list below
            trk.mCurrentFrame.mvMatchedFrom.clear();
            trk.mvpLocalKeyFrames.assign(1, &KF1);
            trk.mpLastKeyFrame = &KF0;
            trk.mvpLocalMapPoints.clear();
            trk.mvpDirectMapPointsCache = cache0;
            trk.mnCacheHitTh = phase == 0 ? 150 : 10;
            for (auto &mp : mps) { mp.mnTrackReferenceForFrame = 0; mp.mbTrackInView = false; }
            if (batch) trk.SearchLocalPointsDirect();
            else ygz_ref_Tracking_SearchLocalPointsDirect(&trk);
            DirectOut o;
            const int n0 = aligned.N;
            o.keys.assign(trk.mCurrentFrame.mvKeys.begin() + n0, trk.mCurrentFrame.mvKeys.end());
            o.mp.assign(trk.mCurrentFrame.mvpMapPoints.begin() + n0, trk.mCurrentFrame.mvpMapPoints.end());
            o.from = trk.mCurrentFrame.mvMatchedFrom;
            o.cache = trk.mvpDirectMapPointsCache;
            o.N = trk.mCurrentFrame.N;
            return o;
        };
        auto same_direct = [](const DirectOut &a, const DirectOut &b) {
            return a.N == b.N && a.mp == b.mp && a.from == b.from && a.cache == b.cache && a.keys.size() == b.keys.size() &&
                   (a.keys.empty() || !std::memcmp(a.keys.data(), b.keys.data(), a.keys.size() * sizeof(cv::KeyPoint)));
        };
        const DirectOut r0 = run_direct(false, 0, std::set<MapPoint *>()), b0 = run_direct(true, 0, std::set<MapPoint *>());
        const DirectOut r1 = run_direct(false, 1, r0.cache), b1 = run_direct(true, 1, r0.cache);
        if (!same_direct(r0, b0) || !same_direct(r1, b1)) {
            fprintf(stderr, "Tracking::SearchLocalPointsDirect: the batch binding differs from the reference's per-call body (phase 0: %zu vs %zu keys, phase 1: %zu vs %zu)\n",
                    r0.keys.size(), b0.keys.size(), r1.keys.size(), b1.keys.size());
            return 7;
        }
        auto dump_direct = [&](const std::string &tag, const DirectOut &o) {
            dump(dir + "/" + tag + "_keys.bin", o.keys.data(), o.keys.size() * sizeof(cv::KeyPoint));
            std::vector<int> idx;
            for (MapPoint *m : o.mp) idx.push_back((int) (m - mps.data()));
            dump(dir + "/" + tag + "_mp.bin", idx.data(), idx.size() * sizeof(int));
            dump(dir + "/" + tag + "_from.bin", o.from.data(), o.from.size() * sizeof(int));
            std::vector<int> c;
            for (MapPoint *m : o.cache) c.push_back((int) (m - mps.data()));
            dump(dir + "/" + tag + "_cache.bin", c.data(), c.size() * sizeof(int));
        };
        dump_direct("x0", b0);
        dump_direct("x1", b1);
        {   // what the two forms cost per frame (median of 7 runs each; the images are in the HBM cache by now)
            auto med = [&](bool batch, int phase, const std::set<MapPoint *> &c0) {
                std::vector<double> us;
                for (int it = 0; it < 7; it++) {
                    const auto t0 = std::chrono::steady_clock::now();
                    run_direct(batch, phase, c0);
                    us.push_back(std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count());
                }
                std::sort(us.begin(), us.end());
                return us[3];
            };
            printf("latency search_local_points_direct_local_map percall_us %.1f batch_us %.1f candidates %zu\n", med(false, 0, std::set<MapPoint *>()),
                   med(true, 0, std::set<MapPoint *>()), mps.size());
            printf("latency search_local_points_direct_cache percall_us %.1f batch_us %.1f cached %zu\n", med(false, 1, r0.cache), med(true, 1, r0.cache), r0.cache.size());
        }
        {
            const Eigen::Quaternionf q = aligned.mTcw.unit_quaternion();
            const float c7[7] = {q.x(), q.y(), q.z(), q.w(), aligned.mTcw.translation()[0], aligned.mTcw.translation()[1], aligned.mTcw.translation()[2]};
            dump(dir + "/x_pose7.bin", c7, sizeof c7);
            float Rt[12];
            for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) Rt[3 * r + c] = aligned.mRcw(r, c);
            for (int r = 0; r < 3; r++) Rt[9 + r] = aligned.mtcw[r];
            dump(dir + "/x_pose.bin", Rt, sizeof Rt);
        }
    }
    // ---- direct-tracked frame: keys carried over from the last frame, no extraction yet -> ExtractORB takes DSO_KEYPOINT (src/Frame.cc:335-337)
    {
        Frame D(imN, 0.2, &ex, &voc, K, dist, bf, thDepth);
        const int keep = std::min(cur.N, 120);
        D.mvKeys.assign(cur.mvKeys.begin(), cur.mvKeys.begin() + keep);
        D.N = keep;
        D.mvpMapPoints.assign(keep, (MapPoint *) nullptr);
        D.mvbOutlier.assign(keep, false);
        D.ExtractFeatures();
        dump(dir + "/d_keys.bin", D.mvKeys.data(), D.mvKeys.size() * sizeof(cv::KeyPoint));
        dump_desc(dir + "/d_desc.bin", D.mDescriptors, D.N);
    }
    printf("boundary ok: stereo %d / %d keys, mono %d -> %d keys, align ret %zu, %d matches\n", S.N, (int) S.mvKeysRight.size(), last.N, cur.N, ret, nm);
    return 0;
}
