// test_shells.cc -- drives the three class shells (reference signatures) the way Tracking.cc does and dumps their
// outputs as raw binaries for tests/test_gpu_shells.py to compare with the oracle.
// usage: test_shells <dir>   reads <dir>/a.u8, <dir>/b.u8 (752x480 u8), <dir>/depth.f32 (plane depth) and <dir>/tri.f32 (two-view geometry).
#include <cmath>
#include <cstdio>
#include <set>
#include <cstdlib>
#include <string>
#include <thread>
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
    long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    std::vector<unsigned char> b(n);
    if (fread(b.data(), 1, n, f) != (size_t) n) exit(2);
    fclose(f);
    return b;
}
static void dump(const std::string &p, const void *d, size_t n) {
    FILE *f = fopen(p.c_str(), "wb");
    if (n) fwrite(d, 1, n, f);
    fclose(f);
}

int main(int argc, char **argv) {
    using namespace ygz;
    if (argc < 2) return 2;
    const std::string dir = argv[1];
    const int W = 752, H = 480, L = 8;
    Frame::fx = 458.654f; Frame::fy = 457.296f; Frame::cx = 367.215f; Frame::cy = 248.375f;
    Frame::mnMinX = 0; Frame::mnMinY = 0; Frame::mnMaxX = (float) W; Frame::mnMaxY = (float) H;
    std::vector<unsigned char> ia = slurp(dir + "/a.u8"), ib = slurp(dir + "/b.u8"), wd = slurp(dir + "/depth.f32");
    const float depth = *(const float *) wd.data();
    ORBextractor ex(600, 1.2f, L, 20, 7);
    {   // Frame's constructors read the tables before any image has been processed (src/Frame.cc:119-125): they must exist right after
        // construction, with no device work behind them
        std::vector<float> t;
        for (const std::vector<float> &v : {ex.GetScaleFactors(), ex.GetInverseScaleFactors(), ex.GetScaleSigmaSquares(), ex.GetInverseScaleSigmaSquares()}) {
            if ((int) v.size() != L) { fprintf(stderr, "scale table of %zu entries right after construction\n", v.size()); return 4; }
            t.insert(t.end(), v.begin(), v.end());
        }
        if (ex.GetLevels() != L || ex.GetScaleFactor() != 1.2f) return 4;
        dump(dir + "/tables.bin", t.data(), t.size() * sizeof(float));
    }
    Frame A, B;
    A.mnId = 1;                          // Frame::nNextId++ in the reference's constructors: ids are unique per image
    B.mnId = 2;
    Frame *fr[2] = {&A, &B};
    std::vector<unsigned char> *im[2] = {&ia, &ib};
    for (int k = 0; k < 2; k++) {
        Frame &F = *fr[k];
        F.mImGray = cv::Mat(H, W, CV_8UC1, im[k]->data());
        ex.ComputePyramid(F.mImGray);                                  // Frame ctor -> ComputeImagePyramid (src/Frame.cc:807-813)
        F.mvImagePyramid.clear();
        for (int l = 0; l < L; l++) F.mvImagePyramid.push_back(ex.mvImagePyramid[l].clone());
        F.mvScaleFactors = ex.GetScaleFactors();
        F.mvInvScaleFactors = ex.GetInverseScaleFactors();
        ex(&F, F.mvKeys, cv::_OutputArray(F.mDescriptors), ORBextractor::ORBSLAM_KEYPOINT, true);   // Frame::ExtractORB
        F.N = (int) F.mvKeys.size();
        F.mvpMapPoints.assign(F.N, nullptr);
        F.mvbOutlier.assign(F.N, false);
        F.mvuRight.assign(F.N, -1.f);
    }
    dump(dir + "/a_kps.bin", A.mvKeys.data(), A.mvKeys.size() * sizeof(cv::KeyPoint));
    dump(dir + "/a_desc.bin", A.mDescriptors.ptr(0), (size_t) A.N * 32);
    dump(dir + "/b_kps.bin", B.mvKeys.data(), B.mvKeys.size() * sizeof(cv::KeyPoint));
    dump(dir + "/b_desc.bin", B.mDescriptors.ptr(0), (size_t) B.N * 32);
    // MapPoints of A: plane at `depth`, descriptor = the keypoint's own
    std::vector<MapPoint> mps(A.N);
    for (int i = 0; i < A.N; i++) {
        mps[i].mWorldPos[0] = (A.mvKeys[i].pt.x - Frame::cx) / Frame::fx * depth;
        mps[i].mWorldPos[1] = (A.mvKeys[i].pt.y - Frame::cy) / Frame::fy * depth;
        mps[i].mWorldPos[2] = depth;
        mps[i].mDescriptor = A.mDescriptors.row(i).clone();
        A.mvpMapPoints[i] = &mps[i];
    }
    // TrackWithSparseAlignment: SparseImgAlign(nLevels-1, 1).run(&last, &cur, TCR)   (src/Tracking.cc:207, :2087)
    SparseImgAlign align(L - 1, 1);
    SE3f TCR;
    const size_t ret = align.run(&A, &B, TCR);
    float t7[8];
    ygz_compat::se3_to7(TCR, t7);
    t7[7] = (float) ret;
    dump(dir + "/tcr.bin", t7, sizeof t7);
    // The image cache behind run(): (i) a COPY of B (deep-cloned pyramid at another address, same mnId: mLastFrame = Frame(mCurrentFrame)) must give
    // the same answer (it hits B's slot); (ii) a frame that reuses B's id AND B's buffer with other pixels (Tracking::Reset restarts the id
    // counters, malloc reuses addresses) must be aligned against its OWN pixels, not against the stale slot
    {
        Frame Bc = B;
        for (auto &m : Bc.mvImagePyramid) m = m.clone();
        SE3f T2;
        const size_t r2 = align.run(&A, &Bc, T2);
        float u7[8];
        ygz_compat::se3_to7(T2, u7);
        u7[7] = (float) r2;
        dump(dir + "/tcr_copy.bin", u7, sizeof u7);
        std::vector<cv::Mat> keep = B.mvImagePyramid;                 // B's pixels, to put back afterwards
        std::vector<cv::Mat> saved;
        for (auto &m : keep) saved.push_back(m.clone());
        for (size_t l = 0; l < B.mvImagePyramid.size(); l++)          // same buffers, A's pixels: the alignment A -> "B" is now the identity
            for (int y = 0; y < B.mvImagePyramid[l].rows; y++) std::memcpy(B.mvImagePyramid[l].ptr(y), A.mvImagePyramid[l].ptr(y), (size_t) B.mvImagePyramid[l].cols);
        SE3f T3;
        const size_t r3 = align.run(&A, &B, T3);
        ygz_compat::se3_to7(T3, u7);
        u7[7] = (float) r3;
        dump(dir + "/tcr_reuse.bin", u7, sizeof u7);
        for (size_t l = 0; l < B.mvImagePyramid.size(); l++)
            for (int y = 0; y < B.mvImagePyramid[l].rows; y++) std::memcpy(B.mvImagePyramid[l].ptr(y), saved[l].ptr(y), (size_t) B.mvImagePyramid[l].cols);
    }
    // (iii) a frame whose extractor still holds its image on the device (the Frame constructor's ComputePyramid ran last on that extractor):
    // the cache takes level 0 and the pyramid from that context device to device -- same bytes, same answer.  A fresh id makes it a miss.
    {
        Frame Bd = B;
        for (auto &m : Bd.mvImagePyramid) m = m.clone();
        Bd.mnId = B.mnId + 1000;
        ex.ComputePyramid(Bd.mImGray);
        Bd.mpORBextractorLeft = &ex;
        if (!ex.ResidentContext(Bd.mvImagePyramid[0])) { fprintf(stderr, "the extractor does not report the image it just processed as resident\n"); return 1; }
        if (ex.ResidentContext(A.mvImagePyramid[0])) { fprintf(stderr, "the extractor reports another image as resident\n"); return 1; }
        SE3f T4;
        const size_t r4 = align.run(&A, &Bd, T4);
        float u7[8];
        ygz_compat::se3_to7(T4, u7);
        u7[7] = (float) r4;
        dump(dir + "/tcr_resident.bin", u7, sizeof u7);
    }
    // TrackWithMotionModel: cur pose = TCR * last pose, then SearchByProjection(cur, last, 15, mono)  (:1072-1093)
    B.mTcw = TCR;
    ORBmatcher matcher(0.9f, true);
    const int nm = matcher.SearchByProjection(B, A, 15.f, true);
    std::vector<int> assigned(B.N, -1);
    for (int i = 0; i < B.N; i++)
        if (B.mvpMapPoints[i]) assigned[i] = (int) (B.mvpMapPoints[i] - mps.data());
    dump(dir + "/match.bin", assigned.data(), assigned.size() * sizeof(int));
    dump(dir + "/nmatch.bin", &nm, sizeof nm);
    // Tracking::SearchLocalPointsDirect: matcher.FindDirectProjection(KF, &cur, mp, px, level) once per candidate (src/Tracking.cc:2210, :2289)
    {
        KeyFrame KF;
        KF.mnId = 7;
        KF.mvKeys = A.mvKeys;
        KF.mvImagePyramid = A.mvImagePyramid;
        KF.mPose = A.mTcw;
        B.mnId = 41;
        const int nd = std::min(A.N, 160);
        std::vector<float> out((size_t) nd * 4);
        ORBmatcher dm;
        for (int i = 0; i < nd; i++) {
            mps[i].mObservations[&KF] = (size_t) i;
            Vector2f px;
            px[0] = A.mvKeys[i].pt.x + ((i % 5) - 2) * 0.75f;      // initial guess: the KeyFrame position, up to 1.5 px off
            px[1] = A.mvKeys[i].pt.y + ((i % 3) - 1) * 0.5f;
            int level = -1;
            const bool ok = dm.FindDirectProjection(&KF, &B, &mps[i], px, level);
            out[4 * i] = px[0]; out[4 * i + 1] = px[1]; out[4 * i + 2] = (float) level; out[4 * i + 3] = ok ? 1.f : 0.f;
        }
        dump(dir + "/direct.bin", out.data(), out.size() * sizeof(float));
    }
    // Tracking::SearchLocalPoints: SearchByProjection(frame, localMapPoints, th) with the fields Frame::isInFrustum sets
    std::vector<MapPoint *> local;
    for (int i = 0; i < A.N; i++) {
        mps[i].mbTrackInView = (i % 7) != 0;
        mps[i].mTrackProjX = A.mvKeys[i].pt.x + 0.5f; mps[i].mTrackProjY = A.mvKeys[i].pt.y - 0.25f;
        mps[i].mTrackViewCos = (i % 3) ? 0.9995f : 0.99f;
        mps[i].mnTrackScaleLevel = A.mvKeys[i].octave;
        local.push_back(&mps[i]);
    }
    Frame A2 = A;                       // match the points back into a copy of their own frame
    A2.mvpMapPoints.assign(A2.N, nullptr);
    ORBmatcher matcher2(0.8f, true);
    const int nm2 = matcher2.SearchByProjection(A2, local, 3.f, true);
    std::vector<int> assigned2(A2.N, -1);
    for (int i = 0; i < A2.N; i++)
        if (A2.mvpMapPoints[i]) assigned2[i] = (int) (A2.mvpMapPoints[i] - mps.data());
    dump(dir + "/match2.bin", assigned2.data(), assigned2.size() * sizeof(int));
    dump(dir + "/nmatch2.bin", &nm2, sizeof nm2);
    // Frame::ExtractORB on a direct-tracked frame (src/Frame.cc:335-337): N existing keys, DSO_KEYPOINT adds FAST-10 grid keys
    {
        Frame C = B;
        C.mvKeys.assign(B.mvKeys.begin(), B.mvKeys.begin() + 150);
        for (auto &k : C.mvKeys) k.angle = 0.f;                      // stale angles: the DSO path recomputes them
        C.N = 150;
        C.mDescriptors = cv::Mat();
        ex(&C, C.mvKeys, cv::_OutputArray(C.mDescriptors), ORBextractor::DSO_KEYPOINT, true);
        dump(dir + "/c_kps.bin", C.mvKeys.data(), C.mvKeys.size() * sizeof(cv::KeyPoint));
        dump(dir + "/c_desc.bin", C.mDescriptors.ptr(0), C.mvKeys.size() * 32);
        // a second direct-tracked frame: the grid size persists inside the extractor
        Frame C2 = A;
        C2.mvKeys.assign(A.mvKeys.begin(), A.mvKeys.begin() + 80);
        C2.N = 80;
        C2.mDescriptors = cv::Mat();
        ex(&C2, C2.mvKeys, cv::_OutputArray(C2.mDescriptors), ORBextractor::DSO_KEYPOINT, true);
        dump(dir + "/c2_kps.bin", C2.mvKeys.data(), C2.mvKeys.size() * sizeof(cv::KeyPoint));
        dump(dir + "/c2_desc.bin", C2.mDescriptors.ptr(0), C2.mvKeys.size() * 32);
        // FAST_KEYPOINT (ComputeKeyPointsFast) on a frame that already holds keys: occupancy of their 5-px cells, re-oriented, new keys appended
        Frame E = B;
        E.mvKeys.assign(B.mvKeys.begin() + 5, B.mvKeys.begin() + 125);
        for (auto &k : E.mvKeys) k.angle = 0.f;
        E.N = 120;
        E.mDescriptors = cv::Mat();
        ex(&E, E.mvKeys, cv::_OutputArray(E.mDescriptors), ORBextractor::FAST_KEYPOINT, true);
        dump(dir + "/e_kps.bin", E.mvKeys.data(), E.mvKeys.size() * sizeof(cv::KeyPoint));
        dump(dir + "/e_desc.bin", E.mDescriptors.ptr(0), E.mvKeys.size() * 32);
        // ORBSLAM_KEYPOINT on a frame that already holds keys: their angle is kept, descriptors first, new keys appended
        Frame D = B;
        D.mvKeys.assign(B.mvKeys.begin() + 10, B.mvKeys.begin() + 70);
        D.N = 60;
        D.mDescriptors = cv::Mat();
        ex(&D, D.mvKeys, cv::_OutputArray(D.mDescriptors), ORBextractor::ORBSLAM_KEYPOINT, true);
        dump(dir + "/d_kps.bin", D.mvKeys.data(), D.mvKeys.size() * sizeof(cv::KeyPoint));
        dump(dir + "/d_desc.bin", D.mDescriptors.ptr(0), D.mvKeys.size() * 32);
    }
    // Tracking::Relocalization refinement: SearchByProjection(cur, KF, found, 10, 100)  (src/Tracking.cc:1830-1860)
    {
        KeyFrame KF;
        KF.mvKeys = A.mvKeys;
        for (int i = 0; i < A.N; i++) {
            mps[i].mfMaxDistance = depth * A.mvScaleFactors[A.mvKeys[i].octave];   // MapPoint::UpdateNormalAndDepth
            mps[i].mfMinDistance = mps[i].mfMaxDistance / A.mvScaleFactors[L - 1];
            KF.mvpMapPoints.push_back((i % 11 == 0) ? nullptr : &mps[i]);
        }
        std::set<MapPoint *> found;
        for (int i = 0; i < A.N; i += 5) found.insert(&mps[i]);
        Frame B2 = B;
        B2.mvpMapPoints.assign(B2.N, nullptr);
        for (int i = 0; i < B2.N; i += 13) B2.mvpMapPoints[i] = &mps[0];          // slots filled by the first pass
        B2.mfLogScaleFactor = std::log(1.2f);
        B2.mnScaleLevels = L;
        ORBmatcher matcher3(0.9f, true);
        const int nm3 = matcher3.SearchByProjection(B2, &KF, found, 10.f, 100);
        std::vector<int> assigned3(B2.N, -1);
        for (int i = 0; i < B2.N; i++)
            if (B2.mvpMapPoints[i] && !(i % 13 == 0 && B2.mvpMapPoints[i] == &mps[0])) assigned3[i] = (int) (B2.mvpMapPoints[i] - mps.data());
        dump(dir + "/match3.bin", assigned3.data(), assigned3.size() * sizeof(int));
        dump(dir + "/nmatch3.bin", &nm3, sizeof nm3);
    }
    // MonocularInitialization: SearchForInitialization(ini, cur, prevMatched, matches, 100)  (src/Tracking.cc:741-743)
    {
        std::vector<cv::Point2f> prev(A.N);
        for (int i = 0; i < A.N; i++) prev[i] = A.mvKeys[i].pt;
        std::vector<int> m12;
        ORBmatcher matcher4(0.9f, true);
        const int nm4 = matcher4.SearchForInitialization(A, B, prev, m12, 100);
        dump(dir + "/match4.bin", m12.data(), m12.size() * sizeof(int));
        dump(dir + "/prev4.bin", prev.data(), prev.size() * sizeof(cv::Point2f));
        dump(dir + "/nmatch4.bin", &nm4, sizeof nm4);
    }
    // TrackReferenceKeyFrame: SearchByBoW(refKF, cur, out)  (src/Tracking.cc:934-936); FeatureVector stand-in: node = leading descriptor bits
    {
        KeyFrame KF;
        KF.mvKeys = A.mvKeys;
        KF.mDescriptors = A.mDescriptors;
        for (int i = 0; i < A.N; i++) {
            KF.mvpMapPoints.push_back((i % 9 == 0) ? nullptr : &mps[i]);
            KF.mFeatVec[A.mDescriptors.ptr<unsigned char>(i)[0] >> 3].push_back(i);
        }
        Frame B3 = B;
        for (int i = 0; i < B3.N; i++) {
            const unsigned node = B3.mDescriptors.ptr<unsigned char>(i)[0] >> 3;
            if (node != 7) B3.mFeatVec[node].push_back(i);                     // a node missing on one side exercises lower_bound
        }
        std::vector<MapPoint *> out;
        ORBmatcher matcher5(0.7f, true);
        const int nm5 = matcher5.SearchByBoW(&KF, B3, out);
        std::vector<int> assigned5(B3.N, -1);
        for (int i = 0; i < B3.N; i++)
            if (out[i]) assigned5[i] = (int) (out[i] - mps.data());
        dump(dir + "/match5.bin", assigned5.data(), assigned5.size() * sizeof(int));
        dump(dir + "/nmatch5.bin", &nm5, sizeof nm5);
    }
    // LocalMapping::CreateNewMapPoints: SearchForTriangulation(KF1 = A, KF2 = B, F12, pairs, false)  (src/LocalMapping.cc); F12 / R2w / t2w / Cw1
    // come from <dir>/tri.f32 (9 + 9 + 3 + 3 floats, written by the Python side); node = leading descriptor bits, node 5 absent in KF1
    {
        std::vector<unsigned char> tr = slurp(dir + "/tri.f32");
        const float *tf = (const float *) tr.data();
        KeyFrame K1, K2;
        Frame *src[2] = {&A, &B};
        KeyFrame *kf[2] = {&K1, &K2};
        for (int s = 0; s < 2; s++) {
            KeyFrame &K = *kf[s];
            const Frame &F = *src[s];
            K.N = F.N;
            K.mvKeys = F.mvKeys;
            K.mDescriptors = F.mDescriptors;
            K.mvScaleFactors = F.mvScaleFactors;
            for (float x : F.mvScaleFactors) K.mvLevelSigma2.push_back(x * x);
            K.fx = Frame::fx; K.fy = Frame::fy; K.cx = Frame::cx; K.cy = Frame::cy;
            for (int i = 0; i < F.N; i++) {
                K.mvpMapPoints.push_back((i % 5 == s) ? &mps[0] : nullptr);
                K.mvuRight.push_back((i % 3 == 0) ? F.mvKeys[i].pt.x - 4.f : -1.f);
                const unsigned node = F.mDescriptors.ptr<unsigned char>(i)[0] >> 4;
                if (!(s == 0 && node == 5)) K.mFeatVec[node].push_back(i);
            }
        }
        Matrix3f F12;
        for (int i = 0; i < 9; i++) { F12.m[i] = tf[i]; K2.mRcw.m[i] = tf[9 + i]; }
        for (int i = 0; i < 3; i++) { K2.mtcw[i] = tf[18 + i]; K1.mOw[i] = tf[21 + i]; }
        std::vector<std::pair<size_t, size_t>> pairs;
        ORBmatcher matcher6(0.6f, true);
        const int nm6 = matcher6.SearchForTriangulation(&K1, &K2, F12, pairs, false);
        std::vector<int> m6(A.N, -1);
        for (const auto &pr : pairs) m6[pr.first] = (int) pr.second;
        if ((int) pairs.size() != nm6) { fprintf(stderr, "SearchForTriangulation: %zu pairs, return value %d\n", pairs.size(), nm6); return 5; }
        dump(dir + "/match6.bin", m6.data(), m6.size() * sizeof(int));
        dump(dir + "/nmatch6.bin", &nm6, sizeof nm6);
    }
    // stereo: left = A, right eye image r.u8 -> ExtractORB(1, imRight) then ComputeStereoMatches  (src/Frame.cc:728-738)
    {
        std::vector<unsigned char> irr = slurp(dir + "/r.u8");
        Frame S = A;
        S.mImRight = cv::Mat(H, W, CV_8UC1, irr.data());
        S.mb = 0.11f; S.mbf = 47.9f;
        ORBextractor exR(600, 1.2f, L, 20, 7);
        exR(&S, S.mvKeysRight, cv::_OutputArray(S.mDescriptorsRight), ORBextractor::ORBSLAM_KEYPOINT, false);
        ex.ComputeStereoMatches(S);
        dump(dir + "/s_kpsr.bin", S.mvKeysRight.data(), S.mvKeysRight.size() * sizeof(cv::KeyPoint));
        dump(dir + "/s_uright.bin", S.mvuRight.data(), S.mvuRight.size() * sizeof(float));
        dump(dir + "/s_depth.bin", S.mvDepth.data(), S.mvDepth.size() * sizeof(float));
    }
    // ORBmatcher objects live on the stacks of several threads at once (Tracking / LocalMapping / LoopClosing): the context pool
    {
        std::vector<int> results(3 * 4, -1);
        std::vector<std::thread> th;
        for (int t = 0; t < 3; t++)
            th.emplace_back([&, t]() {
                for (int it = 0; it < 4; it++) {
                    Frame Bt = B;                                   // private copy: the search writes mvpMapPoints
                    Bt.mvpMapPoints.assign(Bt.N, nullptr);
                    ORBmatcher m(0.9f, true);
                    results[t * 4 + it] = m.SearchByProjection(Bt, A, 15.f, true);
                }
            });
        for (auto &t : th) t.join();
        for (int r : results)
            if (r != nm) { fprintf(stderr, "threaded SearchByProjection: %d != %d\n", r, nm); return 3; }
    }
    cv::Mat d0 = A.mDescriptors.row(0), d1 = A.mDescriptors.row(1);
    printf("shells ok: %d / %d keypoints, align ret %zu, %d matches, dist(0,1)=%d\n", A.N, B.N, ret, nm, ORBmatcher::DescriptorDistance(d0, d1));
    return 0;
}
