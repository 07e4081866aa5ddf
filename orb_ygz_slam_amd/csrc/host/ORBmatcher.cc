// ORBmatcher.cc -- the hot-path members of ygz::ORBmatcher over libygzf's C ABI (product code, host side): each packs the Frame / MapPoint /
// KeyFrame fields the reference function reads into the plain arrays of its ygzf_* entry point and scatters the result back.
// Inside the reference tree this file is compiled against the reference's own, unchanged include/ORBmatcher.h and replaces the bodies of
//   ORBmatcher(float, bool), DescriptorDistance, the three Tracking-side SearchByProjection overloads, SearchByBoW(KeyFrame*, Frame&, ...),
//   SearchForInitialization, SearchForTriangulation, FindDirectProjection   (src/ORBmatcher.cc:36-133, 155-263, 375-478, 596-741, 1218-1602);
// the other LocalMapping / LoopClosing members (Fuse x2, SearchBySim3, SearchByProjection(KF, Scw, ...), SearchByBoW(KF, KF, ...)) are
// outside the hot path and keep their reference bodies (link recipe).
#include "ORBextractor.h"   // first: inside the reference tree this is the replacement header (same include guard)
#include "ORBmatcher.h"     // the reference's own header (reference tree) or standalone/ORBmatcher.h, by include path
#include "ygz_compat.h"

#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>

#include "../../../include/ygzf.h"
#include "ygzf_pool.h"

namespace ygz {

const int ORBmatcher::TH_HIGH = 100;
const int ORBmatcher::TH_LOW = 50;
const int ORBmatcher::HISTO_LENGTH = 30;

ORBmatcher::ORBmatcher(float nnratio, bool checkOri) : mfNNratio(nnratio), mbCheckOrientation(checkOri) {}

// One pair of 32-byte rows: a popcount over 4 x u64 on the host (a device launch for 64 bytes would be absurd); the batched
// device form is ygzf_descriptor_distance / the matcher kernels.
int ORBmatcher::DescriptorDistance(const cv::Mat &a, const cv::Mat &b) {
    uint64_t pa[4], pb[4];
    std::memcpy(pa, a.data, 32);
    std::memcpy(pb, b.data, 32);
    int dist = 0;
    for (int i = 0; i < 4; i++) dist += __builtin_popcountll(pa[i] ^ pb[i]);
    return dist;
}

namespace {
// ORBmatcher objects are created on the stack per call from several threads: every call leases a context of the device
// extractors are created on (ORBextractor::sDevice) from the per-device pool.
inline int device() { return ORBextractor::sDevice; }
// the rows of an N x 32 descriptor matrix as one block: the Mat's own buffer when it is continuous (what the extractor produces), a packed copy
// in `hold` otherwise (a Mat assembled from row ranges)
inline const uint8_t *desc_rows(const cv::Mat &D, int n, std::vector<uint8_t> &hold) {
    if (n <= 0) return nullptr;
    if (D.isContinuous() && D.cols == 32) return D.ptr<uint8_t>(0);
    hold.resize((size_t) n * 32);
    for (int i = 0; i < n; i++) std::memcpy(&hold[(size_t) i * 32], D.ptr<uint8_t>(i), 32);
    return hold.data();
}
}  // namespace

int ORBmatcher::SearchByProjection(Frame &CurrentFrame, const Frame &LastFrame, const float th, const bool bMono, bool checkLevel) {
    const int nt = CurrentFrame.N, nq = LastFrame.N;
    if (nt <= 0 || nq <= 0) return 0;
    ygzf_host::Lease lease(device());
    if (!lease) return 0;
    ygzf_ctx *c = lease.get();
    // ---- pack LastFrame ----
    std::vector<uint8_t> valid(nq), outl(nq), obs(nq), mpdesc((size_t) nq * 32);
    std::vector<float> world((size_t) nq * 3);
    for (int i = 0; i < nq; i++) {
        MapPoint *mp = LastFrame.mvpMapPoints[i];
        valid[i] = mp != nullptr;
        outl[i] = LastFrame.mvbOutlier[i];
        if (mp) {
            obs[i] = mp->Observations() > 0;
            ygz_compat::world_pos(mp, &world[3 * (size_t) i]);
            const cv::Mat d = mp->GetDescriptor();
            std::memcpy(&mpdesc[(size_t) i * 32], d.ptr<uint8_t>(0), 32);
        }
    }
    // ---- pack CurrentFrame ----
    std::vector<uint8_t> owner(nt), cdescHold;
    for (int i = 0; i < nt; i++) {
        MapPoint *mp = CurrentFrame.mvpMapPoints[i];
        owner[i] = mp ? (mp->Observations() > 0 ? 2 : 1) : 0;
    }
    const uint8_t *cdesc = desc_rows(CurrentFrame.mDescriptors, nt, cdescHold);
    ygzf_frame_view cur;
    cur.n = nt;
    cur.keys = (const ygzf_kp *) CurrentFrame.mvKeys.data();
    cur.desc = cdesc;
    cur.u_right = CurrentFrame.mvuRight.empty() ? nullptr : CurrentFrame.mvuRight.data();
    cur.scale_factors = CurrentFrame.mvScaleFactors.data();
    cur.nlevels = (int) CurrentFrame.mvScaleFactors.size();
    ygzf_camera cam = {Frame::fx, Frame::fy, Frame::cx, Frame::cy, CurrentFrame.mb, CurrentFrame.mbf,
                       Frame::mnMinX, Frame::mnMinY, Frame::mnMaxX, Frame::mnMaxY};
    float Rcw[9], tcw[3], Rlw[9], tlw[3];
    ygz_compat::se3_to_Rt(CurrentFrame.mTcw, Rcw, tcw);
    ygz_compat::se3_to_Rt(LastFrame.mTcw, Rlw, tlw);
    std::vector<int> match(nt, -1);
    int nmatches = 0;
    const int rc = ygzf_search_by_projection_last(c, &cur, &cam, nq, (const ygzf_kp *) LastFrame.mvKeys.data(), valid.data(), outl.data(),
                                                  obs.data(), world.data(), mpdesc.data(), Rcw, tcw, Rlw, tlw, th, bMono, checkLevel,
                                                  mbCheckOrientation, owner.data(), match.data(), &nmatches);
    if (rc != YGZF_OK) {
        ygzf_host::report_failure("ygz::ORBmatcher::SearchByProjection", ygzf_last_error(c));
        return 0;
    }
    for (int i2 = 0; i2 < nt; i2++) {
        if (match[i2] >= 0) CurrentFrame.mvpMapPoints[i2] = LastFrame.mvpMapPoints[match[i2]];
        else if (match[i2] == -2) CurrentFrame.mvpMapPoints[i2] = static_cast<MapPoint *>(nullptr);   // culled by the rotation check
    }
    return nmatches;
}

// src/ORBmatcher.cc:43-126
int ORBmatcher::SearchByProjection(Frame &F, const std::vector<MapPoint *> &vpMapPoints, const float th, bool checkLevel) {
    const int nt = F.N, M = (int) vpMapPoints.size();
    if (nt <= 0 || M <= 0) return 0;
    ygzf_host::Lease lease(device());
    if (!lease) return 0;
    ygzf_ctx *c = lease.get();
    // (scratch of this thread's calls, kept from frame to frame: nine vectors of a thousand entries were allocated and zeroed per call; entries of
    // points that are not in view are never read by the device -- it tests the in-view flag first)
    struct Scratch {
        std::vector<uint8_t> tiv, bad, obs, mpdesc, owner;
        std::vector<float> px, py, pxr, vc;
        std::vector<int> lvl, match;
    };
    static thread_local Scratch S;
    std::vector<uint8_t> &tiv = S.tiv, &bad = S.bad, &obs = S.obs, &mpdesc = S.mpdesc;
    std::vector<float> &px = S.px, &py = S.py, &pxr = S.pxr, &vc = S.vc;
    std::vector<int> &lvl = S.lvl;
    tiv.resize(M); bad.assign(M, 0); obs.assign(M, 0); mpdesc.resize((size_t) M * 32);
    px.resize(M); py.resize(M); pxr.resize(M); vc.resize(M); lvl.assign(M, 0);
    for (int i = 0; i < M; i++) {
        MapPoint *mp = vpMapPoints[i];
        tiv[i] = mp->mbTrackInView;
        if (!tiv[i]) continue;
        bad[i] = mp->isBad();
        obs[i] = mp->Observations() > 0;
        px[i] = mp->mTrackProjX; py[i] = mp->mTrackProjY; pxr[i] = mp->mTrackProjXR; vc[i] = mp->mTrackViewCos;
        lvl[i] = mp->mnTrackScaleLevel;
        const cv::Mat d = mp->GetDescriptor();
        std::memcpy(&mpdesc[(size_t) i * 32], d.ptr<uint8_t>(0), 32);
    }
    std::vector<uint8_t> &owner = S.owner, cdescHold;
    owner.resize(nt);
    for (int i = 0; i < nt; i++) {
        MapPoint *mp = F.mvpMapPoints[i];
        owner[i] = mp ? (mp->Observations() > 0 ? 2 : 1) : 0;
    }
    const uint8_t *cdesc = desc_rows(F.mDescriptors, nt, cdescHold);
    ygzf_frame_view fv;
    fv.n = nt;
    fv.keys = (const ygzf_kp *) F.mvKeys.data();
    fv.desc = cdesc;
    fv.u_right = F.mvuRight.empty() ? nullptr : F.mvuRight.data();
    fv.scale_factors = F.mvScaleFactors.data();
    fv.nlevels = (int) F.mvScaleFactors.size();
    ygzf_camera cam = {Frame::fx, Frame::fy, Frame::cx, Frame::cy, F.mb, F.mbf, Frame::mnMinX, Frame::mnMinY, Frame::mnMaxX, Frame::mnMaxY};
    std::vector<int> &match = S.match;
    match.assign(nt, -1);
    int nmatches = 0;
    const int rc = ygzf_search_by_projection_mappoints(c, &fv, &cam, M, tiv.data(), bad.data(), obs.data(), px.data(), py.data(), pxr.data(),
                                                       vc.data(), lvl.data(), mpdesc.data(), th, checkLevel, mfNNratio, owner.data(),
                                                       match.data(), &nmatches);
    if (rc != YGZF_OK) ygzf_host::report_failure("ygz::ORBmatcher::SearchByProjection", ygzf_last_error(c));
    if (rc != YGZF_OK) return 0;
    for (int i = 0; i < nt; i++)
        if (match[i] >= 0) F.mvpMapPoints[i] = vpMapPoints[match[i]];
    return nmatches;
}

// src/ORBmatcher.cc:1352-1469.  The scalar prologue (:1371-1400) runs here with the reference's expressions -- MapPoint::PredictScale
// keeps its own logf -- and the window search / in-order slot resolution / rotation histogram run on the device.
int ORBmatcher::SearchByProjection(Frame &CurrentFrame, KeyFrame *pKF, const std::set<MapPoint *> &sAlreadyFound, const float th, const int ORBdist) {
    const int nt = CurrentFrame.N;
    const std::vector<MapPoint *> vpMPs = pKF->GetMapPointMatches();
    const int M = (int) vpMPs.size();
    if (nt <= 0 || M <= 0) return 0;
    float Rcw[9], tcw[3], Ow[3];
    ygz_compat::se3_to_Rt(CurrentFrame.mTcw, Rcw, tcw);
    for (int i = 0; i < 3; i++) Ow[i] = -1 * (Rcw[i] * tcw[0] + Rcw[3 + i] * tcw[1] + Rcw[6 + i] * tcw[2]);
    std::vector<uint8_t> valid(M, 0), mpdesc((size_t) M * 32);
    std::vector<float> pu(M), pv(M), ang(M);
    std::vector<int> lvl(M);
    for (int i = 0; i < M; i++) {
        MapPoint *pMP = vpMPs[i];
        if (!pMP || pMP->isBad() || sAlreadyFound.count(pMP)) continue;
        float x3Dw[3];
        ygz_compat::world_pos(pMP, x3Dw);
        const float xc = (Rcw[0] * x3Dw[0] + Rcw[1] * x3Dw[1] + Rcw[2] * x3Dw[2]) + tcw[0];
        const float yc = (Rcw[3] * x3Dw[0] + Rcw[4] * x3Dw[1] + Rcw[5] * x3Dw[2]) + tcw[1];
        const float zc = (Rcw[6] * x3Dw[0] + Rcw[7] * x3Dw[1] + Rcw[8] * x3Dw[2]) + tcw[2];
        const float invzc = 1.0 / zc;
        const float u = Frame::fx * xc * invzc + Frame::cx;
        const float v = Frame::fy * yc * invzc + Frame::cy;
        if (u < Frame::mnMinX || u > Frame::mnMaxX) continue;
        if (v < Frame::mnMinY || v > Frame::mnMaxY) continue;
        const float PO[3] = {x3Dw[0] - Ow[0], x3Dw[1] - Ow[1], x3Dw[2] - Ow[2]};
        float dist3D = std::sqrt(PO[0] * PO[0] + PO[1] * PO[1] + PO[2] * PO[2]);
        const float maxDistance = pMP->GetMaxDistanceInvariance();
        const float minDistance = pMP->GetMinDistanceInvariance();
        if (dist3D < minDistance || dist3D > maxDistance) continue;
        valid[i] = 1;
        pu[i] = u;
        pv[i] = v;
        lvl[i] = pMP->PredictScale(dist3D, &CurrentFrame);
        ang[i] = pKF->mvKeys[i].angle;
        const cv::Mat d = pMP->GetDescriptor();
        std::memcpy(&mpdesc[(size_t) i * 32], d.ptr<uint8_t>(0), 32);
    }
    std::vector<uint8_t> owner(nt), cdescHold;
    for (int i = 0; i < nt; i++) owner[i] = CurrentFrame.mvpMapPoints[i] != nullptr;
    const uint8_t *cdesc = desc_rows(CurrentFrame.mDescriptors, nt, cdescHold);
    ygzf_frame_view cur;
    cur.n = nt;
    cur.keys = (const ygzf_kp *) CurrentFrame.mvKeys.data();
    cur.desc = cdesc;
    cur.u_right = nullptr;
    cur.scale_factors = CurrentFrame.mvScaleFactors.data();
    cur.nlevels = (int) CurrentFrame.mvScaleFactors.size();
    ygzf_camera cam = {Frame::fx, Frame::fy, Frame::cx, Frame::cy, CurrentFrame.mb, CurrentFrame.mbf,
                       Frame::mnMinX, Frame::mnMinY, Frame::mnMaxX, Frame::mnMaxY};
    ygzf_host::Lease lease(device());
    if (!lease) return 0;
    ygzf_ctx *c = lease.get();
    std::vector<int> match(nt, -1);
    int nmatches = 0;
    const int rc = ygzf_search_by_projection_kf(c, &cur, &cam, M, valid.data(), pu.data(), pv.data(), lvl.data(), ang.data(), mpdesc.data(), th, ORBdist,
                                                mbCheckOrientation, owner.data(), match.data(), &nmatches);
    if (rc != YGZF_OK) ygzf_host::report_failure("ygz::ORBmatcher::SearchByProjection", ygzf_last_error(c));
    if (rc != YGZF_OK) return 0;
    for (int i2 = 0; i2 < nt; i2++) {
        if (match[i2] >= 0) CurrentFrame.mvpMapPoints[i2] = vpMPs[match[i2]];
        else if (match[i2] == -2) CurrentFrame.mvpMapPoints[i2] = static_cast<MapPoint *>(nullptr);
    }
    return nmatches;
}

// src/ORBmatcher.cc:155-263.  The FeatureVector merge-join runs here exactly as in the reference (:169-247); the per-node brute force,
// the slot chain and the rotation histogram run on the device.
int ORBmatcher::SearchByBoW(KeyFrame *pKF, Frame &F, std::vector<MapPoint *> &vpMapPointMatches) {
    const std::vector<MapPoint *> vpMapPointsKF = pKF->GetMapPointMatches();
    vpMapPointMatches = std::vector<MapPoint *>(F.N, static_cast<MapPoint *>(nullptr));
    const DBoW2::FeatureVector &vFeatVecKF = pKF->mFeatVec;
    std::vector<int> kfOff{0}, fOff{0}, kfIdx, fIdx;
    DBoW2::FeatureVector::const_iterator KFit = vFeatVecKF.begin(), Fit = F.mFeatVec.begin();
    const DBoW2::FeatureVector::const_iterator KFend = vFeatVecKF.end(), Fend = F.mFeatVec.end();
    while (KFit != KFend && Fit != Fend) {
        if (KFit->first == Fit->first) {
            kfIdx.insert(kfIdx.end(), KFit->second.begin(), KFit->second.end());
            fIdx.insert(fIdx.end(), Fit->second.begin(), Fit->second.end());
            kfOff.push_back((int) kfIdx.size());
            fOff.push_back((int) fIdx.size());
            KFit++;
            Fit++;
        } else if (KFit->first < Fit->first) {
            KFit = vFeatVecKF.lower_bound(Fit->first);
        } else {
            Fit = F.mFeatVec.lower_bound(KFit->first);
        }
    }
    const int nNodes = (int) kfOff.size() - 1, nKF = (int) vpMapPointsKF.size();
    if (nNodes <= 0 || F.N <= 0 || nKF <= 0) return 0;
    std::vector<uint8_t> valid(nKF), kfHold, fHold;
    for (int i = 0; i < nKF; i++) valid[i] = vpMapPointsKF[i] && !vpMapPointsKF[i]->isBad();
    const uint8_t *kfDesc = desc_rows(pKF->mDescriptors, nKF, kfHold), *fDesc = desc_rows(F.mDescriptors, F.N, fHold);
    ygzf_host::Lease lease(device());
    if (!lease) return 0;
    ygzf_ctx *c = lease.get();
    std::vector<int> match(F.N, -1);
    int nmatches = 0;
    const int rc = ygzf_search_by_bow(c, nNodes, kfOff.data(), kfIdx.data(), fOff.data(), fIdx.data(), nKF, valid.data(), (const ygzf_kp *) pKF->mvKeys.data(),
                                      kfDesc, F.N, (const ygzf_kp *) F.mvKeys.data(), fDesc, mfNNratio, mbCheckOrientation, match.data(),
                                      &nmatches);
    if (rc != YGZF_OK) ygzf_host::report_failure("ygz::ORBmatcher::SearchByBoW", ygzf_last_error(c));
    if (rc != YGZF_OK) return 0;
    for (int i = 0; i < F.N; i++)
        if (match[i] >= 0) vpMapPointMatches[i] = vpMapPointsKF[match[i]];
    return nmatches;
}

// src/ORBmatcher.cc:596-741 (LocalMapping::CreateNewMapPoints, once per neighbour KeyFrame of every new KeyFrame).  The FeatureVector
// merge-join runs here as in the reference (:631-717); the per-node brute force, the epipole exclusion disc, CheckDistEpipolarLine
// (:136-153) and the rotation histogram run on the device (ygzf_search_for_triangulation).
int ORBmatcher::SearchForTriangulation(KeyFrame *pKF1, KeyFrame *pKF2, Matrix3f &F12, std::vector<std::pair<size_t, size_t> > &vMatchedPairs,
                                       const bool bOnlyStereo) {
    vMatchedPairs.clear();
    const DBoW2::FeatureVector &vFeatVec1 = pKF1->mFeatVec;
    const DBoW2::FeatureVector &vFeatVec2 = pKF2->mFeatVec;
    std::vector<int> off1{0}, off2{0}, idx1, idx2;
    DBoW2::FeatureVector::const_iterator f1it = vFeatVec1.begin(), f2it = vFeatVec2.begin();
    const DBoW2::FeatureVector::const_iterator f1end = vFeatVec1.end(), f2end = vFeatVec2.end();
    while (f1it != f1end && f2it != f2end) {
        if (f1it->first == f2it->first) {
            idx1.insert(idx1.end(), f1it->second.begin(), f1it->second.end());
            idx2.insert(idx2.end(), f2it->second.begin(), f2it->second.end());
            off1.push_back((int) idx1.size());
            off2.push_back((int) idx2.size());
            f1it++;
            f2it++;
        } else if (f1it->first < f2it->first) {
            f1it = vFeatVec1.lower_bound(f2it->first);
        } else {
            f2it = vFeatVec2.lower_bound(f1it->first);
        }
    }
    const int nNodes = (int) off1.size() - 1, n1 = pKF1->N, n2 = pKF2->N;
    if (nNodes <= 0 || n1 <= 0 || n2 <= 0) return 0;
    std::vector<uint8_t> has1(n1), has2(n2), hold1, hold2;
    for (int i = 0; i < n1; i++) has1[i] = pKF1->GetMapPoint(i) != nullptr;
    for (int i = 0; i < n2; i++) has2[i] = pKF2->GetMapPoint(i) != nullptr;
    ygzf_frame_view v1, v2;
    v1.n = n1; v1.keys = (const ygzf_kp *) pKF1->mvKeys.data(); v1.desc = desc_rows(pKF1->mDescriptors, n1, hold1);
    v1.u_right = (int) pKF1->mvuRight.size() == n1 ? pKF1->mvuRight.data() : nullptr;
    v1.scale_factors = nullptr; v1.nlevels = 0;
    v2.n = n2; v2.keys = (const ygzf_kp *) pKF2->mvKeys.data(); v2.desc = desc_rows(pKF2->mDescriptors, n2, hold2);
    v2.u_right = (int) pKF2->mvuRight.size() == n2 ? pKF2->mvuRight.data() : nullptr;
    v2.scale_factors = pKF2->mvScaleFactors.data(); v2.nlevels = (int) pKF2->mvScaleFactors.size();
    if (pKF2->mvLevelSigma2.size() != pKF2->mvScaleFactors.size()) return 0;
    float F[9], R[9], t[3], Cw[3];
    const Vector3f Cw1 = pKF1->GetCameraCenter();
    const Matrix3f R2w = pKF2->GetRotation();
    const Vector3f t2w = pKF2->GetTranslation();
    for (int r = 0; r < 3; r++) {
        for (int cc = 0; cc < 3; cc++) { F[3 * r + cc] = F12(r, cc); R[3 * r + cc] = R2w(r, cc); }
        t[r] = t2w[r];
        Cw[r] = Cw1[r];
    }
    ygzf_camera cam2 = {pKF2->fx, pKF2->fy, pKF2->cx, pKF2->cy, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    ygzf_host::Lease lease(device());
    if (!lease) return 0;
    ygzf_ctx *c = lease.get();
    std::vector<int> match12(n1, -1);
    int nmatches = 0;
    const int rc = ygzf_search_for_triangulation(c, nNodes, off1.data(), idx1.data(), off2.data(), idx2.data(), &v1, has1.data(), &v2, has2.data(),
                                                 pKF2->mvLevelSigma2.data(), F, Cw, R, t, &cam2, bOnlyStereo, mbCheckOrientation, match12.data(),
                                                 &nmatches);
    if (rc != YGZF_OK) {
        ygzf_host::report_failure("ygz::ORBmatcher::SearchForTriangulation", ygzf_last_error(c));
        return 0;
    }
    vMatchedPairs.reserve(nmatches);
    for (int i = 0; i < n1; i++)       // :731-736
        if (match12[i] >= 0) vMatchedPairs.push_back(std::make_pair((size_t) i, (size_t) match12[i]));
    return nmatches;
}

// src/ORBmatcher.cc:375-478
int ORBmatcher::SearchForInitialization(Frame &F1, Frame &F2, std::vector<cv::Point2f> &vbPrevMatched, std::vector<int> &vnMatches12, int windowSize) {
    vnMatches12 = std::vector<int>(F1.N, -1);
    if (F1.N <= 0 || F2.N <= 0) return 0;
    std::vector<uint8_t> h1, h2;
    const uint8_t *d1 = desc_rows(F1.mDescriptors, F1.N, h1), *d2 = desc_rows(F2.mDescriptors, F2.N, h2);
    ygzf_frame_view v1, v2;
    v1.n = F1.N; v1.keys = (const ygzf_kp *) F1.mvKeys.data(); v1.desc = d1; v1.u_right = nullptr;
    v1.scale_factors = F1.mvScaleFactors.data(); v1.nlevels = (int) F1.mvScaleFactors.size();
    v2.n = F2.N; v2.keys = (const ygzf_kp *) F2.mvKeys.data(); v2.desc = d2; v2.u_right = nullptr;
    v2.scale_factors = F2.mvScaleFactors.data(); v2.nlevels = (int) F2.mvScaleFactors.size();
    ygzf_camera cam = {Frame::fx, Frame::fy, Frame::cx, Frame::cy, F2.mb, F2.mbf, Frame::mnMinX, Frame::mnMinY, Frame::mnMaxX, Frame::mnMaxY};
    static_assert(sizeof(cv::Point2f) == 8, "cv::Point2f layout");
    ygzf_host::Lease lease(device());
    if (!lease) return 0;
    ygzf_ctx *c = lease.get();
    int nmatches = 0;
    const int rc = ygzf_search_for_initialization(c, &v1, &v2, &cam, (float *) vbPrevMatched.data(), windowSize, mfNNratio, mbCheckOrientation,
                                                  vnMatches12.data(), &nmatches);
    if (rc != YGZF_OK) ygzf_host::report_failure("ygz::ORBmatcher::SearchForInitialization", ygzf_last_error(c));
    return rc == YGZF_OK ? nmatches : 0;
}

// src/ORBmatcher.cc:1574-1602 (+ GetWarpAffineMatrix :1525-1548, GetBestSearchLevel / GetBilateralInterpUchar include/ORBmatcher.h:185-211,
// WarpAffine :1550-1572, ygz::Align2D src/Align.cc:8-104) for ONE (MapPoint, KeyFrame) candidate -- the signature Tracking::
// SearchLocalPointsDirect calls (src/Tracking.cc:2210, :2289).  The KeyFrames' and the current frame's level-0 images live in an
// HBM-resident cache on the device shared with SparseImgAlign::run (ygzf_host::ImageCache: their pyramids are rebuilt there by the same
// resize kernel the extractor used, so they equal the host copies): an image is uploaded the first time it is referenced; slots are
// recycled least recently used.  A launch per candidate is latency-bound: callers that own the candidate loop should hand the whole list to
// ygzf_find_direct_projection_batch instead (INTEGRATION.md shows that binding for SearchLocalPointsDirect); this member exists so that the
// unmodified caller keeps working.
bool ORBmatcher::FindDirectProjection(KeyFrame *ref, Frame *curr, MapPoint *mp, Vector2f &px_curr, int &search_level) {
    const int L = (int) ref->mvImagePyramid.size();
    if (L < 1 || (int) curr->mvImagePyramid.size() != L || (int) curr->mvScaleFactors.size() < L) return false;
    const cv::Mat &cur0 = curr->mvImagePyramid[0], &ref0 = ref->mvImagePyramid[0];
    if (cur0.cols != ref0.cols || cur0.rows != ref0.rows) return false;
    static const char *who = "ygz::ORBmatcher::FindDirectProjection";
    ygzf_host::ImageCache &dc = ygzf_host::ImageCache::instance();
    ygzf_host::ImageCache::Guard lk(dc);
    if (!dc.prepare(device(), cur0.cols, cur0.rows, L, L > 1 ? curr->mvScaleFactors[1] : 1.2f, who)) return false;
    const int curSlot = dc.slot(ygzf_host::ImageCache::kFrame, curr->mnId, cur0.data, cur0.cols, cur0.rows, (int) cur0.step, who,
                                curr->mpORBextractorLeft ? curr->mpORBextractorLeft->ResidentContext(cur0) : nullptr);
    const int slot = dc.slot(ygzf_host::ImageCache::kKeyFrame, ref->mnId, ref0.data, ref0.cols, ref0.rows, (int) ref0.step, who);
    if (curSlot < 0 || slot < 0) return false;
    const int index = (int) mp->GetObservations()[ref];
    const cv::KeyPoint kp = ref->mvKeys[index];
    float refT[7], curT[7], world[3], px[2] = {px_curr[0], px_curr[1]};
    ygz_compat::se3_to7(ref->GetPose(), refT);
    ygz_compat::se3_to7(curr->mTcw, curT);
    ygz_compat::world_pos(mp, world);
    ygzf_camera cam = {Frame::fx, Frame::fy, Frame::cx, Frame::cy, 0, 0, Frame::mnMinX, Frame::mnMinY, Frame::mnMaxX, Frame::mnMaxY};
    int level = 0;
    uint8_t ok = 0;
    static_assert(sizeof(cv::KeyPoint) == sizeof(ygzf_kp), "cv::KeyPoint layout");
    if (ygzf_find_direct_projection_batch(dc.ctx(), &cam, curSlot, curT, 1, &slot, refT, (const ygzf_kp *) &kp, world, px, &level, &ok, nullptr) != YGZF_OK) {
        ygzf_host::report_failure("ygz::ORBmatcher::FindDirectProjection", ygzf_last_error(dc.ctx()));
        return false;
    }
    px_curr[0] = px[0];
    px_curr[1] = px[1];
    search_level = level;
    return ok != 0;
}

}  // namespace ygz
