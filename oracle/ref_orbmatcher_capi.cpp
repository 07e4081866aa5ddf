// oracle/ref_orbmatcher_capi.cpp -- TEST INFRASTRUCTURE: the REFERENCE's ygz::ORBmatcher, compiled from /root/reference/src/ORBmatcher.cc
// where it lies (oracle/Makefile target `ref_matcher`, output oracle/_ref/libref_orbmatcher.so) against oracle/ref_shim/ (OpenCV stand-in +
// matcher_stubs.h: plain-data Frame / KeyFrame / MapPoint, 3x3 arithmetic, DBoW2::FeatureVector as a std::map).
//
// The entry points have the SAME NAMES AND SIGNATURES as the oracle's (oracle_capi.cpp: yo_search_by_projection_last, ...): they unpack
// the flat arrays into Frame / MapPoint objects, call the reference's member function, and report which MapPoint ended up in which
// keypoint slot.  oracle_py.reference_matcher() swaps the library under the Python wrappers, so tests/test_ref_matcher.py drives both
// implementations with identical inputs.
//
// Pinned by this: every line of the five search functions and of DescriptorDistance / ComputeThreeMaxima / RadiusByViewingCos.
// Not pinned (restated here, as in the oracle): Frame::GetFeaturesInArea + grid assignment (src/Frame.cc), MapPoint::PredictScale
// (src/MapPoint.cc:359-373), the 3x3 float arithmetic behind Eigen / Sophus.
#include <opencv2/core/core.hpp>   // oracle/ref_shim: mini_cv.h + matcher_stubs.h (YGZ_REF_MATCHER)

#define private public             // the warped patch (_patch_with_border) is a private member of the reference class; the test reads it
#include "ORBmatcher.h"            // the reference's own header
#undef private
#include "ygz_oracle.h"            // grid restatement (ygzo::Grid), struct layouts of the flat API

namespace {

struct FrameAux {
    ygzo::FrameView view;
    ygzo::Grid grid;
};

struct yo_frame {  // as in oracle_capi.cpp
    int N;
    const ygzo::KeyPoint *keys;
    const uint8_t *desc;
    const float *uRight;
    float minX, minY, maxX, maxY;
    float fx, fy, cx, cy, mb, mbf;
    const float *scaleFactors;
    int nlevels;
};

ygzo::FrameView to_view(const yo_frame *f) {
    ygzo::FrameView v;
    v.N = f->N; v.keys = f->keys; v.desc = f->desc; v.uRight = f->uRight;
    v.minX = f->minX; v.minY = f->minY; v.maxX = f->maxX; v.maxY = f->maxY;
    v.gridInvW = (float) ygzo::Grid::COLS / (f->maxX - f->minX);
    v.gridInvH = (float) ygzo::Grid::ROWS / (f->maxY - f->minY);
    v.fx = f->fx; v.fy = f->fy; v.cx = f->cx; v.cy = f->cy; v.mb = f->mb; v.mbf = f->mbf;
    v.scaleFactors = f->scaleFactors; v.nlevels = f->nlevels;
    return v;
}

cv::Mat desc_rows(const uint8_t *d, int n) {
    cv::Mat m(std::max(n, 1), 32, CV_8U);
    if (n) std::memcpy(m.data, d, (size_t) n * 32);
    return m;
}

std::vector<cv::KeyPoint> to_cv_keys(const ygzo::KeyPoint *k, int n) {
    std::vector<cv::KeyPoint> out;
    for (int i = 0; i < n; i++) out.push_back(cv::KeyPoint(k[i].x, k[i].y, k[i].size, k[i].angle, k[i].response, k[i].octave, k[i].class_id));
    return out;
}

void fill_frame(ygz::Frame &F, FrameAux &aux, const yo_frame *f, float logScaleFactor = 0.f) {
    aux.view = to_view(f);
    aux.grid.Assign(aux.view);
    F.grid = &aux;
    F.N = f->N;
    F.mvKeys = to_cv_keys(f->keys, f->N);
    F.mvuRight.assign(f->N, -1.f);
    if (f->uRight) F.mvuRight.assign(f->uRight, f->uRight + f->N);
    F.mDescriptors = desc_rows(f->desc, f->N);
    F.mvpMapPoints.assign(f->N, (ygz::MapPoint *) nullptr);
    F.mvbOutlier.assign(f->N, false);
    F.fx = f->fx; F.fy = f->fy; F.cx = f->cx; F.cy = f->cy; F.mb = f->mb; F.mbf = f->mbf;
    F.mnMinX = f->minX; F.mnMaxX = f->maxX; F.mnMinY = f->minY; F.mnMaxY = f->maxY;
    F.mnScaleLevels = f->nlevels;
    F.mfLogScaleFactor = logScaleFactor;
    F.mvScaleFactors.assign(f->scaleFactors, f->scaleFactors + f->nlevels);
    F.mvInvScaleFactors.clear();
    for (float s : F.mvScaleFactors) F.mvInvScaleFactors.push_back(1.0f / s);
}

SE3f pose(const float *R, const float *t) {
    SE3f T;
    for (int i = 0; i < 9; i++) T.R.m[i] = R[i];
    T.t = Vector3f(t[0], t[1], t[2]);
    return T;
}

// the initial occupants of Cur.mvpMapPoints: owner 1 = a MapPoint nobody observes yet, owner 2 = an observed one
struct Occupants {
    ygz::MapPoint free_mp, taken_mp;
    Occupants() { free_mp.nObs = 0; taken_mp.nObs = 1; }
    void apply(ygz::Frame &F, const uint8_t *owner) {
        for (int i = 0; i < F.N; i++) F.mvpMapPoints[i] = owner[i] == 0 ? nullptr : (owner[i] == 1 ? &free_mp : &taken_mp);
    }
};

// after the call: slot i2 holds candidate k -> match k, owner by its observations; the initial occupant -> untouched; NULL -> owner 0
// (whether the slot was matched and then culled by the rotation check cannot be seen from outside: reported as -1, see the test)
void report(const ygz::Frame &F, const std::vector<ygz::MapPoint> &cands, const Occupants &occ, uint8_t *owner, int *match) {
    for (int i = 0; i < F.N; i++) {
        ygz::MapPoint *p = F.mvpMapPoints[i];
        if (!p) { owner[i] = 0; match[i] = -1; continue; }
        if (p == &occ.free_mp || p == &occ.taken_mp) { match[i] = -1; continue; }
        match[i] = (int) (p - cands.data());
        owner[i] = p->nObs > 0 ? 2 : 1;
    }
}

}  // namespace

namespace ygz {
// src/Frame.cc:424-481 through the oracle's restatement of the 64 x 48 grid
std::vector<size_t> Frame::GetFeaturesInArea(const float &x, const float &y, const float &r, const int minLevel, const int maxLevel) const {
    const FrameAux *a = (const FrameAux *) grid;
    std::vector<int> idx;
    a->grid.FeaturesInArea(a->view, x, y, r, minLevel, maxLevel, idx);
    return std::vector<size_t>(idx.begin(), idx.end());
}
// src/MapPoint.cc:359-373: float log / ceil overloads (<cmath> with `using namespace std`)
int MapPoint::PredictScale(const float &currentDist, Frame *pF) {
    const float ratio = mfMaxDistance / currentDist;
    int nScale = (int) std::ceil(std::log(ratio) / pF->mfLogScaleFactor);
    if (nScale < 0) nScale = 0;
    else if (nScale >= pF->mnScaleLevels) nScale = pF->mnScaleLevels - 1;
    return nScale;
}
int MapPoint::PredictScale(const float &, KeyFrame *) { yr_unsupported("MapPoint::PredictScale(KeyFrame*)"); }
}  // namespace ygz

extern "C" {

int yo_descriptor_distance(const uint8_t *a, const uint8_t *b) { return ygz::ORBmatcher::DescriptorDistance(desc_rows(a, 1), desc_rows(b, 1)); }

// ORBmatcher::SearchByProjection(Frame &CurrentFrame, const Frame &LastFrame, th, bMono, checkLevel)   src/ORBmatcher.cc:1218-1350
int yo_search_by_projection_last(const yo_frame *cur, int lastN, const ygzo::KeyPoint *last_keys, const uint8_t *mp_valid, const uint8_t *outlier,
                                 const uint8_t *mp_has_obs, const float *mp_world, const uint8_t *mp_desc, const float *Rcw, const float *tcw,
                                 const float *Rlw, const float *tlw, float th, int bMono, int checkLevel, int checkOri, uint8_t *cur_owner,
                                 int *cur_match) {
    ygz::Frame Cur, Last;
    FrameAux aux;
    fill_frame(Cur, aux, cur);
    Occupants occ;
    occ.apply(Cur, cur_owner);
    Cur.mTcw = pose(Rcw, tcw);
    Last.N = lastN;
    Last.mvKeys = to_cv_keys(last_keys, lastN);
    Last.mTcw = pose(Rlw, tlw);
    std::vector<ygz::MapPoint> mps((size_t) std::max(lastN, 1));
    Last.mvpMapPoints.assign(lastN, (ygz::MapPoint *) nullptr);
    Last.mvbOutlier.assign(lastN, false);
    for (int i = 0; i < lastN; i++) {
        mps[i].mWorldPos = Vector3f(mp_world[3 * i], mp_world[3 * i + 1], mp_world[3 * i + 2]);
        mps[i].mDescriptor = desc_rows(mp_desc + 32 * (size_t) i, 1);
        mps[i].nObs = mp_has_obs[i] ? 1 : 0;
        if (mp_valid[i]) Last.mvpMapPoints[i] = &mps[i];
        Last.mvbOutlier[i] = outlier[i] != 0;
    }
    ygz::ORBmatcher matcher(0.9f, checkOri != 0);          // Tracking::TrackWithMotionModel: ORBmatcher matcher(0.9, true)
    const int n = matcher.SearchByProjection(Cur, Last, th, bMono != 0, checkLevel != 0);
    report(Cur, mps, occ, cur_owner, cur_match);
    return n;
}

// ORBmatcher::SearchByProjection(Frame &F, const vector<MapPoint*> &vpMapPoints, th, checkLevel)       :43-126
int yo_search_by_projection_mappoints(const yo_frame *F_, int M, const uint8_t *track_in_view, const uint8_t *bad, const uint8_t *mp_has_obs,
                                      const float *projX, const float *projY, const float *projXR, const float *viewCos, const int *scaleLevel,
                                      const uint8_t *mp_desc, float th, int checkLevel, float nnratio, uint8_t *owner, int *match) {
    ygz::Frame F;
    FrameAux aux;
    fill_frame(F, aux, F_);
    Occupants occ;
    occ.apply(F, owner);
    std::vector<ygz::MapPoint> mps((size_t) std::max(M, 1));
    std::vector<ygz::MapPoint *> vp;
    for (int i = 0; i < M; i++) {
        ygz::MapPoint &p = mps[i];
        p.mbTrackInView = track_in_view[i] != 0;
        p.mbBad = bad[i] != 0;
        p.nObs = mp_has_obs[i] ? 1 : 0;
        p.mTrackProjX = projX[i]; p.mTrackProjY = projY[i]; p.mTrackProjXR = projXR[i]; p.mTrackViewCos = viewCos[i];
        p.mnTrackScaleLevel = scaleLevel[i];
        p.mDescriptor = desc_rows(mp_desc + 32 * (size_t) i, 1);
        vp.push_back(&p);
    }
    ygz::ORBmatcher matcher(nnratio, true);
    const int n = matcher.SearchByProjection(F, vp, th, checkLevel != 0);
    report(F, mps, occ, owner, match);
    return n;
}

// ORBmatcher::SearchByProjection(Frame &CurrentFrame, KeyFrame *pKF, const set<MapPoint*> &sAlreadyFound, th, ORBdist)   :1352-1469
int yo_search_by_projection_kf(const yo_frame *cur, int M, const uint8_t *usable, const float *world, const float *maxDistInv, const float *minDistInv,
                               const float *mfMaxDistance, const float *kf_angle, const uint8_t *mp_desc, const float *Rcw, const float *tcw,
                               float logScaleFactor, int nScaleLevels, float th, int ORBdist, int checkOri, uint8_t *cur_owner, int *cur_match,
                               uint8_t *out_valid, float *out_u, float *out_v, int *out_level) {
    ygz::Frame Cur;
    FrameAux aux;
    fill_frame(Cur, aux, cur, logScaleFactor);
    Cur.mnScaleLevels = nScaleLevels;
    Occupants occ;
    occ.apply(Cur, cur_owner);
    Cur.mTcw = pose(Rcw, tcw);
    ygz::KeyFrame KF;
    std::vector<ygz::MapPoint> mps((size_t) std::max(M, 1));
    KF.mvKeys.resize(M);
    KF.mvpMapPoints.assign(M, (ygz::MapPoint *) nullptr);
    for (int i = 0; i < M; i++) {
        ygz::MapPoint &p = mps[i];
        p.mWorldPos = Vector3f(world[3 * i], world[3 * i + 1], world[3 * i + 2]);
        p.maxDistInv = maxDistInv[i]; p.minDistInv = minDistInv[i]; p.mfMaxDistance = mfMaxDistance[i];
        p.mDescriptor = desc_rows(mp_desc + 32 * (size_t) i, 1);
        KF.mvKeys[i].angle = kf_angle[i];
        if (usable[i]) KF.mvpMapPoints[i] = &p;      // usable = non-null, not bad, not in sAlreadyFound
        if (out_valid) out_valid[i] = 0;             // the scalar prologue is internal to the reference function: not reported
    }
    (void) out_u; (void) out_v; (void) out_level;
    std::set<ygz::MapPoint *> found;
    ygz::ORBmatcher matcher(0.9f, checkOri != 0);         // Tracking::Relocalization: ORBmatcher matcher2(0.9, true)
    const int n = matcher.SearchByProjection(Cur, &KF, found, th, ORBdist);
    report(Cur, mps, occ, cur_owner, cur_match);
    return n;
}

// ORBmatcher::SearchForInitialization(Frame &F1, Frame &F2, vbPrevMatched, vnMatches12, windowSize)     :375-478
int yo_search_for_initialization(const yo_frame *F1_, const yo_frame *F2_, float *prevMatchedXY, int windowSize, float nnratio, int checkOri,
                                 int *matches12) {
    ygz::Frame F1, F2;
    FrameAux a1, a2;
    fill_frame(F1, a1, F1_);
    fill_frame(F2, a2, F2_);
    std::vector<cv::Point2f> prev;
    for (int i = 0; i < F1.N; i++) prev.push_back(cv::Point2f(prevMatchedXY[2 * i], prevMatchedXY[2 * i + 1]));
    std::vector<int> m12;
    ygz::ORBmatcher matcher(nnratio, checkOri != 0);
    const int n = matcher.SearchForInitialization(F1, F2, prev, m12, windowSize);
    for (int i = 0; i < F1.N; i++) {
        matches12[i] = m12[i];
        prevMatchedXY[2 * i] = prev[i].x;
        prevMatchedXY[2 * i + 1] = prev[i].y;
    }
    return n;
}

// ORBmatcher::SearchByBoW(KeyFrame *pKF, Frame &F, vector<MapPoint*> &vpMapPointMatches)                :155-263
// The flat API carries the JOINED node list; both FeatureVectors get exactly those nodes (ids 2k+1), plus one node on either side that the
// other does not have, so that the reference's merge-join (equal ids, lower_bound skips) runs as well.
int yo_search_by_bow(int nNodes, const int *kf_off, const int *kf_idx, const int *f_off, const int *f_idx, const uint8_t *kf_valid,
                     const ygzo::KeyPoint *kf_keys, const uint8_t *kf_desc, int nF, const ygzo::KeyPoint *f_keys, const uint8_t *f_desc, float nnratio,
                     int checkOri, int *match) {
    int nKF = 0;
    for (int k = 0; k < nNodes; k++)
        for (int j = kf_off[k]; j < kf_off[k + 1]; j++) nKF = std::max(nKF, kf_idx[j] + 1);
    ygz::KeyFrame KF;
    ygz::Frame F;
    KF.mvKeys = to_cv_keys(kf_keys, nKF);
    KF.mDescriptors = desc_rows(kf_desc, nKF);
    std::vector<ygz::MapPoint> mps((size_t) std::max(nKF, 1));
    KF.mvpMapPoints.assign(nKF, (ygz::MapPoint *) nullptr);
    for (int i = 0; i < nKF; i++) {
        if (kf_valid[i]) KF.mvpMapPoints[i] = &mps[i];
    }
    F.N = nF;
    F.mvKeys = to_cv_keys(f_keys, nF);
    F.mDescriptors = desc_rows(f_desc, nF);
    for (int k = 0; k < nNodes; k++) {
        const unsigned id = 2u * (unsigned) k + 1u;
        for (int j = kf_off[k]; j < kf_off[k + 1]; j++) KF.mFeatVec[id].push_back((unsigned) kf_idx[j]);
        for (int j = f_off[k]; j < f_off[k + 1]; j++) F.mFeatVec[id].push_back((unsigned) f_idx[j]);
    }
    KF.mFeatVec[0];                       // empty nodes on one side only
    F.mFeatVec[2u * (unsigned) nNodes + 2u];
    if (nNodes > 1) { KF.mFeatVec[2]; F.mFeatVec[4]; }
    std::vector<ygz::MapPoint *> out;
    ygz::ORBmatcher matcher(nnratio, checkOri != 0);
    const int n = matcher.SearchByBoW(&KF, F, out);
    for (int i = 0; i < nF; i++) match[i] = out[i] ? (int) (out[i] - mps.data()) : -1;
    return n;
}

// ORBmatcher::SearchForTriangulation(KeyFrame *pKF1, KeyFrame *pKF2, Matrix3f &F12, vector<pair<size_t,size_t>> &vMatchedPairs,
//                                    const bool bOnlyStereo)                                             :596-741 (+ CheckDistEpipolarLine :136-153)
// Joined node list -> two FeatureVectors as in yo_search_by_bow.  The reference returns only the surviving pairs: match12 = -1 wherever
// the oracle reports -1 (never matched) or -2 (culled by the rotation check).
int yo_search_for_triangulation(int nNodes, const int *off1, const int *idx1, const int *off2, const int *idx2, int n1, const ygzo::KeyPoint *keys1,
                                const uint8_t *desc1, const uint8_t *has_mp1, const float *uRight1, int n2, const ygzo::KeyPoint *keys2,
                                const uint8_t *desc2, const uint8_t *has_mp2, const float *uRight2, int nlevels2, const float *scaleFactors2,
                                const float *levelSigma2_2, const float *F12, const float *Cw1, const float *R2w, const float *t2w,
                                const float *cam2, int onlyStereo, int checkOri, int *match12) {
    ygz::KeyFrame K1, K2;
    ygz::MapPoint some;
    auto fill = [&](ygz::KeyFrame &K, int n, const ygzo::KeyPoint *keys, const uint8_t *desc, const uint8_t *has_mp, const float *uR) {
        K.N = n;
        K.mvKeys = to_cv_keys(keys, n);
        K.mDescriptors = desc_rows(desc, n);
        K.mvuRight.assign(n, -1.f);
        if (uR) K.mvuRight.assign(uR, uR + n);
        K.mvpMapPoints.assign(n, (ygz::MapPoint *) nullptr);
        for (int i = 0; i < n; i++)
            if (has_mp[i]) K.mvpMapPoints[i] = &some;
    };
    fill(K1, n1, keys1, desc1, has_mp1, uRight1);
    fill(K2, n2, keys2, desc2, has_mp2, uRight2);
    K1.mOw = Vector3f(Cw1[0], Cw1[1], Cw1[2]);
    K2.mHasPose = true;
    for (int i = 0; i < 9; i++) K2.mRcw.m[i] = R2w[i];
    K2.mtcw = Vector3f(t2w[0], t2w[1], t2w[2]);
    K2.fx = cam2[0]; K2.fy = cam2[1]; K2.cx = cam2[2]; K2.cy = cam2[3];
    K2.mnScaleLevels = nlevels2;
    K2.mvScaleFactors.assign(scaleFactors2, scaleFactors2 + nlevels2);
    K2.mvLevelSigma2.assign(levelSigma2_2, levelSigma2_2 + nlevels2);
    for (int k = 0; k < nNodes; k++) {
        const unsigned id = 2u * (unsigned) k + 1u;
        for (int j = off1[k]; j < off1[k + 1]; j++) K1.mFeatVec[id].push_back((unsigned) idx1[j]);
        for (int j = off2[k]; j < off2[k + 1]; j++) K2.mFeatVec[id].push_back((unsigned) idx2[j]);
    }
    K1.mFeatVec[0];                       // nodes on one side only: the merge-join's lower_bound skips run as well
    K2.mFeatVec[2u * (unsigned) nNodes + 2u];
    if (nNodes > 1) { K1.mFeatVec[2]; K2.mFeatVec[4]; }
    Matrix3f F;
    for (int i = 0; i < 9; i++) F.m[i] = F12[i];
    std::vector<std::pair<size_t, size_t> > pairs;
    ygz::ORBmatcher matcher(0.6f, checkOri != 0);
    const int n = matcher.SearchForTriangulation(&K1, &K2, F, pairs, onlyStereo != 0);
    for (int i = 0; i < n1; i++) match12[i] = -1;
    size_t prev = 0;
    for (size_t j = 0; j < pairs.size(); j++) {
        if (j && pairs[j].first <= prev) return -1000;   // :731-736 emits ascending i1
        prev = pairs[j].first;
        match12[pairs[j].first] = (int) pairs[j].second;
    }
    if ((int) pairs.size() != n) return -1001;
    return n;
}

// ORBmatcher::FindDirectProjection(KeyFrame *ref, Frame *curr, MapPoint *mp, Vector2f &px_curr, int &search_level)   :1573-1602, with
// GetWarpAffineMatrix :1525-1548, WarpAffine :1550-1572, GetBestSearchLevel / GetBilateralInterpUchar (include/ORBmatcher.h:185-211) and
// the reference's own ygz::Align2D (src/Align.cc, compiled into this library).  Pyramids come in level by level (tight 8-bit images).
void yr_find_direct_projection_batch(int nlevels, const float *scaleFactors, const float *invLevelSigma2, const int *lw, const int *lh, int n_refs,
                                     const uint8_t *const *ref_levels /* n_refs x nlevels */, const uint8_t *const *cur_levels /* nlevels */,
                                     const float *cur_Tcw7, float fx, float fy, float cx, float cy, int n, const int *ref_slot, const float *ref_Tcw7,
                                     const ygzo::KeyPoint *ref_kp, const float *mp_world, float *px_curr, int *search_level, uint8_t *success,
                                     uint8_t *patches) {
    auto level_mat = [&](const uint8_t *p, int l) {
        cv::Mat m(lh[l], lw[l], CV_8U);
        std::memcpy(m.data, p, (size_t) lw[l] * lh[l]);
        return m;
    };
    auto se3 = [](const float *T7) {
        ygzo::SE3f q;
        std::memcpy(q.q, T7, 16);
        std::memcpy(q.t, T7 + 4, 12);
        return SE3f(q);
    };
    std::vector<ygz::KeyFrame> kfs((size_t) std::max(n_refs, 1));
    for (int r = 0; r < n_refs; r++) {
        ygz::KeyFrame &K = kfs[r];
        K.fx = fx; K.fy = fy; K.cx = cx; K.cy = cy;
        K.mvScaleFactors.assign(scaleFactors, scaleFactors + nlevels);
        K.mvInvLevelSigma2.assign(invLevelSigma2, invLevelSigma2 + nlevels);
        for (int l = 0; l < nlevels; l++) K.mvImagePyramid.push_back(level_mat(ref_levels[r * nlevels + l], l));
    }
    ygz::Frame cur;
    cur.fx = fx; cur.fy = fy; cur.cx = cx; cur.cy = cy;
    cur.mTcw = se3(cur_Tcw7);
    cur.mvScaleFactors.assign(scaleFactors, scaleFactors + nlevels);
    for (int l = 0; l < nlevels; l++) {
        cur.mvInvScaleFactors.push_back(1.0f / scaleFactors[l]);
        cur.mvImagePyramid.push_back(level_mat(cur_levels[l], l));
    }
    ygz::ORBmatcher matcher;
    for (int i = 0; i < n; i++) {
        ygz::KeyFrame &K = kfs[ref_slot[i]];
        K.mPose = se3(ref_Tcw7 + 7 * i);                       // candidates of one KeyFrame share its pose in the caller's data
        K.mvKeys.assign(1, cv::KeyPoint(ref_kp[i].x, ref_kp[i].y, ref_kp[i].size, ref_kp[i].angle, ref_kp[i].response, ref_kp[i].octave, ref_kp[i].class_id));
        ygz::MapPoint mp;
        mp.mWorldPos = Vector3f(mp_world[3 * i], mp_world[3 * i + 1], mp_world[3 * i + 2]);
        mp.mObservations[&K] = 0;
        Vector2f px(px_curr[2 * i], px_curr[2 * i + 1]);
        int sl = 0;
        success[i] = matcher.FindDirectProjection(&K, &cur, &mp, px, sl) ? 1 : 0;
        px_curr[2 * i] = px[0];
        px_curr[2 * i + 1] = px[1];
        search_level[i] = sl;
        if (patches) std::memcpy(patches + 100 * (size_t) i, matcher._patch_with_border, 100);
    }
}

}  // extern "C"
