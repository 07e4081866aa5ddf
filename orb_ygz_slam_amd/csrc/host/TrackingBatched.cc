// TrackingBatched.cc -- the two per-frame loops of ygz::Tracking that call the hot path once per MapPoint, re-bound to libygzf's BATCH entry
// points (product code, host side; optional: without this file the reference's own bodies keep working through the per-call members of
// ORBmatcher.cc).  Compiled inside the reference tree against the reference's own, unchanged include/Tracking.h, it defines
//   void Tracking::SearchLocalPoints()         src/Tracking.cc:1544-1593   isInFrustum over the local map + SearchByProjection(F, MapPoints):
//                                              ONE ygzf_search_local_points call (frustum and window search fused on the device, no host round trip)
//   void Tracking::SearchLocalPointsDirect()   src/Tracking.cc:2174-2326   FindDirectProjection once per (MapPoint, KeyFrame) candidate:
//                                              the candidates of all points are gathered, ONE ygzf_is_in_frustum_batch + ONE
//                                              ygzf_find_direct_projection_batch per half of the function, then the reference's loop consumes the results
// as strong definitions; the link drops the reference's bodies of exactly these two members (objcopy --weaken on Tracking.o, the recipe
// INTEGRATION.md gives for ORBmatcher.o).  Every other member of Tracking stays the reference's.
//
// Why the results cannot differ from the per-call form: the candidates are independent.  isInFrustum reads the frame pose and one MapPoint and
// writes that MapPoint's tracking fields; FindDirectProjection reads two images, two poses and one MapPoint and writes its two outputs.  The
// only coupling in SearchLocalPointsDirect is the coverage grid of its first loop (a point whose projected cell is already taken is skipped,
// and a success marks the cell of the ALIGNED position): the batch evaluates every candidate and the sequential loop below decides, in
// the reference's order, which results are used -- a skipped point's result is simply not read.
#include "ORBextractor.h"   // first: inside the reference tree this is the replacement header (same include guard)
#include "ORBmatcher.h"
#include "Tracking.h"
#include "ygz_compat.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <vector>

#include "../../../include/ygzf.h"
#include "ygzf_pool.h"

namespace {
// MapPoint::mfMaxDistance -- the numerator of MapPoint::PredictScale (src/MapPoint.cc:359-373), which the device frustum needs -- is private and
// has no exact accessor (GetMaxDistanceInvariance() returns 1.2f * it).  Read through [temp.explicit]: access checking does not apply to the
// arguments of an explicit instantiation; include/MapPoint.h stays untouched.
template <typename Tag, typename Tag::type M>
struct MemberOf {
    friend typename Tag::type member_ptr(Tag) { return M; }
};
struct MaxDistanceTag {
    typedef float ygz::MapPoint::*type;
    friend type member_ptr(MaxDistanceTag);
};
template struct MemberOf<MaxDistanceTag, &ygz::MapPoint::mfMaxDistance>;
inline float max_distance(ygz::MapPoint *mp) { return mp->*member_ptr(MaxDistanceTag()); }
// MapPoint::GetDescriptor() returns mDescriptor.clone(): one heap allocation per MapPoint, a thousand per frame (the reference's own loop pays
// them only for the points in view).  The 32 bytes are read in place instead, under the MapPoint's own feature mutex -- the lock GetDescriptor and
// Observations take (src/MapPoint.cc:80-86, :127-130) -- together with the observation count.
#ifdef YGZ_STUB_MAPPOINT   // this repository's boundary build: the MapPoint stand-in of oracle/ref_shim has no mutexes and its GetDescriptor() does not clone
inline void descriptor_and_observations(ygz::MapPoint *mp, uint8_t *desc32, uint8_t *hasObs) {
    const cv::Mat d = mp->GetDescriptor();
    if (!d.empty()) std::memcpy(desc32, d.ptr<uint8_t>(0), 32);
    else std::memset(desc32, 0, 32);
    *hasObs = mp->Observations() > 0;
}
#else
struct DescriptorTag {
    typedef cv::Mat ygz::MapPoint::*type;
    friend type member_ptr(DescriptorTag);
};
template struct MemberOf<DescriptorTag, &ygz::MapPoint::mDescriptor>;
struct FeatureMutexTag {
    typedef std::mutex ygz::MapPoint::*type;
    friend type member_ptr(FeatureMutexTag);
};
template struct MemberOf<FeatureMutexTag, &ygz::MapPoint::mMutexFeatures>;
inline void descriptor_and_observations(ygz::MapPoint *mp, uint8_t *desc32, uint8_t *hasObs) {
    std::unique_lock<std::mutex> lock(mp->*member_ptr(FeatureMutexTag()));
    const cv::Mat &d = mp->*member_ptr(DescriptorTag());
    if (!d.empty()) std::memcpy(desc32, d.ptr<uint8_t>(0), 32);
    else std::memset(desc32, 0, 32);
    *hasObs = mp->nObs > 0;
}
#endif

inline int device() { return ygz::ORBextractor::sDevice; }

struct FrustumPack {   // the MapPoint fields Frame::isInFrustum reads (src/Frame.cc:363-422), as the arrays of ygzf_frustum_in
    std::vector<float> world, normal, maxInv, minInv, maxDist;
    std::vector<uint8_t> cand;
    ygzf_frustum_in in;
    void gather(const std::vector<ygz::MapPoint *> &pts, const std::vector<uint8_t> &candidate, ygz::Frame &F) {
        const size_t n = pts.size();
        // (entries of points that are not candidates are never read: the device tests the candidate flag first -- no zero fill, and the pack
        // itself is kept from frame to frame by its callers, so the steady state allocates nothing)
        world.resize(3 * n); normal.resize(3 * n); maxInv.resize(n); minInv.resize(n); maxDist.resize(n);
        cand = candidate;
        for (size_t i = 0; i < n; i++) {
            if (!cand[i]) continue;
            ygz::MapPoint *mp = pts[i];
            ygz_compat::world_pos(mp, &world[3 * i]);
            const Vector3f nrm = mp->GetNormal();
            normal[3 * i] = nrm[0]; normal[3 * i + 1] = nrm[1]; normal[3 * i + 2] = nrm[2];
            maxInv[i] = mp->GetMaxDistanceInvariance();
            minInv[i] = mp->GetMinDistanceInvariance();
            maxDist[i] = max_distance(mp);
        }
        in.world = world.data(); in.normal = normal.data();
        in.max_dist_inv = maxInv.data(); in.min_dist_inv = minInv.data(); in.mf_max_distance = maxDist.data();
        in.candidate = cand.data();
        for (int r = 0; r < 3; r++) {
            for (int c = 0; c < 3; c++) in.Rcw[3 * r + c] = F.mRcw(r, c);
            in.tcw[r] = F.mtcw[r];
            in.Ow[r] = F.mOw[r];
        }
        in.log_scale_factor = F.mfLogScaleFactor;
        in.viewing_cos_limit = 0.5f;
    }
};

// what isInFrustum leaves in the MapPoint (src/Frame.cc:364, :413-419)
inline void scatter_frustum(ygz::MapPoint *mp, bool inView, float px, float py, float pxr, int level, float viewCos) {
    mp->mbTrackInView = false;
    if (!inView) return;
    mp->mbTrackInView = true;
    mp->mTrackProjX = px;
    mp->mTrackProjXR = pxr;
    mp->mTrackProjY = py;
    mp->mnTrackScaleLevel = level;
    mp->mTrackViewCos = viewCos;
}

inline ygzf_camera camera_of(const ygz::Frame &F) {
    ygzf_camera cam = {ygz::Frame::fx, ygz::Frame::fy, ygz::Frame::cx, ygz::Frame::cy, F.mb, F.mbf,
                       ygz::Frame::mnMinX, ygz::Frame::mnMinY, ygz::Frame::mnMaxX, ygz::Frame::mnMaxY};
    return cam;
}
}  // namespace

// candidates from which Tracking::SearchLocalPoints takes the fused device frustum + matcher call (below: the reference's own loop over the per-call members)
// YGZF_FRUSTUM_DEVICE_MIN overrides the default (a non-negative integer; anything else is ignored with a warning).  0 = always fused.
static int frustum_min_from_env() {
    const int dflt = 1500;
    const char *e = getenv("YGZF_FRUSTUM_DEVICE_MIN");
    if (!e || !*e) return dflt;
    char *end = nullptr;
    const long v = strtol(e, &end, 10);
    if (*end != '\0' || v < 0 || v > 100000000) {
        fprintf(stderr, "[ygzf] YGZF_FRUSTUM_DEVICE_MIN=%s ignored (want a non-negative integer); using %d\n", e, dflt);
        return dflt;
    }
    return (int) v;
}
int ygzf_host_device_frustum_min = frustum_min_from_env();

namespace ygz {

// ---- src/Tracking.cc:1544-1593 ------------------------------------------------------------------------------------------------------------
void Tracking::SearchLocalPoints() {
    // Do not search map points already matched (:1546-1559, unchanged)
    for (std::vector<MapPoint *>::iterator vit = mCurrentFrame.mvpMapPoints.begin(), vend = mCurrentFrame.mvpMapPoints.end(); vit != vend; vit++) {
        MapPoint *pMP = *vit;
        if (pMP) {
            if (pMP->isBad()) {
                *vit = static_cast<MapPoint *>(NULL);
            } else {
                pMP->IncreaseVisible();
                pMP->mnLastFrameSeen = mCurrentFrame.mnId;
                pMP->mbTrackInView = false;
            }
        }
    }
    const int M = (int) mvpLocalMapPoints.size(), nt = mCurrentFrame.N;
    if (M <= 0) return;
    // The device builds the Frame grid over ALL nt keys and reads their descriptor rows; the reference only touches the rows of keys its grid
    // returns.  A frame whose key count ran ahead of its descriptors (SearchLocalPointsDirect appends keys without descriptors; the reference's own
    // flow re-extracts before TrackLocalMap, src/Tracking.cc:1105-1118) would make either read past mDescriptors: refuse it loudly instead.
    if (nt > 0 && (mCurrentFrame.mDescriptors.rows < nt || (int) mCurrentFrame.mvpMapPoints.size() < nt || (int) mCurrentFrame.mvKeys.size() < nt)) {
        char msg[160];
        snprintf(msg, sizeof msg, "frame holds %d keys but %d descriptor rows / %zu MapPoint slots / %zu keypoints: no matching done", nt,
                 mCurrentFrame.mDescriptors.rows, mCurrentFrame.mvpMapPoints.size(), mCurrentFrame.mvKeys.size());
        ygzf_host::report_failure("ygz::Tracking::SearchLocalPoints", msg);
        return;
    }
    // the points the reference's loop hands to isInFrustum (:1564-1575)
    std::vector<uint8_t> cand(M, 0);
    int nCand = 0;
    for (int i = 0; i < M; i++) {
        MapPoint *pMP = mvpLocalMapPoints[i];
        if (pMP->mnLastFrameSeen == mCurrentFrame.mnId) continue;
        if (pMP->isBad()) continue;
        cand[i] = 1;
        nCand++;
    }
    if (nCand == 0) return;
    // search parameters (:1577-1590): ORBmatcher(0.8), checkLevel = false literal
    int th = 1;
    if (mSensor == System::RGBD) th = 3;
    if (mCurrentFrame.mnId < mnLastRelocFrameId + 2) th = 5;
    if (mbDirectFailed) th = 5;
    // A local map of a thousand points: Frame::isInFrustum on the host is ~25 us for all of them, which is what shipping the frustum inputs / outputs
    // and one more launch cost the fused call -- measured 122-125 us fused against 118-120 us for the reference's own loop (profiles/r05_*_boundary_latency.txt).
    // Below kDeviceFrustumMin points the binding therefore IS the reference's loop (isInFrustum per point, then the matcher's device call); the fused
    // device stage takes over where the host loop would dominate.  ygzf_host_device_frustum_min (initialised from YGZF_FRUSTUM_DEVICE_MIN; 0: always fused).
    if (nCand < ygzf_host_device_frustum_min) {
        int nToMatch = 0;
        for (int i = 0; i < M; i++) {
            if (!cand[i]) continue;
            MapPoint *pMP = mvpLocalMapPoints[i];
            if (mCurrentFrame.isInFrustum(pMP, 0.5)) {   // (:1569-1574)
                pMP->IncreaseVisible();
                nToMatch++;
            }
        }
        if (nToMatch > 0) {
            ORBmatcher matcher(0.8);
            matcher.SearchByProjection(mCurrentFrame, mvpLocalMapPoints, th, false);
        }
        return;
    }
    // scratch of this thread's calls, kept from frame to frame (fifteen vectors of a thousand entries were allocated and zeroed per call)
    struct Scratch {
        FrustumPack fp;
        std::vector<uint8_t> hasObs, mpdesc, owner, inView;
        std::vector<float> px, py, pxr, vc;
        std::vector<int> lvl, match;
    };
    static thread_local Scratch S;
    FrustumPack &fp = S.fp;
    fp.gather(mvpLocalMapPoints, cand, mCurrentFrame);
    std::vector<uint8_t> &hasObs = S.hasObs, &mpdesc = S.mpdesc;
    hasObs.resize(M); mpdesc.resize((size_t) M * 32);
    for (int i = 0; i < M; i++) {
        if (!cand[i]) { hasObs[i] = 0; continue; }
        descriptor_and_observations(mvpLocalMapPoints[i], &mpdesc[(size_t) i * 32], &hasObs[i]);
    }
    std::vector<uint8_t> &owner = S.owner, cdescHold;
    owner.assign(std::max(nt, 1), 0);
    for (int i = 0; i < nt; i++) {
        MapPoint *mp = mCurrentFrame.mvpMapPoints[i];
        owner[i] = mp ? (mp->Observations() > 0 ? 2 : 1) : 0;
    }
    const uint8_t *cdesc = nullptr;
    if (nt > 0) {
        if (mCurrentFrame.mDescriptors.isContinuous()) cdesc = mCurrentFrame.mDescriptors.ptr<uint8_t>(0);   // N x 32, one block
        else {
            cdescHold.resize((size_t) nt * 32);
            for (int i = 0; i < nt; i++) std::memcpy(&cdescHold[(size_t) i * 32], mCurrentFrame.mDescriptors.ptr<uint8_t>(i), 32);
            cdesc = cdescHold.data();
        }
    }
    ygzf_frame_view fv;
    fv.n = nt;
    fv.keys = (const ygzf_kp *) mCurrentFrame.mvKeys.data();
    fv.desc = cdesc;
    fv.u_right = mCurrentFrame.mvuRight.empty() ? nullptr : mCurrentFrame.mvuRight.data();
    fv.scale_factors = mCurrentFrame.mvScaleFactors.data();
    fv.nlevels = (int) mCurrentFrame.mvScaleFactors.size();
    const ygzf_camera cam = camera_of(mCurrentFrame);
    std::vector<uint8_t> &inView = S.inView;
    std::vector<float> &px = S.px, &py = S.py, &pxr = S.pxr, &vc = S.vc;
    inView.assign(M, 0);
    px.resize(M); py.resize(M); pxr.resize(M); vc.resize(M);
    std::vector<int> &lvl = S.lvl, &match = S.match;
    lvl.resize(M);
    match.assign(std::max(nt, 1), -1);
    int nmatches = 0;
    ygzf_host::Lease lease(device());
    if (!lease) return;
    ygzf_ctx *c = lease.get();
    if (ygzf_search_local_points(c, &fv, &cam, M, &fp.in, hasObs.data(), mpdesc.data(), (float) th, 0, 0.8f, owner.data(), match.data(), &nmatches,
                                 inView.data(), px.data(), py.data(), pxr.data(), lvl.data(), vc.data()) != YGZF_OK) {
        ygzf_host::report_failure("ygz::Tracking::SearchLocalPoints", ygzf_last_error(c));
        return;
    }
    for (int i = 0; i < M; i++) {
        if (!cand[i]) continue;
        MapPoint *pMP = mvpLocalMapPoints[i];
        scatter_frustum(pMP, inView[i] != 0, px[i], py[i], pxr[i], lvl[i], vc[i]);
        if (inView[i]) pMP->IncreaseVisible();   // :1571
    }
    for (int i = 0; i < nt; i++)
        if (match[i] >= 0) mCurrentFrame.mvpMapPoints[i] = mvpLocalMapPoints[match[i]];
}

// ---- src/Tracking.cc:2174-2326 ------------------------------------------------------------------------------------------------------------
namespace {
struct DirectBatch {
    // candidates of a list of MapPoints, in the reference's order: point by point, per point its SelectNearestKeyframe order
    std::vector<int> first;              // per point: index of its first candidate (first[p + 1] - first[p] candidates)
    std::vector<KeyFrame *> kf;
    std::vector<int> refSlot, level;
    std::vector<float> refT, world, px;
    std::vector<ygzf_kp> refKp;
    std::vector<uint8_t> ok;
};
}  // namespace

// frustum of `pts` (candidate[i] = evaluated at all) on the device, results scattered into the MapPoints as isInFrustum does; inView out
static bool frustum_batch(Frame &F, const std::vector<MapPoint *> &pts, const std::vector<uint8_t> &candidate, std::vector<uint8_t> &inView, const char *who) {
    const int n = (int) pts.size();
    inView.assign(n, 0);
    if (n == 0) return true;
    FrustumPack fp;
    fp.gather(pts, candidate, F);
    std::vector<float> px(n), py(n), pxr(n), vc(n);
    std::vector<int> lvl(n);
    const ygzf_camera cam = camera_of(F);
    ygzf_host::Lease lease(device());
    if (!lease) return false;
    if (ygzf_is_in_frustum_batch(lease.get(), &cam, (int) F.mvScaleFactors.size(), n, &fp.in, inView.data(), px.data(), py.data(), pxr.data(), lvl.data(),
                                 vc.data()) != YGZF_OK) {
        ygzf_host::report_failure(who, ygzf_last_error(lease.get()));
        return false;
    }
    for (int i = 0; i < n; i++)
        if (candidate[i]) scatter_frustum(pts[i], inView[i] != 0, px[i], py[i], pxr[i], lvl[i], vc[i]);
    return true;
}

void Tracking::SearchLocalPointsDirect() {
    static const char *who = "ygz::Tracking::SearchLocalPointsDirect";
    int cntSuccess = 0;
    const int grid_size = 5;
    const int grid_rows = mCurrentFrame.mvImagePyramid[0].rows / grid_size;
    const int grid_cols = mCurrentFrame.mvImagePyramid[0].cols / grid_size;
    std::vector<bool> grid(grid_rows * grid_cols, false);
    const cv::Mat &cur0 = mCurrentFrame.mvImagePyramid[0];
    const int L = (int) mCurrentFrame.mvImagePyramid.size();

    // every FindDirectProjection of `pts` (those with use[i]) as batch calls; false when the device path failed (nothing is matched then)
    auto run_batch = [&](const std::vector<MapPoint *> &pts, const std::vector<uint8_t> &use, DirectBatch &B) -> bool {
        const int n = (int) pts.size();
        B.first.assign(n + 1, 0);
        B.kf.clear(); B.refT.clear(); B.world.clear(); B.px.clear(); B.refKp.clear();
        std::vector<const cv::Mat *> img;
        for (int p = 0; p < n; p++) {
            B.first[p] = (int) B.kf.size();
            if (!use[p]) continue;
            MapPoint *mp = pts[p];
            const std::vector<std::pair<KeyFrame *, size_t> > obs_sorted = SelectNearestKeyframe(mp->GetObservations(), 5);
            float w3[3];
            ygz_compat::world_pos(mp, w3);
            for (const auto &o : obs_sorted) {
                KeyFrame *ref = o.first;
                if ((int) ref->mvImagePyramid.size() != L || ref->mvImagePyramid[0].cols != cur0.cols || ref->mvImagePyramid[0].rows != cur0.rows) {
                    B.kf.push_back(nullptr);              // the member returns false for such a KeyFrame: a candidate that never succeeds
                } else
                    B.kf.push_back(ref);
                float t7[7];
                ygz_compat::se3_to7(ref->GetPose(), t7);
                B.refT.insert(B.refT.end(), t7, t7 + 7);
                B.world.insert(B.world.end(), w3, w3 + 3);
                B.px.push_back(mp->mTrackProjX);
                B.px.push_back(mp->mTrackProjY);
                static_assert(sizeof(cv::KeyPoint) == sizeof(ygzf_kp), "cv::KeyPoint layout");
                ygzf_kp kp;
                std::memcpy(&kp, &ref->mvKeys[o.second], sizeof kp);
                B.refKp.push_back(kp);
            }
        }
        B.first[n] = (int) B.kf.size();
        const int nc = (int) B.kf.size();
        B.ok.assign(nc, 0);
        B.level.assign(nc, 0);
        B.refSlot.assign(nc, 0);
        if (nc == 0) return true;
        if (L < 1 || (int) mCurrentFrame.mvScaleFactors.size() < L) return true;   // (the member returns false for every candidate)
        ygzf_host::ImageCache &dc = ygzf_host::ImageCache::instance();
        ygzf_host::ImageCache::Guard lk(dc);
        if (!dc.prepare(device(), cur0.cols, cur0.rows, L, L > 1 ? mCurrentFrame.mvScaleFactors[1] : 1.2f, who)) return false;
        float curT[7];
        ygz_compat::se3_to7(mCurrentFrame.mTcw, curT);
        const ygzf_camera cam = {Frame::fx, Frame::fy, Frame::cx, Frame::cy, 0, 0, Frame::mnMinX, Frame::mnMinY, Frame::mnMaxX, Frame::mnMaxY};
        // a batch call may reference as many distinct images as the cache has slots: chunks of candidates whose KeyFrames (+ the current frame)
        // fit; within a chunk every image is touched after every image outside it, so the LRU replacement cannot evict one that is in use
        const int maxDistinct = ygzf_host::ImageCache::capacity() - 2;
        int c0 = 0;
        while (c0 < nc) {
            const int curSlot = dc.slot(ygzf_host::ImageCache::kFrame, mCurrentFrame.mnId, cur0.data, cur0.cols, cur0.rows, (int) cur0.step, who,
                                        mCurrentFrame.mpORBextractorLeft ? mCurrentFrame.mpORBextractorLeft->ResidentContext(cur0) : nullptr);
            if (curSlot < 0) return false;
            std::map<KeyFrame *, int> slotOf;
            int c1 = c0;
            for (; c1 < nc; c1++) {
                KeyFrame *ref = B.kf[c1];
                if (!ref) { B.refSlot[c1] = curSlot; continue; }          // placeholder (its result is discarded below)
                auto it = slotOf.find(ref);
                if (it == slotOf.end()) {
                    if ((int) slotOf.size() >= maxDistinct) break;
                    const cv::Mat &r0 = ref->mvImagePyramid[0];
                    const int s = dc.slot(ygzf_host::ImageCache::kKeyFrame, ref->mnId, r0.data, r0.cols, r0.rows, (int) r0.step, who);
                    if (s < 0) return false;
                    it = slotOf.insert(std::make_pair(ref, s)).first;
                }
                B.refSlot[c1] = it->second;
            }
            const int m = c1 - c0;
            if (ygzf_find_direct_projection_batch(dc.ctx(), &cam, curSlot, curT, m, &B.refSlot[c0], &B.refT[7 * (size_t) c0], &B.refKp[c0], &B.world[3 * (size_t) c0],
                                                  &B.px[2 * (size_t) c0], &B.level[c0], &B.ok[c0], nullptr) != YGZF_OK) {
                ygzf_host::report_failure(who, ygzf_last_error(dc.ctx()));
                return false;
            }
            c0 = c1;
        }
        for (int i = 0; i < nc; i++)
            if (!B.kf[i]) B.ok[i] = 0;
        return true;
    };
    // the candidate loop of one point on the batch results (:2205-2221 / :2284-2300): first success at least 20 px inside the image
    auto consume = [&](const DirectBatch &B, int p, std::vector<Vector2f> &matched_pixels) {
        for (int i = B.first[p]; i < B.first[p + 1]; i++) {
            if (!B.ok[i]) continue;
            Vector2f px_curr(B.px[2 * (size_t) i], B.px[2 * (size_t) i + 1]);
            if (px_curr[0] < 20 || px_curr[1] < 20 || px_curr[0] >= cur0.cols - 20 || px_curr[1] >= cur0.rows - 20) continue;
            matched_pixels.push_back(px_curr);
            mCurrentFrame.mvMatchedFrom.push_back(B.kf[i]->mnId);
            break;
        }
    };

    // a point aligned at `at`: the frame gets a key there (size 7, level 0, no angle) that belongs to the point (:2228-2236 / :2307-2313)
    auto append_key = [&](MapPoint *mp, const Vector2f &at) {
        mCurrentFrame.mvKeys.push_back(cv::KeyPoint(cv::Point2f(at[0], at[1]), 7, -1, 0, 0));
        mCurrentFrame.mvpMapPoints.push_back(mp);
        mCurrentFrame.mvDepth.push_back(-1);
        mCurrentFrame.mvbOutlier.push_back(false);
    };
    // mean of the accepted pixels of one point (the reference averages a list that holds at most one entry: the candidate loop stops at the first success)
    auto mean_px = [](const std::vector<Vector2f> &v) {
        Vector2f m(0, 0);
        for (const Vector2f &q : v) m += q;
        return Vector2f(m / v.size());
    };
    auto finish = [&]() {
        mCurrentFrame.N = mCurrentFrame.mvKeys.size();
        mCurrentFrame.mvuRight.resize(mCurrentFrame.N, -1);
    };

    if (!mvpDirectMapPointsCache.empty()) {
        // the cached points: frustum of all of them at once, every candidate of those in view at once, then the reference's loop (:2185-2256)
        const std::vector<MapPoint *> pts(mvpDirectMapPointsCache.begin(), mvpDirectMapPointsCache.end());
        std::vector<uint8_t> notBad(pts.size()), inView;
        for (size_t i = 0; i < pts.size(); i++) notBad[i] = !pts[i]->isBad();
        DirectBatch B;
        if (!frustum_batch(mCurrentFrame, pts, notBad, inView, who) || !run_batch(pts, inView, B)) {
            finish();
            return;
        }
        int p = 0;
        for (auto iter = mvpDirectMapPointsCache.begin(); iter != mvpDirectMapPointsCache.end(); p++) {
            MapPoint *mp = *iter;
            if (!notBad[p] || !inView[p]) {
                iter = mvpDirectMapPointsCache.erase(iter);
                continue;
            }
            int gx = static_cast<int>(mp->mTrackProjX / grid_size);
            int gy = static_cast<int>(mp->mTrackProjY / grid_size);
            int k = gy * grid_cols + gx;
            if (grid[k] == true) {
                iter++;
                continue;
            }
            std::vector<Vector2f> matched_pixels;
            consume(B, p, matched_pixels);
            if (matched_pixels.empty()) {                       // never aligned: the point leaves the cache
                iter = mvpDirectMapPointsCache.erase(iter);
                continue;
            }
            const Vector2f px_ave = mean_px(matched_pixels);
            append_key(mp, px_ave);
            grid[static_cast<int>(px_ave[1] / grid_size) * grid_cols + static_cast<int>(px_ave[0] / grid_size)] = true;   // the cell of the ALIGNED position
            iter++;
            cntSuccess++;
        }
    }
    if (cntSuccess > mnCacheHitTh) {   // enough hits in the cache: the local map is not consulted (:2248-2254)
        finish();
        return;
    }
    UpdateLocalMap();
    {
        // the local map (:2263-2321): points outside the cache, not bad, in view
        const std::vector<MapPoint *> &pts = mvpLocalMapPoints;
        std::vector<uint8_t> eval(pts.size()), inView;
        for (size_t i = 0; i < pts.size(); i++)
            eval[i] = mvpDirectMapPointsCache.find(pts[i]) == mvpDirectMapPointsCache.end() && !pts[i]->isBad();
        DirectBatch B;
        if (frustum_batch(mCurrentFrame, pts, eval, inView, who) && run_batch(pts, inView, B)) {
            for (size_t p = 0; p < pts.size(); p++) {
                MapPoint *mp = pts[p];
                // (a point that occurs twice in the list was put into the cache by its first occurrence: :2264)
                if (!eval[p] || mvpDirectMapPointsCache.find(mp) != mvpDirectMapPointsCache.end() || !inView[p]) continue;
                std::vector<Vector2f> matched_pixels;
                consume(B, (int) p, matched_pixels);
                if (matched_pixels.empty()) continue;
                append_key(mp, mean_px(matched_pixels));
                mvpDirectMapPointsCache.insert(mp);
            }
        }
    }
    finish();
}

}  // namespace ygz
