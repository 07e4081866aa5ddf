// oracle/ref_frame_capi.cpp -- TEST INFRASTRUCTURE: the REFERENCE's ygz::Frame -- src/Frame.cc with the REAL include/Frame.h, compiled where
// they lie into oracle/_ref/libref_frame.so over oracle/ref_shim/ (-DYGZ_REF_FRAME; IMU / vocabulary / double-precision pose types are
// aborting stand-ins, MapPoint and KeyFrame the plain-data stubs; the reference's own ORBextractor.cc and ORBmatcher.cc are linked in) --
// behind three of the oracle's flat entry points:
//   yo_features_in_area       : AssignFeaturesToGrid + PosInGrid + GetFeaturesInArea   (src/Frame.cc:314-330, 424-493)
//   yo_is_in_frustum          : isInFrustum                                            (:363-422)
//   yo_compute_stereo_matches : ComputeStereoMatches                                   (:509-682)
#include <opencv2/core/core.hpp>

#define private public
#define protected public
#include "ORBmatcher.h"   // pulls the real Frame.h and ORBextractor.h
#undef private
#undef protected

namespace {
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

std::vector<cv::KeyPoint> to_cv_keys(const ygzo::KeyPoint *k, int n) {
    std::vector<cv::KeyPoint> out;
    for (int i = 0; i < n; i++) out.push_back(cv::KeyPoint(k[i].x, k[i].y, k[i].size, k[i].angle, k[i].response, k[i].octave, k[i].class_id));
    return out;
}

void set_statics(const yo_frame *f) {   // Frame's camera / bounds / grid scales are static members set by the first constructed frame (:126-150)
    ygz::Frame::fx = f->fx; ygz::Frame::fy = f->fy; ygz::Frame::cx = f->cx; ygz::Frame::cy = f->cy;
    ygz::Frame::mnMinX = f->minX; ygz::Frame::mnMaxX = f->maxX; ygz::Frame::mnMinY = f->minY; ygz::Frame::mnMaxY = f->maxY;
    ygz::Frame::mfGridElementWidthInv = static_cast<float>(FRAME_GRID_COLS) / (ygz::Frame::mnMaxX - ygz::Frame::mnMinX);
    ygz::Frame::mfGridElementHeightInv = static_cast<float>(FRAME_GRID_ROWS) / (ygz::Frame::mnMaxY - ygz::Frame::mnMinY);
}

void fill(ygz::Frame &F, const yo_frame *f) {
    set_statics(f);
    F.N = f->N;
    F.mvKeys = to_cv_keys(f->keys, f->N);
    F.mvuRight.assign(f->N, -1.f);
    if (f->uRight) F.mvuRight.assign(f->uRight, f->uRight + f->N);
    F.mb = f->mb; F.mbf = f->mbf;
    F.mnScaleLevels = f->nlevels;
    F.mvScaleFactors.assign(f->scaleFactors, f->scaleFactors + f->nlevels);
}
}  // namespace

namespace ygz {
// src/MapPoint.cc:359-373 for the stub MapPoint (pinned separately by tests/test_ref_mappoint.py)
int MapPoint::PredictScale(const float &currentDist, Frame *pF) {
    const float ratio = mfMaxDistance / currentDist;
    int nScale = (int) std::ceil(std::log(ratio) / pF->mfLogScaleFactor);
    if (nScale < 0) nScale = 0;
    else if (nScale >= pF->mnScaleLevels) nScale = pF->mnScaleLevels - 1;
    return nScale;
}
int MapPoint::PredictScale(const float &, KeyFrame *) { yr_unsupported("MapPoint::PredictScale(KeyFrame*)"); }
bool Align2D(const cv::Mat &, uint8_t *, uint8_t *, const int, Vector2f &, bool) { yr_unsupported("Align2D"); }
}  // namespace ygz

extern "C" {

int yo_features_in_area(const yo_frame *f, float x, float y, float r, int minLevel, int maxLevel, int *out, int cap) {
    ygz::Frame F;
    fill(F, f);
    F.AssignFeaturesToGrid();
    const std::vector<size_t> idx = F.GetFeaturesInArea(x, y, r, minLevel, maxLevel);
    for (size_t i = 0; i < idx.size() && (int) i < cap; i++) out[i] = (int) idx[i];
    return (int) idx.size();
}

void yo_is_in_frustum(const yo_frame *f, int M, const float *world, const float *normal, const float *maxDistInv, const float *minDistInv,
                      const float *mfMaxDistance, const float *Rcw, const float *tcw, const float *Ow, float logScaleFactor, int nScaleLevels,
                      float viewingCosLimit, uint8_t *in_view, float *projX, float *projY, float *projXR, int *level, float *viewCos) {
    ygz::Frame F;
    fill(F, f);
    F.mfLogScaleFactor = logScaleFactor;
    F.mnScaleLevels = nScaleLevels;
    for (int i = 0; i < 9; i++) F.mRcw.m[i] = Rcw[i];
    F.mtcw = Vector3f(tcw[0], tcw[1], tcw[2]);
    F.mOw = Vector3f(Ow[0], Ow[1], Ow[2]);
    for (int i = 0; i < M; i++) {
        ygz::MapPoint mp;
        mp.mWorldPos = Vector3f(world[3 * i], world[3 * i + 1], world[3 * i + 2]);
        mp.mNormal = Vector3f(normal[3 * i], normal[3 * i + 1], normal[3 * i + 2]);
        mp.maxDistInv = maxDistInv[i]; mp.minDistInv = minDistInv[i]; mp.mfMaxDistance = mfMaxDistance[i];
        const bool ok = F.isInFrustum(&mp, viewingCosLimit);
        in_view[i] = ok ? 1 : 0;
        if (ok) { projX[i] = mp.mTrackProjX; projY[i] = mp.mTrackProjY; projXR[i] = mp.mTrackProjXR; level[i] = mp.mnTrackScaleLevel; viewCos[i] = mp.mTrackViewCos; }
    }
}

// the first argument is the oracle's extractor handle there; here: (nlevels, scaleFactor) come separately, see yr_stereo_config
static int g_nlevels = 8;
static float g_scale = 1.2f;
void yr_stereo_config(int nlevels, float scaleFactor) { g_nlevels = nlevels; g_scale = scaleFactor; }

void yo_compute_stereo_matches(void *, const uint8_t *imgL, const uint8_t *imgR, int w, int h, int N, const ygzo::KeyPoint *keysL, const uint8_t *descL,
                               int Nr, const ygzo::KeyPoint *keysR, const uint8_t *descR, float mb, float mbf, float *uRight, float *depth) {
    ygz::ORBextractor exL(1000, g_scale, g_nlevels, 20, 7), exR(1000, g_scale, g_nlevels, 20, 7);
    auto wrap = [&](const uint8_t *p) { cv::Mat m(h, w, CV_8U); std::memcpy(m.data, p, (size_t) w * h); return m; };
    exL.ComputePyramid(wrap(imgL));
    exR.ComputePyramid(wrap(imgR));
    ygz::Frame F;
    F.mpORBextractorLeft = &exL;
    F.mpORBextractorRight = &exR;
    F.N = N;
    F.mvKeys = to_cv_keys(keysL, N);
    F.mvKeysRight = to_cv_keys(keysR, Nr);
    F.mDescriptors = cv::Mat(std::max(N, 1), 32, CV_8U);
    F.mDescriptorsRight = cv::Mat(std::max(Nr, 1), 32, CV_8U);
    if (N) std::memcpy(F.mDescriptors.data, descL, (size_t) N * 32);
    if (Nr) std::memcpy(F.mDescriptorsRight.data, descR, (size_t) Nr * 32);
    F.mvScaleFactors = exL.GetScaleFactors();
    F.mvInvScaleFactors = exL.GetInverseScaleFactors();
    F.mb = mb; F.mbf = mbf;
    F.ComputeStereoMatches();
    for (int i = 0; i < N; i++) { uRight[i] = F.mvuRight[i]; depth[i] = F.mvDepth[i]; }
}

}  // extern "C"
