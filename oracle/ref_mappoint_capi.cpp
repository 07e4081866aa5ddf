// oracle/ref_mappoint_capi.cpp -- TEST INFRASTRUCTURE: the REFERENCE's ygz::MapPoint (src/MapPoint.cc with the real include/MapPoint.h,
// compiled where they lie into oracle/_ref/libref_mappoint.so over oracle/ref_shim/ with -DYGZ_REF_MAPPOINT: plain-data KeyFrame / Frame /
// Map stubs) behind two of the oracle's flat entry points:
//   yo_distinctive_descriptors : MapPoint::ComputeDistinctiveDescriptors  (src/MapPoint.cc:211-271)  -> index of the winning observation
//   yo_predict_scale           : MapPoint::PredictScale(dist, Frame*)      (:359-373)
// plus yr_distance_invariance (GetMin/MaxDistanceInvariance after UpdateNormalAndDepth, :289-355).
#include <opencv2/core/core.hpp>

#define private public    // mfMaxDistance / mDescriptor are private members of the reference class; the harness sets / reads them
#include "MapPoint.h"
#undef private
#include "ORBmatcher.h"

namespace ygz {
// the projection searches of ORBmatcher.cc are linked into this library only for DescriptorDistance; they are not called here
std::vector<size_t> Frame::GetFeaturesInArea(const float &, const float &, const float &, const int, const int) const { yr_unsupported("Frame::GetFeaturesInArea"); }
}  // namespace ygz

extern "C" {

void yo_distinctive_descriptors(int nPoints, const int *obs_off, const uint8_t *desc, int *best) {
    ygz::Map map;
    for (int p = 0; p < nPoints; p++) {
        const int n = obs_off[p + 1] - obs_off[p];
        if (n <= 0) { best[p] = -1; continue; }
        // one KeyFrame per observation, allocated contiguously: std::map<KeyFrame*, size_t> iterates in pointer order = observation order
        std::vector<ygz::KeyFrame> kfs((size_t) n);
        for (int i = 0; i < n; i++) {
            kfs[i].mDescriptors = cv::Mat(1, 32, CV_8U);
            std::memcpy(kfs[i].mDescriptors.data, desc + 32 * (size_t) (obs_off[p] + i), 32);
            kfs[i].mvuRight.assign(1, -1.f);
            kfs[i].mvKeys.resize(1);
            kfs[i].mvScaleFactors.assign(1, 1.f);
            kfs[i].mnScaleLevels = 1;
        }
        ygz::MapPoint mp(Vector3f(0, 0, 1), &kfs[0], &map);
        for (int i = 0; i < n; i++) mp.AddObservation(&kfs[i], 0);
        mp.ComputeDistinctiveDescriptors();
        const cv::Mat d = mp.GetDescriptor();
        best[p] = -1;
        for (int i = 0; i < n && best[p] < 0; i++)
            if (std::memcmp(d.data, desc + 32 * (size_t) (obs_off[p] + i), 32) == 0) best[p] = i;   // first observation carrying the winning descriptor
    }
}

void yo_predict_scale(const float *ratio, int n, float logScaleFactor, int nScaleLevels, int *level) {
    ygz::Map map;
    ygz::KeyFrame kf;
    kf.mvKeys.resize(1);
    ygz::Frame F;
    F.mfLogScaleFactor = logScaleFactor;
    F.mnScaleLevels = nScaleLevels;
    ygz::MapPoint mp(Vector3f(0, 0, 1), &kf, &map);
    for (int i = 0; i < n; i++) {
        mp.mfMaxDistance = ratio[i];      // ratio = mfMaxDistance / currentDist with currentDist = 1
        level[i] = mp.PredictScale(1.0f, &F);
    }
}

// UpdateNormalAndDepth for a point seen by one reference KeyFrame at camera centre Ow, keypoint octave `level`: out = (min, max) distance
// invariance = (0.8 mfMinDistance, 1.2 mfMaxDistance) with mfMaxDistance = |Pos - Ow| * scale[level], mfMinDistance = mfMaxDistance / scale[n-1]
void yr_distance_invariance(int n, const float *pos, const float *Ow, const int *level, const float *scaleFactors, int nlevels, float *out_min,
                            float *out_max) {
    ygz::Map map;
    for (int i = 0; i < n; i++) {
        ygz::KeyFrame kf;
        kf.mvKeys.resize(1);
        kf.mvKeys[0].octave = level[i];
        kf.mvuRight.assign(1, -1.f);
        kf.mvScaleFactors.assign(scaleFactors, scaleFactors + nlevels);
        kf.mnScaleLevels = nlevels;
        kf.mOw = Vector3f(Ow[3 * i], Ow[3 * i + 1], Ow[3 * i + 2]);
        ygz::MapPoint mp(Vector3f(pos[3 * i], pos[3 * i + 1], pos[3 * i + 2]), &kf, &map);
        mp.AddObservation(&kf, 0);
        mp.UpdateNormalAndDepth();
        out_min[i] = mp.GetMinDistanceInvariance();
        out_max[i] = mp.GetMaxDistanceInvariance();
    }
}

}  // extern "C"
