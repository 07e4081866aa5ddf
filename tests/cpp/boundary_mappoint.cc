// tests/cpp/boundary_mappoint.cc -- TEST DRIVER (built by tests/cpp/build_boundary.sh into tests/cpp/bin/libboundary_mappoint.so): the REFERENCE's own
// src/MapPoint.cc, compiled where it lies with the REAL include/MapPoint.h (private members, mutexes and all; KeyFrame / Frame / Map are the plain
// stand-ins of oracle/ref_shim), linked with the product's MapPointBatch.cc -- whose strong MapPoint::ComputeDistinctiveDescriptors replaces the
// reference's body -- and libygzf.  The reference's own body stays callable under another name (a renamed, all-weak copy of the object), so the
// three forms run on the SAME MapPoint objects:
//     reference body (CPU)   |   product member, one point per call (device)   |   ygz::ComputeDistinctiveDescriptorsBatch (ONE device call)
#include "../../orb_ygz_slam_amd/csrc/host/ORBextractor.h"   // the product's (same include guard as the reference's header: first one wins)
#define private public
#define protected public
#include "MapPoint.h"
#undef private
#undef protected
#include "MapPointBatch.h"

#include <cstdint>
#include <cstdio>
#include <string>
#include <cstring>
#include <memory>
#include <vector>

#include "ygzf_pool.h"

extern "C" void ygz_ref_MapPoint_ComputeDistinctiveDescriptors(ygz::MapPoint *);
namespace ygz {
int ORBextractor::sDevice = 0;
// the reference's src/ORBmatcher.cc (in the link for MapPoint.cc's DescriptorDistance) names it; nothing here reaches it
std::vector<size_t> Frame::GetFeaturesInArea(const float &, const float &, const float &, const int, const int) const { yr_unsupported("Frame::GetFeaturesInArea"); }
}

extern "C" {
// points p = 0 .. n_points-1 with the observation descriptors desc[obs_off[p] .. obs_off[p+1]) (one stand-in KeyFrame per observation, bad[...] marks
// KeyFrames that are bad).  out[3 * p + {0, 1, 2}] = the observation whose descriptor the MapPoint holds after the reference body / the product
// member called per point / the batch front end (index among ALL its observations, first one carrying those 32 bytes; -1: descriptor untouched).
// Returns the number of shell failures reported during the call.
int bm_distinctive(int n_points, const int *obs_off, const uint8_t *desc, const uint8_t *bad, int *out) {
    using namespace ygz;
    const unsigned long f0 = ygzf_host::failure_count();
    Map map;
    const int total = obs_off[n_points];
    std::vector<KeyFrame> kfs(total);
    for (int i = 0; i < total; i++) {
        kfs[i].mDescriptors.create(1, 32, CV_8UC1);
        std::memcpy(kfs[i].mDescriptors.data, desc + 32 * (size_t) i, 32);
        kfs[i].mvuRight.assign(1, -1.f);
        kfs[i].mvKeys.resize(1);
        kfs[i].mvScaleFactors.assign(1, 1.f);
        kfs[i].mnScaleLevels = 1;
        kfs[i].mbBad = bad && bad[i];
    }
    KeyFrame anchor;                       // reference KeyFrame of points without observations
    anchor.mvKeys.resize(1);
    anchor.mvScaleFactors.assign(1, 1.f);
    anchor.mnScaleLevels = 1;
    std::vector<std::unique_ptr<MapPoint>> pts;
    for (int p = 0; p < n_points; p++) {
        const int n = obs_off[p + 1] - obs_off[p];
        pts.emplace_back(new MapPoint(Vector3f(0, 0, 1), n ? &kfs[obs_off[p]] : &anchor, &map));
        for (int i = 0; i < n; i++) pts.back()->AddObservation(&kfs[obs_off[p] + i], 0);
    }
    auto winner = [&](int p) {
        const cv::Mat d = pts[p]->GetDescriptor();
        if (d.empty()) return -1;
        for (int i = obs_off[p]; i < obs_off[p + 1]; i++)
            if (std::memcmp(d.data, desc + 32 * (size_t) i, 32) == 0) return i - obs_off[p];
        return -2;
    };
    auto reset = [&]() { for (auto &mp : pts) mp->mDescriptor = cv::Mat(); };
    for (int p = 0; p < n_points; p++) ygz_ref_MapPoint_ComputeDistinctiveDescriptors(pts[p].get());
    for (int p = 0; p < n_points; p++) out[3 * p] = winner(p);
    reset();
    for (int p = 0; p < n_points; p++) pts[p]->ComputeDistinctiveDescriptors();          // the product's strong member, one device call per point
    for (int p = 0; p < n_points; p++) out[3 * p + 1] = winner(p);
    reset();
    std::vector<MapPoint *> raw;
    for (auto &mp : pts) raw.push_back(mp.get());
    raw.push_back(nullptr);                                                             // tolerated
    if (n_points > 1) raw.push_back(pts[0].get());                                      // a point listed twice
    ComputeDistinctiveDescriptorsBatch(raw);
    for (int p = 0; p < n_points; p++) out[3 * p + 2] = winner(p);
    return (int) (ygzf_host::failure_count() - f0);
}

// The shells have no error channel of their own (the reference's signatures return void / counts): a failing device call must at least be
// observable.  Points the shells at a device that does not exist, runs the batch front end and the member, and reports what the failure channel
// saw: returns the number of failures counted, copies the last message, and counts the callback's invocations into *callbacks.
static void on_failure(const char *, const char *, void *user) { ++*(int *) user; }
int bm_provoke_failure(char *last, int cap, int *callbacks) {
    using namespace ygz;
    const unsigned long f0 = ygzf_host::failure_count();
    const int keep = ORBextractor::sDevice;
    ORBextractor::sDevice = 1000;                    // no such device: the context pool cannot create a context
    *callbacks = 0;
    ygzf_host::set_failure_callback(on_failure, callbacks);
    Map map;
    KeyFrame kf[2];
    for (auto &k : kf) {
        k.mDescriptors.create(1, 32, CV_8UC1);
        std::memset(k.mDescriptors.data, 0x5A, 32);
        k.mvuRight.assign(1, -1.f); k.mvKeys.resize(1); k.mvScaleFactors.assign(1, 1.f); k.mnScaleLevels = 1;
    }
    MapPoint mp(Vector3f(0, 0, 1), &kf[0], &map);
    mp.AddObservation(&kf[0], 0);
    mp.AddObservation(&kf[1], 0);
    std::vector<MapPoint *> pts(1, &mp);
    ComputeDistinctiveDescriptorsBatch(pts);         // batch call fails, then the member's one-point call fails: the descriptor stays empty
    const bool untouched = mp.mDescriptor.empty();
    ygzf_host::set_failure_callback(nullptr, nullptr);
    ORBextractor::sDevice = keep;
    const std::string msg = ygzf_host::last_failure();
    std::snprintf(last, cap, "%s", msg.c_str());
    return untouched ? (int) (ygzf_host::failure_count() - f0) : -1;
}
}
