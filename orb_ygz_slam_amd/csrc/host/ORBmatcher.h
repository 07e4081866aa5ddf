// ORBmatcher.h -- the Tracking-called part of ygz::ORBmatcher (reference include/ORBmatcher.h:38-149) over libygzf.
#ifndef YGZF_HOST_ORBMATCHER_H
#define YGZF_HOST_ORBMATCHER_H
#include <set>
#include <vector>

#include "ygz_compat.h"

struct ygzf_ctx;

namespace ygz {
class ORBmatcher {
public:
    ORBmatcher(float nnratio = 0.6, bool checkOri = true);

    // Hamming distance between two 256-bit ORB descriptors (static, called all over the reference).
    static int DescriptorDistance(const cv::Mat &a, const cv::Mat &b);

    // Project MapPoints tracked in the last frame into the current frame and search matches (TrackWithMotionModel).
    int SearchByProjection(Frame &CurrentFrame, const Frame &LastFrame, const float th, const bool bMono, bool checkLevel = true);

    // Search matches between Frame keypoints and projected MapPoints (Tracking::SearchLocalPoints).
    int SearchByProjection(Frame &F, const std::vector<MapPoint *> &vpMapPoints, const float th = 3, bool checkLevel = true);

    // Project MapPoints seen in a KeyFrame into the current frame and search matches (Tracking::Relocalization).
    int SearchByProjection(Frame &CurrentFrame, KeyFrame *pKF, const std::set<MapPoint *> &sAlreadyFound, const float th, const int ORBdist);

    // Brute force inside equal vocabulary nodes between KeyFrame MapPoints and Frame keypoints (TrackReferenceKeyFrame, Relocalization).
    int SearchByBoW(KeyFrame *pKF, Frame &F, std::vector<MapPoint *> &vpMapPointMatches);

    // Matching for the map initialisation (monocular only).
    int SearchForInitialization(Frame &F1, Frame &F2, std::vector<cv::Point2f> &vbPrevMatched, std::vector<int> &vnMatches12, int windowSize = 10);

    static const int TH_LOW;
    static const int TH_HIGH;
    static const int HISTO_LENGTH;

    // Matcher objects are created on the stack per call from several threads: contexts come from a small
    // thread-safe pool and are handed back by the destructor of the lease.
    static int sDevice;

protected:
    float mfNNratio;
    bool mbCheckOrientation;
};
}  // namespace ygz
#endif
