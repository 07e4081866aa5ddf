// standalone/ORBmatcher.h -- ygz::ORBmatcher's public interface (reference include/ORBmatcher.h:38-149) for builds WITHOUT the reference
// tree (this repository's tests: no OpenCV / Eigen installed).  Inside the reference tree this file is not used: ORBmatcher.cc is compiled
// against the reference's own, unchanged include/ORBmatcher.h (found first on the include path) and defines the members listed under
// "hot path" below; the remaining members keep their reference bodies (INTEGRATION.md shows the link recipe).
#ifndef ORBMATCHER_H
#define ORBMATCHER_H
#include <set>
#include <utility>
#include <vector>

#include "ygz_compat.h"

namespace ygz {
const int WarpHalfPatchSize = 4;
const int WarpPatchSize = 8;

class ORBmatcher {
public:
    ORBmatcher(float nnratio = 0.6, bool checkOri = true);

    // ---- hot path: defined by orb_ygz_slam_amd/csrc/host/ORBmatcher.cc over libygzf --------------------------------------------
    // Hamming distance between two 256-bit ORB descriptors (static, called all over the reference).
    static int DescriptorDistance(const cv::Mat &a, const cv::Mat &b);
    // Search matches between Frame keypoints and projected MapPoints (Tracking::SearchLocalPoints).
    int SearchByProjection(Frame &F, const std::vector<MapPoint *> &vpMapPoints, const float th = 3, bool checkLevel = true);
    // Project MapPoints tracked in the last frame into the current frame and search matches (TrackWithMotionModel).
    int SearchByProjection(Frame &CurrentFrame, const Frame &LastFrame, const float th, const bool bMono, bool checkLevel = true);
    // Project MapPoints seen in a KeyFrame into the current frame and search matches (Tracking::Relocalization).
    int SearchByProjection(Frame &CurrentFrame, KeyFrame *pKF, const std::set<MapPoint *> &sAlreadyFound, const float th, const int ORBdist);
    // Brute force inside equal vocabulary nodes between KeyFrame MapPoints and Frame keypoints (TrackReferenceKeyFrame, Relocalization).
    int SearchByBoW(KeyFrame *pKF, Frame &F, std::vector<MapPoint *> &vpMapPointMatches);
    // Matching for the map initialisation (monocular only).
    int SearchForInitialization(Frame &F1, Frame &F2, std::vector<cv::Point2f> &vbPrevMatched, std::vector<int> &vnMatches12, int windowSize = 10);
    // Matching to triangulate new MapPoints; checks the epipolar constraint (LocalMapping::CreateNewMapPoints).
    int SearchForTriangulation(KeyFrame *pKF1, KeyFrame *pKF2, Matrix3f &F12, std::vector<std::pair<size_t, size_t>> &vMatchedPairs, const bool bOnlyStereo);
    // Direct (photometric) projection of a MapPoint observed in `ref` into `curr` (Tracking::SearchLocalPointsDirect): affine warp of the
    // reference patch + Align2D.  px_curr: initial guess in, refined pixel out.
    bool FindDirectProjection(KeyFrame *ref, Frame *curr, MapPoint *mp, Vector2f &px_curr, int &search_level);

    // ---- outside the hot path (LocalMapping / LoopClosing threads): declared for interface parity, bodies stay the reference's --------
    int SearchByProjection(KeyFrame *pKF, cv::Mat Scw, const std::vector<MapPoint *> &vpPoints, std::vector<MapPoint *> &vpMatched, int th);
    int SearchByBoW(KeyFrame *pKF1, KeyFrame *pKF2, std::vector<MapPoint *> &vpMatches12);
    int SearchBySim3(KeyFrame *pKF1, KeyFrame *pKF2, std::vector<MapPoint *> &vpMatches12, const float &s12, const cv::Mat &R12, const cv::Mat &t12,
                     const float th);
    int Fuse(KeyFrame *pKF, const std::vector<MapPoint *> &vpMapPoints, const float th = 3.0);
    int Fuse(KeyFrame *pKF, cv::Mat Scw, const std::vector<MapPoint *> &vpPoints, float th, std::vector<MapPoint *> &vpReplacePoint);

public:
    static const int TH_LOW;
    static const int TH_HIGH;
    static const int HISTO_LENGTH;

private:
    float mfNNratio;
    bool mbCheckOrientation;
};
}  // namespace ygz
#endif
