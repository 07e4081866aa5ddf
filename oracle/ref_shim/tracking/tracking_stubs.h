// oracle/ref_shim/tracking/tracking_stubs.h -- TEST INFRASTRUCTURE (boundary build only, tests/cpp/build_boundary.sh): what the reference's
// src/Tracking.cc needs from the classes OUTSIDE the hot path -- System, Viewer, FrameDrawer, MapDrawer, Initializer, Optimizer, PnPsolver,
// LocalMapping, LoopClosing, KeyFrameDatabase, ConfigParam -- as DECLARATIONS, so that the file compiles where it lies, unchanged, against the
// product's ORBextractor.h and the reference's own ORBmatcher.h / SparseImageAlign.h / Frame.h.  No member has a body here (except the two
// trivial getters the constructor calls): the linker resolves them to one aborting stand-in (tests/cpp/build_boundary.sh generates the alias
// list from the link's own undefined-symbol report).  The test drives only the Tracking members whose callees are the hot path
// (TrackWithSparseAlignment, SearchLocalPoints).  These files shadow the reference headers of the same names through the include order.
#ifndef YGZ_ORACLE_REF_SHIM_TRACKING_STUBS_H
#define YGZ_ORACLE_REF_SHIM_TRACKING_STUBS_H
// the reference headers of the same names are found first when they are included from include/Tracking.h (a quoted include searches the
// including file's own directory): switched off through their include guards
#define YGZ_MAPDRAWER_H_
#define YGZ_KEYFRAMEDATABASE_H
#define INITIALIZER_H
#define YGZ_SYSTEM_H_
#define YGZ_VIEWER_H_
#define YGZ_FRAMEDRAWER_H_
#define YGZ_OPTIMIZER_H_
#define YGZ_PNPSOLVER_H_
#define YGZ_LOCALMAPPING_H_
#define YGZ_LOOPCLOSING_H_
#include <unistd.h>

#include <list>
#include <mutex>
#include <set>
#include <string>
#include <vector>

#include "mini_cv.h"

namespace pangolin {
inline void BindToContext(const std::string &) {}
}

namespace ygz {
class Tracking;
class KeyFrame;
class MapPoint;
class Map;
class Frame;

class System {
public:
    enum eSensor { MONOCULAR = 0, STEREO = 1, RGBD = 2 };
    void Reset();
};

class ConfigParam {
public:
    bool GetUseIMUFlag() const { return false; }     // the boundary test tracks without an IMU
    static SE3d GetSE3Tbc();
    static double GetVINSInitTime();
    double GetImageDelayToIMU() const;
    static int GetLocalWindowSize();
    bool GetRealTimeFlag() const;
};

class Viewer {
public:
    void RequestStop();
    bool isStopped();
    void Release();
    void RequestFinish();
};
class FrameDrawer {
public:
    void Update(Tracking *pTracker);
};
class MapDrawer {
public:
    void SetCurrentCameraPose(const SE3d &Tcw);
    void SetReferenceKeyFrame(KeyFrame *pKF);
};
class LocalMapping {
public:
    bool GetVINSIniting(void);
    bool GetVINSInited(void);
    bool GetFirstVINSInited(void);
    void SetFirstVINSInited(bool flag);
    Vector3d GetGravityVec(void);
    bool GetMapUpdateFlagForTracking();
    void SetMapUpdateFlagInTracking(bool bflag);
    bool GetUpdatingInitPoses(void);
    void InsertKeyFrame(KeyFrame *pKF);
    void RequestReset();
    void Release();
    bool isStopped();
    bool stopRequested();
    bool AcceptKeyFrames();
    void SetAcceptKeyFrames(bool flag);
    bool SetNotStop(bool flag);
    void InterruptBA();
    int KeyframesInQueue();
    KeyFrame *GetMapUpdateKF();
    double GetVINSInitScale(void);
};
class LoopClosing {
public:
    bool GetMapUpdateFlagForTracking();
    void SetMapUpdateFlagInTracking(bool bflag);
    void RequestReset();
    bool isRunningGBA();
    bool isFinishedGBA();
};
class KeyFrameDatabase {
public:
    void clear();
    std::vector<KeyFrame *> DetectRelocalizationCandidates(Frame *F);
};
class Initializer {
public:
    Initializer(const Frame &ReferenceFrame, float sigma = 1.0, int iterations = 200);
    bool Initialize(const Frame &CurrentFrame, const std::vector<int> &vMatches12, Matrix3f &R21, Vector3f &t21, std::vector<Vector3f> &vP3D,
                    std::vector<bool> &vbTriangulated);
};
class Optimizer {
public:
    static void GlobalBundleAdjustemnt(Map *pMap, int nIterations = 5, bool *pbStopFlag = NULL, const unsigned long nLoopKF = 0, const bool bRobust = true);
    static int PoseOptimization(Frame *pFrame);
    static int PoseOptimization(Frame *pFrame, KeyFrame *pLastKF, const IMUPreintegrator &imupreint, const Vector3d &gw, const bool &bComputeMarg = false);
    static int PoseOptimization(Frame *pFrame, Frame *pLastFrame, const IMUPreintegrator &imupreint, const Vector3d &gw, const bool &bComputeMarg = false);
};
class PnPsolver {
public:
    PnPsolver(const Frame &F, const std::vector<MapPoint *> &vpMapPointMatches);
    void SetRansacParameters(double probability = 0.99, int minInliers = 8, int maxIterations = 300, int minSet = 4, float epsilon = 0.4, float th2 = 5.991);
    cv::Mat iterate(int nIterations, bool &bNoMore, std::vector<bool> &vbInliers, int &nInliers);
};
}  // namespace ygz
#endif
