// ygz_oracle.h -- CPU ORACLE (test infrastructure, NOT product code).
//
// A dependency-free C++17 restatement of the per-frame hot path of gaoxiang12/ORB-YGZ-SLAM:
//   ORBextractor  (src/ORBextractor.cc)      pyramid, FAST-9 cells, octree, IC angle, 7x7 blur, rBRIEF
//   ORBmatcher    (src/ORBmatcher.cc)        Hamming distance + SearchByProjection / SearchForInitialization
//   Frame grid    (src/Frame.cc:314-330,424-493)
//   SparseImgAlign(src/SparseImageAlign.cc)  + NLLSSolver Gauss-Newton + Sophus SE3f
// plus restatements of the OpenCV 2.4.11/3.2 primitives the reference calls (resize INTER_LINEAR, FAST,
// GaussianBlur, fastAtan2, cvRound), which are NOT under /root/reference.
//
// PARITY STATUS: "parity unpinned" for the OpenCV primitives (oracle_cvprims.cpp) and the Eigen / Sophus algebra (the reference ships no test or golden vector for them, OpenCV is neither vendored nor version-pinned, and those sources
// need Eigen / Sophus / the whole Frame-MapPoint graph).  PINNED: the extractor (oracle_extractor.cpp) against the
// reference's own src/ORBextractor.cc, the matcher's search functions and the direct projection against its src/ORBmatcher.cc +
// src/Align.cc, the sparse image aligner against its src/SparseImageAlign.cc + NLSSolver, the Frame grid / isInFrustum /
// ComputeStereoMatches against its src/Frame.cc, ComputeDistinctiveDescriptors / PredictScale against its src/MapPoint.cc, all compiled where they lie
// over oracle/ref_shim/ (tests/test_ref_*.py), and FAST-10 against
// the reference's own libfast incl. Thirdparty/fast's 167-corner KAT (tests/test_oracle_fast10.py); both live in oracle/_ref.
//
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may link or call this code.
#ifndef YGZ_ORACLE_H
#define YGZ_ORACLE_H
#include <cstddef>
#include <cstdint>
#include <vector>

namespace ygzo {

// Layout-compatible with cv::KeyPoint (7 x 4 bytes).
struct KeyPoint {
    float x, y, size, angle, response;
    int octave, class_id;
};

struct Image {  // tight 8-bit image (step == w), like Frame::mvImagePyramid clones (src/Frame.cc:810-813)
    int w = 0, h = 0;
    std::vector<uint8_t> d;
    Image() {}
    Image(int w_, int h_) : w(w_), h(h_), d((size_t) w_ * h_) {}
    inline uint8_t at(int y, int x) const { return d[(size_t) y * w + x]; }
};

// ---- OpenCV primitive restatements (all "recalled", defined here) ----------------------
int cv_round(double v);                                    // B6: round half to even
float fast_atan2_deg(float y, float x);                    // B5
void resize_linear_u8(const Image &src, Image &dst);       // B1 (dst.w/dst.h preset)
// B4, REFLECT_101.  Three recalled OpenCV definitions (see oracle_cvprims.cpp): the process-wide mode is what the extractor uses.
enum { CV_MODE_LEGACY_SSE2 = 0, CV_MODE_LEGACY_INT = 1, CV_MODE_CV4 = 2 };
void set_cv_mode(int mode);
int get_cv_mode();
void gaussian_kernel7_s2(int mode, int kq[7]);
void gaussian_blur7_s2_u8(const Image &src, Image &dst);             // current mode
void gaussian_blur7_s2_u8(const Image &src, Image &dst, int mode);
// cosf/sinf of (angle_deg * (float)(pi/180)): "correctly rounded float of the double result", computed with a
// fixed double polynomial (no FMA) so that host and device can evaluate the identical operation sequence.
void sincos_deg(float angle_deg, float *c, float *s);

// cv::FAST(img_window, kps, threshold, nonmax) TYPE_9_16 on the window [x0,x0+w)x[y0,y0+h) of `img`.
// Output coordinates are relative to the window origin, raster order (B3).
struct FastPt { int x, y, score; };
void fast9(const uint8_t *img, int stride, int w, int h, int threshold, bool nonmax, std::vector<FastPt> &out);

// ---- ORBextractor ------------------------------------------------------------------------------------------
class Extractor {
public:
    // src/ORBextractor.cc:412-470
    Extractor(int nfeatures, float scaleFactor, int nlevels, int iniThFAST, int minThFAST);

    int nfeatures, nlevels, iniThFAST, minThFAST;
    float scaleFactor;
    std::vector<float> mvScaleFactor, mvInvScaleFactor, mvLevelSigma2, mvInvLevelSigma2;
    std::vector<int> mnFeaturesPerLevel, umax;
    std::vector<Image> mvImagePyramid;

    void LevelSize(int w, int h, int level, int *lw, int *lh) const;    // :1131-1132
    void ComputePyramid(const uint8_t *img, int w, int h, int stride);  // :1129-1150
    // FAST candidates of one level in region coordinates (origin = (16,16)), cell-major / raster order: the
    // vToDistributeKeys vector of :747-781.
    void CellCandidates(int level, std::vector<KeyPoint> &out) const;
    void ComputeKeyPointsOctTree(std::vector<std::vector<KeyPoint>> &all) const;  // :725-804
    // :533-723.  Tie-break of the (size, pointer) sort is DEFINED as (size, creation sequence number).
    std::vector<KeyPoint> DistributeOctTree(const std::vector<KeyPoint> &keys, int minX, int maxX, int minY,
                                            int maxY, int N) const;
    float ICAngle(const Image &img, float ptx, float pty) const;  // :77-101
    void ComputeDescriptor(const KeyPoint &kp, const Image &blurred, uint8_t *desc) const;  // :105-149
    // ShiTomasiScore :1152-1187 (8x8 box; all sums are exact integers in float)
    float ShiTomasiScore(const Image &img, int u, int v) const;
    // ComputeKeyPointsDSOSingleLevel :1275-1386.  mnGridSize persists across frames (-1 = not yet set).  Ties of the
    // per-cell score sort (std::sort, unspecified order) are DEFINED as raster order.  exist_kps: angles are recomputed.
    int mnGridSize = -1;
    void ComputeKeyPointsDSOSingleLevel(std::vector<KeyPoint> &allKeypoints, std::vector<KeyPoint> &exist_kps);
    // operator()(Frame*, vector<KeyPoint>&, OutputArray, DSO_KEYPOINT, leftEye = true) :1031-1127 on a frame whose pyramid is
    // computed from `img`: keys = frame->mvKeys (in: the N existing keys, out: + the new ones), desc = N_total x 32.
    void ExtractDSO(const uint8_t *img, int w, int h, int stride, std::vector<KeyPoint> &keys, std::vector<uint8_t> &desc);
    // ComputeKeyPointsFast :1189-1273 (the FAST_KEYPOINT branch of the Frame overload, :1045-1051): per level FAST-10 (libfast, barrier
    // iniThFAST) on the image minus a 20-px top / left margin, score, >= non-maximum suppression; per 5x5-px cell of a level-0 grid the
    // corner with the largest Shi-Tomasi score over all levels (first one on ties: level order, then libfast's order), cells holding an
    // existing key excluded; IC_Angle; existing keys re-oriented.  DEFINED where the reference indexes out of bounds (its author: "has a
    // bug which may corrupt the program", :1191): a key or corner whose cell index gy * cols + gx falls outside the grid is ignored
    // (inside the grid the index is used as computed, row wrap included); levels narrower than 42 px or lower than 27 px are skipped (the
    // SSE2 detector's plain fallback would read past the image rows).
    void ComputeKeyPointsFast(std::vector<std::vector<KeyPoint>> &allKeypoints, std::vector<KeyPoint> &exist_kps) const;
    // operator()(Frame*, ..., FAST_KEYPOINT, leftEye = true)
    void ExtractFast(const uint8_t *img, int w, int h, int stride, std::vector<KeyPoint> &keys, std::vector<uint8_t> &desc);
    // ComputeKeyPointsDSO :1388-1507 (the multi-level grid detector; its call in the Frame overload is commented out, :1053): per level a
    // grid of sqrt(h w / n_level) px, libfast at iniThFAST then minThFAST per inner cell, 20-px edge filter, occupancy test on the
    // LEVEL-0-sized map indexed with LEVEL coordinates (:1466 -- kept), the 2 best Shi-Tomasi corners per cell, every selected corner
    // marks the map at once and stays marked through the retry passes (grid - 5 while fewer than n_level corners, down to 7) and the
    // later levels.  Ties of the score sort: raster order; a pass without a single corner ends the level (the reference would spin).
    // mnGridSize is left at the last level's final value.
    void ComputeKeyPointsDSO(std::vector<std::vector<KeyPoint>> &allKeypoints, std::vector<KeyPoint> &exist_kps);
    // the Frame overload's descriptor / scaling / concatenation part (:1063-1126) over ComputeKeyPointsDSO's keypoints
    void ExtractDSOMultiLevel(const uint8_t *img, int w, int h, int stride, std::vector<KeyPoint> &keys, std::vector<uint8_t> &desc);
    // operator()(InputArray, InputArray, vector<KeyPoint>&, OutputArray)  :970-1028
    void Extract(const uint8_t *img, int w, int h, int stride, std::vector<KeyPoint> &kps,
                 std::vector<uint8_t> &desc);
};

// ---- ORBmatcher / Frame grid -------------------------------------------------------------------------------
int descriptor_distance(const uint8_t *a, const uint8_t *b);  // src/ORBmatcher.cc:1507-1523

// The slice of `Frame` the matcher reads (src/Frame.h), as plain arrays.
struct FrameView {
    int N = 0;
    const KeyPoint *keys = nullptr;       // mvKeys
    const uint8_t *desc = nullptr;        // mDescriptors, N x 32
    const float *uRight = nullptr;        // mvuRight (may be null => all -1)
    float minX = 0, minY = 0, maxX = 0, maxY = 0;  // mnMinX ... (image bounds)
    float gridInvW = 0, gridInvH = 0;     // mfGridElementWidthInv/HeightInv = 64/(maxX-minX), 48/(maxY-minY)
    float fx = 0, fy = 0, cx = 0, cy = 0, mb = 0, mbf = 0;
    const float *scaleFactors = nullptr;  // mvScaleFactors
    int nlevels = 0;
};

struct Grid {  // Frame::mGrid[64][48]
    static const int COLS = 64, ROWS = 48;
    std::vector<int> cell[COLS][ROWS];
    void Assign(const FrameView &f);  // src/Frame.cc:314-330 + PosInGrid :483-493
    // src/Frame.cc:424-481
    void FeaturesInArea(const FrameView &f, float x, float y, float r, int minLevel, int maxLevel,
                        std::vector<int> &out) const;
};

// SearchByProjection(Frame &Cur, const Frame &Last, th, bMono, checkLevel)  src/ORBmatcher.cc:1218-1350
// Last-frame side as arrays: per keypoint i: has MapPoint (mp_valid), outlier flag, world pos, MP descriptor,
// MP->Observations()>0.  cur_owner[i2]: 0 = Cur.mvpMapPoints[i2]==NULL, 1 = non-null with Observations()==0,
// 2 = non-null with Observations()>0.  On return cur_match[i2] = index i of the Last keypoint whose MapPoint
// was written to Cur.mvpMapPoints[i2], or -1 (untouched) / -2 (written then nulled by the rotation check).
struct ProjLastInput {
    int N = 0;
    const KeyPoint *keys = nullptr;
    const uint8_t *mp_valid = nullptr, *outlier = nullptr, *mp_has_obs = nullptr;
    const float *mp_world = nullptr;  // N x 3
    const uint8_t *mp_desc = nullptr; // N x 32
    float Rcw[9], tcw[3];             // CurrentFrame.mTcw
    float Rlw[9], tlw[3];             // LastFrame.mTcw
};
int search_by_projection_last(const FrameView &cur, const Grid &grid, const ProjLastInput &last, float th,
                              bool bMono, bool checkLevel, bool checkOrientation, uint8_t *cur_owner,
                              int *cur_match);

// SearchByProjection(Frame &F, const vector<MapPoint*>&, th, checkLevel)  src/ORBmatcher.cc:43-126
struct ProjMapPointsInput {
    int M = 0;
    const uint8_t *track_in_view = nullptr, *bad = nullptr, *mp_has_obs = nullptr;
    const float *projX = nullptr, *projY = nullptr, *projXR = nullptr, *viewCos = nullptr;
    const int *scaleLevel = nullptr;
    const uint8_t *mp_desc = nullptr;  // M x 32
};
int search_by_projection_mappoints(const FrameView &F, const Grid &grid, const ProjMapPointsInput &in, float th,
                                   bool checkLevel, float nnratio, uint8_t *owner, int *match);

// SearchByProjection(Frame &Cur, KeyFrame *pKF, const set<MapPoint*> &sAlreadyFound, th, ORBdist)  src/ORBmatcher.cc:1352-1469
// KeyFrame side as arrays over pKF->GetMapPointMatches(): usable[i] = pMP && !pMP->isBad() && !sAlreadyFound.count(pMP);
// maxDist/minDist = GetMax/MinDistanceInvariance(); mfMaxDistance feeds MapPoint::PredictScale (src/MapPoint.cc:359-373).
// cur_owner[i2] != 0 <=> Cur.mvpMapPoints[i2] != NULL (any MapPoint blocks, :1419).  The scalar prologue (:1371-1400) is exposed
// through out_* (may be null): validity after all gates, projection, predicted level.
struct ProjKFInput {
    int M = 0;
    const uint8_t *usable = nullptr;
    const float *world = nullptr;      // M x 3
    const float *maxDistInv = nullptr, *minDistInv = nullptr, *mfMaxDistance = nullptr;
    const float *kf_angle = nullptr;   // pKF->mvKeys[i].angle
    const uint8_t *mp_desc = nullptr;  // M x 32
    float Rcw[9], tcw[3];
    float logScaleFactor = 0;          // Frame::mfLogScaleFactor
    int nScaleLevels = 0;              // Frame::mnScaleLevels
};
int search_by_projection_kf(const FrameView &cur, const Grid &grid, const ProjKFInput &in, float th, int ORBdist,
                            bool checkOrientation, uint8_t *cur_owner, int *cur_match, uint8_t *out_valid, float *out_u,
                            float *out_v, int *out_level);

// Frame::isInFrustum(MapPoint*, viewingCosLimit)  src/Frame.cc:363-422 with MapPoint::PredictScale src/MapPoint.cc:359-373, for M points.
// Rcw/tcw = mRcw/mtcw, Ow = mOw (camera centre); out_* = the fields the function writes into the MapPoint (valid where in_view).
struct FrustumInput {
    int M = 0;
    const float *world = nullptr, *normal = nullptr;            // M x 3: GetWorldPos(), GetNormal()
    const float *maxDistInv = nullptr, *minDistInv = nullptr;   // GetMax/MinDistanceInvariance()
    const float *mfMaxDistance = nullptr;                       // PredictScale's numerator
    float Rcw[9], tcw[3], Ow[3];
    float logScaleFactor = 0;
    int nScaleLevels = 0;
};
void is_in_frustum(const FrameView &F, const FrustumInput &in, float viewingCosLimit, uint8_t *in_view, float *projX, float *projY,
                   float *projXR, int *level, float *viewCos);

// MapPoint::ComputeDistinctiveDescriptors  src/MapPoint.cc:211-271: index of the observation descriptor with the least median
// Hamming distance to the others (desc: N x 32); N >= 1.
int distinctive_descriptor(const uint8_t *desc, int N);

// SearchForInitialization  src/ORBmatcher.cc:375-478
int search_for_initialization(const FrameView &F1, const FrameView &F2, const Grid &grid2, float *prevMatchedXY,
                              int windowSize, float nnratio, bool checkOrientation, int *matches12);

// SearchByBoW(KeyFrame *pKF, Frame &F, vector<MapPoint*> &vpMapPointMatches)  src/ORBmatcher.cc:155-263.
// The FeatureVector merge-join (:169-247: equal node ids, lower_bound skips) is the caller's; the oracle gets the joined node list:
// node k pairs KeyFrame feature indices kf_idx[kf_off[k]..kf_off[k+1]) with Frame feature indices f_idx[f_off[k]..f_off[k+1]).
// kf_valid[i] = vpMapPointsKF[i] && !isBad().  match[iF] = KeyFrame feature index whose MapPoint lands in vpMapPointMatches[iF],
// -1 none, -2 culled by the rotation check.
int search_by_bow(int nNodes, const int *kf_off, const int *kf_idx, const int *f_off, const int *f_idx, const uint8_t *kf_valid,
                  const KeyPoint *kf_keys, const uint8_t *kf_desc, int nF, const KeyPoint *f_keys, const uint8_t *f_desc, float nnratio,
                  bool checkOrientation, int *match);

// SearchForTriangulation(KeyFrame *pKF1, KeyFrame *pKF2, Matrix3f &F12, vector<pair<size_t,size_t>> &vMatchedPairs, bool bOnlyStereo)
// src/ORBmatcher.cc:596-741 with CheckDistEpipolarLine :136-153 (LocalMapping::CreateNewMapPoints' matcher).  Joined node list as in
// search_by_bow: node k pairs KF1 features idx1[off1[k]..off1[k+1]) with KF2 features idx2[off2[k]..off2[k+1]).  has_mp[i] = the
// KeyFrame already holds a MapPoint in slot i (GetMapPoint(i) != NULL).  match12[i1] = matched KF2 feature, -1 none, -2 culled by the
// rotation check; vMatchedPairs = the (i1, match12[i1] >= 0) pairs in ascending i1 (:731-736).
struct TriangulationInput {
    int nNodes;
    const int *off1, *idx1, *off2, *idx2;
    int n1;
    const KeyPoint *keys1;
    const uint8_t *desc1, *has_mp1;
    const float *uRight1;        // mvuRight of KF1, or nullptr (monocular: all -1)
    int n2;
    const KeyPoint *keys2;
    const uint8_t *desc2, *has_mp2;
    const float *uRight2;
    const float *scaleFactors2, *levelSigma2_2;   // pKF2->mvScaleFactors, pKF2->mvLevelSigma2
    const float *F12;            // row-major 3 x 3
    const float *Cw1, *R2w, *t2w;   // pKF1->GetCameraCenter(), pKF2->GetRotation() (row-major), pKF2->GetTranslation()
    float fx2, fy2, cx2, cy2;
};
int search_for_triangulation(const TriangulationInput &in, bool onlyStereo, bool checkOrientation, int *match12);

// ---- Sophus SE3f + SparseImgAlign --------------------------------------------------------------------------
struct SE3f {
    float q[4] = {0, 0, 0, 1};  // x,y,z,w (Eigen coeffs order)
    float t[3] = {0, 0, 0};
    static SE3f FromRt(const float R[9], const float t[3]);
    static SE3f Exp(const float a[6]);   // se3.hpp:406-428, so3.hpp:425-456
    SE3f Inverse() const;                // se3.hpp:168-172
    SE3f Mul(const SE3f &o) const;       // operator* = fastMultiply + normalize (se3.hpp:159-163,267-271)
    void Act(const float p[3], float out[3]) const;  // so3*p + t (Eigen _transformVector)
    void RotationMatrix(float R[9]) const;
};

struct AlignFrame {  // slice of Frame read by SparseImgAlign
    int N = 0;
    const KeyPoint *keys = nullptr;
    const uint8_t *mp_valid = nullptr;   // mvpMapPoints[i] != nullptr && !isBad()
    const uint8_t *outlier = nullptr;    // mvbOutlier
    const float *mp_world = nullptr;     // N x 3
    SE3f Tcw;
    std::vector<const Image *> pyramid;  // mvImagePyramid (step == cols)
    const float *invScaleFactors = nullptr;
    float fx = 0, fy = 0, cx = 0, cy = 0;
};

struct AlignResult {
    SE3f TCR;
    size_t ret = 0;        // n_meas_/16
    int iters_total = 0;   // number of computeResiduals(linearize) calls, for reporting
    float chi2 = 0;
    float H[36];
};
bool ldlt_solve6(const float H[36], const float b[6], float x[6]);   // x = H.ldlt().solve(b): pivoted LDL^T in float (oracle_align.cpp)
// SparseImgAlign(max_level, min_level, n_iter=10, GaussNewton).run(ref, cur, TCR)  src/SparseImageAlign.cc:20-49
// device_order: the normal equations evaluated with the HIP kernel's formulation, fused multiply-adds and reduction tree (oracle_align.cpp) --
// the mode the kernel is compared with BIT FOR BIT; false = the reference's own pixel-by-pixel order
AlignResult sparse_img_align(const AlignFrame &ref, const AlignFrame &cur, int max_level, int min_level,
                             int n_iter, bool device_order = false);
// the same algorithm with every quantity in double (oracle_align.cpp): the third party of the aligner's tolerance argument
struct AlignResultF64 {
    double T[7] = {0, 0, 0, 1, 0, 0, 0};   // qx qy qz qw tx ty tz of T_cur_from_ref
    size_t ret = 0;
    int iters_total = 0;
    double chi2 = 0;
};
AlignResultF64 sparse_img_align_f64(const AlignFrame &ref, const AlignFrame &cur, int max_level, int min_level, int n_iter);

// ---- ORBmatcher::FindDirectProjection + Align2D  src/ORBmatcher.cc:1525-1602, src/Align.cc:8-104 -------------------------------------
struct DirectRef {   // the reference KeyFrame's slice
    KeyPoint kp;                    // ref->mvKeys[mp->GetObservations()[ref]]
    SE3f Tcw;                       // ref->GetPose()
    const Image *level_img = nullptr;   // ref->mvImagePyramid[kp.octave]
    const float *scaleFactors = nullptr;
    float invLevelSigma2_1 = 0;     // ref->mvInvLevelSigma2[1]
    int nlevels = 0;                // ref->mvImagePyramid.size()
    float fx = 0, fy = 0, cx = 0, cy = 0;
};
struct DirectCur {
    SE3f Tcw;
    std::vector<const Image *> pyramid;
    const float *scaleFactors = nullptr, *invScaleFactors = nullptr;
    float fx = 0, fy = 0, cx = 0, cy = 0;
};
void inverse3(const float m[9], float r[9]);   // Matrix3f::inverse() as Eigen's compute_inverse_size3 (cofactors; oracle_direct.cpp)
bool align2d(const Image &cur_img, const uint8_t *ref_patch_with_border, const uint8_t *ref_patch, int n_iter, float cur_px_estimate[2]);
bool find_direct_projection(const DirectRef &ref, const DirectCur &cur, const float mp_world[3], float px_curr[2], int *search_level,
                            uint8_t *patch_with_border_out);

// ---- Frame::ComputeStereoMatches  src/Frame.cc:509-682 -------------------------------------------------------------------------
// Left/right keys + descriptors, both extractors' pyramids (tight levels), mvScaleFactors / mvInvScaleFactors, mb, mbf.
// Out: mvuRight / mvDepth (N floats each, -1 = no match).  Defined where the reference is not: an empty match list skips the median
// cut (the reference reads vDistIdx[0] of an empty vector); right keys whose row band leaves the image are clipped to it.
void compute_stereo_matches(int N, const KeyPoint *keysL, const uint8_t *descL, int Nr, const KeyPoint *keysR, const uint8_t *descR,
                            const std::vector<const Image *> &pyrL, const std::vector<const Image *> &pyrR, const float *scaleFactors,
                            const float *invScaleFactors, float mb, float mbf, float *uRight, float *depth);

}  // namespace ygzo
#endif
