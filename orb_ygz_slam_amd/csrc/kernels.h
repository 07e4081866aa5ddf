// kernels.h -- launcher declarations shared by the HIP translation units of libygzf (product code).
#ifndef YGZF_KERNELS_H
#define YGZF_KERNELS_H
#include "ygzf_internal.h"

namespace ygzf {

hipError_t upload_constants(const int *umax16);

struct PyrTabs {   // device tables of one geometry's pyramid
    const int *xofs;
    const short *xalpha;
    const int *yofs;
    const short *ybeta;
    const PyrColRec *cols;
    const PyrRowRec *rows;
    const PyrTileRec *tiles;
};
void launch_pyr_resize(hipStream_t st, const FrameSet &fs, const LevelGeom *dGeom, const LevelGeom &g, int level, int nFrames, const PyrTabs &T);
// k_pyr_resize_tiled's staging area per fold (a wave's lanes cover `fold` rows x 256 / fold columns per pass): 16-byte chunks per staged
// source row, staged rows, LDS row pitch
constexpr int kPyrFolds = 4;
__host__ __device__ constexpr int pyr_fold_chunks(int foldLog2) { return foldLog2 == 0 ? 21 : foldLog2 == 1 ? 11 : foldLog2 == 2 ? 6 : 3; }
__host__ __device__ constexpr int pyr_fold_pitch(int foldLog2) { return 16 * pyr_fold_chunks(foldLog2); }
__host__ __device__ constexpr int pyr_fold_rows(int foldLog2) { return foldLog2 == 0 ? 44 : foldLog2 == 1 ? 84 : foldLog2 == 2 ? 166 : 330; }
constexpr int kPyrTileRows = 32;   // output rows of a tile at fold 1 (fold f: 32 f rows x 256 / f columns)
void launch_repitch_rows(hipStream_t st, const uint8_t *src, size_t srcPitch, uint8_t *dst, size_t dstPitch, int w, size_t rows);
constexpr int kHostFrameListMax = 128;
struct HostFrameList { unsigned long long addr[kHostFrameListMax]; };   // device-visible addresses of page-locked host frames (by value: kernel argument)
void launch_gather_host_frames(hipStream_t st, const HostFrameList &L, int nFrames, size_t srcPitch, uint8_t *dst, size_t dstPitch, size_t dstFrameStride, int w, int h);
struct PyrChainGraph {
    hipGraph_t graph;
    hipGraphExec_t exec;
    hipGraphNode_t nodes[kMaxLevels];
    int nNodes;                  // nodes 1 .. nNodes
    FrameSet fs;                 // argument storage of the nodes
    const LevelGeom *geom;
    PyrTabs tabs;
    int level[kMaxLevels];
    LevelGeom lv[kMaxLevels];
};
hipError_t pyr_chain_graph_build(PyrChainGraph *pg, const FrameSet &fs, const LevelGeom *dGeom, const LevelGeom *lv, int nlevels, const PyrTabs &T);
hipError_t pyr_chain_graph_retarget(PyrChainGraph *pg, const FrameSet &fs);
void pyr_chain_graph_destroy(PyrChainGraph *pg);
// one frame's whole pyramid chain in one launch (k_pyr_strips): per strip and level, the rows [ca, cb) the strip produces in LDS (level 0:
// stages) and the rows [wa, wb) of them it owns, i.e. writes to the pyramid slab
struct PyrStripPlan {
    uint2 lv[kMaxLevels];   // x = ca | cb << 16, y = wa | wb << 16 (32-bit words: the kernel reads them with scalar loads)
};
struct PyrStripLevel {   // what the kernel needs of a level, 32 bytes
    int w, h, pitch, xtab, ytab, pad;
    long long off;
};
constexpr int kPyrStripMaxThreads = 1024;
__host__ __device__ inline int pyr_strip_lds_pitch(int w) { return ((w + 3) & ~3) + 12; }
// LDS: row table (8 bytes per produced row of the levels >= 1), region A from
// offA (even levels), region B from offB (odd levels)
hipError_t pyr_strips_prepare(size_t ldsBytes);
void launch_pyr_strips(hipStream_t st, const FrameSet &fs, int nlevels, const PyrStripPlan *plans, const PyrStripLevel *levels, int nStrips,
                       int offA, int offB, size_t ldsBytes, int nFrames, const int *xofs, const short *xalpha, const int *yofs, const short *ybeta);
// dst[0 .. bytes) = src[0 .. bytes), bytes a multiple of 16, both 16-byte aligned: small blocks between the page-locked staging area (as the device addresses it) and
// device memory -- a launch where the copy engine would start ~10 us after the copy is queued and hand over ~8 us after it ends
void launch_link_copy(hipStream_t st, void *dst, const void *src, size_t bytes);
void launch_carry_slot(hipStream_t st, ygzf_kp *outKp, uint8_t *outDesc, int *outCnt, long long srcSlot, int kpStride);
void launch_pack_results(hipStream_t st, const int *cnt, const ygzf_kp *kp, const uint8_t *desc, int nFrames, int kpStride, void *dst, size_t offKp, size_t offDesc);
void launch_pack_levels(hipStream_t st, const FrameSet &fs, const LevelGeom *dGeom, int firstLevel, int nlevels, const unsigned *offsets, uint8_t *dst);
size_t fast_quads_lds_bytes(int winPitch, int winRows, int smapRows, int quadCap);
void launch_fast_cells(hipStream_t st, const FrameSet &fs, const LevelGeom *dGeom, int nlevels, int iniTh, int minTh,
                       unsigned short *cellCnt, unsigned *slots, int totalCells, long long totalSlots, int totalGroups, int smapRows,
                       int nFrames, int winPitch, int winRows, int quadCap, const int *groupBaseHost, bool iniFirst, unsigned *stats);
// table-driven form of the same cell loop (k_fast_tab): per-cell records built on the host, window staged by LDS-DMA; cells up to 41 px wide
constexpr int kFastTabMaxCell = 41;
size_t fast_tab_lds_bytes(int winRows, int smapRows, int quadCap);
void launch_fast_tab(hipStream_t st, const FrameSet &fs, const FastCellRec *dCells, int iniTh, int minTh, unsigned short *cellCnt, unsigned *slots,
                     int totalCells, long long totalSlots, int totalGroups, int smapRows, int nFrames, int winRows, int quadCap, bool iniFirst,
                     unsigned *stats);
hipError_t phase_clocks_read(int kernel, unsigned long long *out16, bool reset);   // -DYGZF_PHASE_CLOCK builds only (hipErrorNotSupported otherwise)
constexpr size_t kFastPersistCounterBytes = 64 * 256;
void launch_fast_tab_persist(hipStream_t st, const FrameSet &fs, const FastCellRec *dCells, int iniTh, int minTh, unsigned short *cellCnt, unsigned *slots,
                             int totalCells, long long totalSlots, int totalGroups, int smapRows, int nFrames, int winRows, int quadCap, bool iniFirst,
                             unsigned *stats, unsigned *counters, int nWorkgroups);
constexpr int kFastStatWords = 8 * 64;   // `stats`: 64 x {cells sampled, cells whose keypoints are FAST(minTh)'s, score rounds beyond the first, plan: 1 one pass / 2 iniTh first, corner-bearing quads, quads, -, pass-1 runs}
size_t octree_lds_bytes(int maxCellsPerLevel, int cap, int ldsCand, bool globalNodes);
size_t octree_hist_lds_bytes(int regionInts, int histBins);
hipError_t octree_prepare(size_t ldsBytes, bool globalNodes, bool hist);
void launch_octree(hipStream_t st, const LevelGeom *dGeom, int nlevels, int level0, int nLaunchLevels, const unsigned short *cellCnt, const unsigned *slots,
                   int totalCells, long long totalSlots, unsigned *k0, unsigned *v0, unsigned *k1, unsigned *v1, unsigned *xy,
                   long long candStride, unsigned *lvlKpXY, unsigned char *lvlKpScore, int *lvlKpCnt, int *lvlCandCnt,
                   uint2 *procRec, int kpStride, int cap, int ldsCand, size_t ldsBytes, int nFrames, long long *dbg, int *nodeArena,
                   int regionInts, int histBins, int helpers = 1, int *gHist = nullptr, int *gDone = nullptr, int doneTarget = 0, int spinBudget = 0);
constexpr int kOctDbgWords = 16 * 8 + 16 + 16 * 16;   // per level: six phase stamps, M, n; then per level the workgroups that left the histogram plan; then per level up to 8 (list size, time) pairs of the tree passes
void launch_describe(hipStream_t st, const FrameSet &fs, const LevelGeom *dGeom, int nlevels, const int *lvlKpCnt, int *lvlBase,
                     const uint2 *procRec, int kpStride, ygzf_kp *outKp, uint8_t *outDesc, int *outCnt, int outStride, int nFrames, int cvMode);
void launch_hamming_pairs(hipStream_t st, const void *a, const void *b, int n, int *out);

// ---- matcher (match_kernels.hip) -----------------------------------------------------------------------------------
struct MatchQueryScratch;  // opaque: per-query parameters when they do not fit in LDS
struct MatchArgs {
    // Cur side, pair p uses base + p*kpStrideCur
    const ygzf_kp *curKeys;
    const uint8_t *curDesc;
    const float *curURight;        // nullable (mono): all "-1"
    const int *curCnt;             // count of pair p at curCnt[p*cntStrideCur + cntOffCur]
    long long kpStrideCur;
    int cntStrideCur, cntOffCur;
    const uint8_t *ownerIn;        // nullable: initial Cur.mvpMapPoints state (0 free / 1 owned, 0 obs / 2 owned, obs > 0)
    // Last side
    const ygzf_kp *lastKeys;
    const uint8_t *mpDesc;
    const float *world;
    const uint8_t *mpValid, *outlier, *hasObs;  // nullable: all valid / none outlier / all observed
    const int *lastCnt;
    long long kpStrideLast;
    int cntStrideLast, cntOffLast;
    const float *poses;            // per pair: Rcw[9] tcw[3] Rlw[9] tlw[3]
    // mode 1 = SearchByProjection(F, MapPoints): queries come projected (Frame::isInFrustum); mpValid = mbTrackInView, outlier = isBad()
    // mode 2 = SearchByProjection(Cur, KeyFrame, found, th, ORBdist): queries come projected with their predicted level (the host keeps
    //          the scalar prologue incl. MapPoint::PredictScale's logf); window th*scale[lvl], levels lvl-1..lvl+1, accept <= maxDist
    // mode 3 = SearchForInitialization(F1, F2, vbPrevMatched, vnMatches12, windowSize): Last = F1 (keys, descriptors in mpDesc),
    //          Cur = F2, mpProjX/Y = vbPrevMatched, th = windowSize; TH_LOW + ratio accept, take-over bookkeeping, result in match12
    int mode;
    const float *mpProjX, *mpProjY, *mpProjXR, *mpViewCos, *mpAngle;
    const int *mpLevel;
    float nnratio;
    int maxDist;                   // accept threshold of the in-order pass (TH_HIGH, or ORBdist in mode 2)
    // camera / frame statics
    float fx, fy, cx, cy, mb, mbf, minX, minY, maxX, maxY, gridInvW, gridInvH;
    float scaleFactors[kMaxLevels];
    float th;
    int bMono, checkLevel, checkOri;
    // outputs
    uint8_t *owner;                // per pair kpStrideCur
    int *match;                    // per pair kpStrideCur
    int *nmatches;                 // per pair
    int *match12;                  // mode 3 (SearchForInitialization): vnMatches12, per pair kpStrideLast
    // LDS plan
    int capCur, capLast, descInLds, qpInLds;
    int spill;                     // kSpill* bits: array groups that live in spillScratch instead of LDS
    int specDeep;                  // speculative lists of eight entries instead of four (mode 1)
    void *spillScratch;
    long long spillStride;         // bytes per pair
    void *qpScratch;               // capLast * 32 bytes per pair when !qpInLds
    long long *dbg;                // nullable: 8 wall_clock64 stamps per pair (phase timing, debug)
    // A pair spread over `split` workgroups (few pairs per launch: one Tracking frame): every workgroup builds the grid, takes the queries
    // i % split == its part through projection + speculative lists and leaves them in splitX; the workgroup that arrives last at
    // splitCnt[pair] gathers all lists and runs the in-order pass.  split <= 1: one workgroup per pair, nothing crosses global memory.
    int handoverFence;             // 1: __threadfence() on both sides of the hand-over as well (A/B runs: YGZF_MATCH_FENCE)
    int fixedLanes;                // 1: eight lanes per query in the multi-workgroup scan whatever its level (A/B runs: YGZF_MATCH_LANES=fixed)
    int unitWorld;                 // mode 0: the world point of Last keypoint i is ((x - cx) / fx, (y - cy) / fy, 1), not world[3 i ..]
    int serialOrder;               // 1: the one-wave in-order pass instead of the block-wide fixpoint; 2: fixpoint without list extensions, i.e. the
                                   // hand-over to the one-wave pass at the first exhausted list (tests, A/B runs: YGZF_MATCH_SERIAL)
    int split;
    int *splitCnt;                 // one counter per pair, zero between launches (the last workgroup resets it)
    unsigned char *splitX;         // kMatchSplitRec * capLast bytes per pair
    unsigned *serialFallbacks;     // nullable: counts the pairs whose in-order resolution fell back from the block-wide fixpoint to the one-wave pass
};
constexpr int kSpillSpec = 1, kSpillMisc = 2;
constexpr int kMatchSplitRec = 56;   // uint4 + uint4 (lists of eight) + ushort4 + ushort4 + float angle + hasObs word
size_t match_lds_bytes(int capCur, int capLast, bool descInLds, int spill, size_t *spillBytes, bool specDeep);
hipError_t match_prepare(size_t ldsBytes);
void launch_backproject_unit(hipStream_t st, const ygzf_kp *keys, const int *cnt, long long kpStride, int maxKp, int nFrames, float fx,
                             float fy, float cx, float cy, float *world);
void launch_match_last(hipStream_t st, const MatchArgs &A, int nPairs, size_t ldsBytes);


// SearchByBoW per-node brute force (match_kernels.hip); match (nF ints) must be pre-set to -1, hist (30 ints) and nmatches to 0
void launch_bow(hipStream_t st, int nNodes, const int *kfOff, const int *kfIdx, const int *fOff, const int *fIdx, const uint8_t *kfValid,
                const ygzf_kp *kfKeys, const uint8_t *kfDesc, int nF, const ygzf_kp *fKeys, const uint8_t *fDesc, float nnratio, int checkOri, int *match,
                unsigned char *binOf, int *hist, int *nmatches);

// SearchForTriangulation per-node brute force with the epipolar tests (match_kernels.hip); match12 (n1 ints) pre-set to -1, hist (30 ints) and
// nmatches to 0
struct TriArgs {
    int nEntries, nNodes, n1;       // nEntries = off1[nNodes]
    const int *off1, *idx1, *off2, *idx2;
    const ygzf_kp *keys1, *keys2;
    const uint8_t *desc1, *desc2, *hasMp1, *hasMp2;
    const float *uR1, *uR2;         // nullable: monocular
    const float *sf2, *sigma2;      // mvScaleFactors / mvLevelSigma2 of KF2
    float F[9];                     // F12 row-major
    float ex, ey;                   // epipole of KF1's centre in KF2
    int onlyStereo, checkOri;
    int *match12;
    unsigned char *binOf;
    int *hist, *nmatches;
};
void launch_triangulation(hipStream_t st, const TriArgs &A);

// Frame::isInFrustum over a MapPoint batch (match_kernels.hip); outputs are the mode-1 inputs of k_match_last
struct FrustumArgs {
    int n;
    const uint8_t *candidate;      // nullable: evaluate only where nonzero
    const float *world, *normal, *maxDistInv, *minDistInv, *mfMaxDistance;
    float Rcw[9], tcw[3], Ow[3];
    float fx, fy, cx, cy, mbf, minX, minY, maxX, maxY;
    float viewingCosLimit;
    float levelStep[kMaxLevels];   // levelStep[k] = smallest ratio with PredictScale >= k (k = 1 .. nLevels-1), tabulated on the host
    int nLevels;
    uint8_t *inView;
    float *projX, *projY, *projXR, *viewCos;
    int *level;
};
void launch_frustum(hipStream_t st, const FrustumArgs &A);
// MapPoint::ComputeDistinctiveDescriptors over a MapPoint batch (<= 256 observations per point)
void launch_distinctive(hipStream_t st, int nPoints, const int *obsOff, const uint8_t *desc, int *best, int nLarge, const int *large);   // large: points with > 256 observations

// Frame::ComputeBoW: descriptor -> vocabulary tree descent (match_kernels.hip)
void launch_bow_descend(hipStream_t st, int n, const uint8_t *desc, const int *childOff, const int *childIdx, const uint8_t *nodeDesc, int nidLevel,
                        int *leafNode, int *levelNode);

// ---- FAST-10 (fast10_kernels.hip): Thirdparty/fast replacement -------------------------------------------------------
void launch_fast10(hipStream_t st, const uint8_t *img, int pitch, int x0, int y0, int w, int h, int dx0, int dx1, int dy0, int dy1, int barrier,
                   short *S, int *rowCnt, int *rowKept, int *totals, short *xy, int *scores, int *nonmax, int cap);

// ---- FAST-10 grid detector of the DSO_KEYPOINT path (dso_kernels.hip) + list describe (extract_kernels.hip) -----------------
constexpr int kDsoMaxGrid = 64;   // largest supported mnGridSize (cell side in px)
void launch_dso_occ(hipStream_t st, const unsigned *xy, int n, int w, int h, unsigned *occ);
void launch_dso_cells(hipStream_t st, const uint8_t *img, int pitch, int w, int h, int grid, int nCols, int nRows, unsigned *occ, int *cellCnt,
                      unsigned *cellXY, int *total, int th0, int th1, int take, int occW, bool mark);
void launch_dso_compact(hipStream_t st, const int *cellCnt, const unsigned *cellXY, int nInner, int nExisting, void *list, unsigned *newXY, int level);
// ComputeKeyPointsFast: per-cell Shi-Tomasi vote over the levels' non-max-suppressed libfast corners, then the winners' coordinates
void launch_fgrid_vote(hipStream_t st, const uint8_t *img, int pitch, int w, int h, int level, float scale, const short *xy, const int *nonmax,
                       const int *totals, int maxNonmax, int gridCols, long long nCells, const uint8_t *occ, unsigned long long *cellKey);
void launch_fgrid_gather(hipStream_t st, const unsigned long long *cellKey, long long nCells, const short *xyAll, const int *nmAll, const long long *lvlOff,
                         unsigned *cellXY);
void launch_describe_list(hipStream_t st, const FrameSet &fs, const LevelGeom *dGeom, const void *list, int n, int frame,
                          float *outAngle, uint8_t *outDesc, int cvMode);

// ---- Frame::ComputeStereoMatches (stereo_kernels.hip) ---------------------------------------------------------------------
struct StereoRec {   // per right keypoint: x, row band (min | max << 16), octave | index << 16 (octave 1000: empty band)
    float x;
    unsigned band;
    int octave;
};
constexpr int kStereoBinInts = 4096 + 1;   // bin table of one pair (k_stereo_prep): first sorted record of every bin of rows, then the count
struct StereoArgs {
    // keys / descriptors of pair p: left at keys[p*keyStride + keyOffL + i], right at keys[p*keyStride + keyOffR + i] (desc alike, x32)
    const ygzf_kp *keys;
    const uint8_t *desc;
    long long keyStride;
    int keyOffL, keyOffR;
    const int *cnt;                // counts of pair p at cnt[p*cntStride + cntOffL / cntOffR]
    int cntStride, cntOffL, cntOffR;
    // The right eye may live in ANOTHER context's arrays (ygzf_stereo_pair_host: the two eyes of one pair extracted on two streams): keysR / descR /
    // cntR non-null = the right keys of pair p at keysR[p*keyStride + keyOffR + i] etc., its pyramid = frame frame0 + p*frameStep + 1 of fsR
    const ygzf_kp *keysR;
    const uint8_t *descR;
    const int *cntR;
    FrameSet fsR;
    // pyramids: left image of pair p = frame frame0 + p*frameStep of fs, right = the next frame
    FrameSet fs;
    const LevelGeom *geom;
    int frame0, frameStep;
    float scale[kMaxLevels], invScale[kMaxLevels];
    float mb, mbf;
    int nRows;                     // rows of level 0
    StereoRec *rec;                // scratch, pair p at rec + p*recStride: the right keypoints' records sorted by the bin of their band's first row
    long long recStride;
    int *binStart;                 // scratch, pair p at binStart + p*kStereoBinInts
    int nBins, binShift;           // bins of 2^binShift rows, nBins = ((nRows - 1) >> binShift) + 1 <= 4096
    int bandMax;                   // >= last row - first row of every band: 2 * (2 * largest scale factor) + 2
    float *uRight, *depth;         // outputs, pair p at + p*outStride, one per left keypoint
    int *sad;                      // accepted SAD per left keypoint (-1: no match), same stride
    long long outStride;
};
void launch_stereo(hipStream_t st, const StereoArgs &A, int nPairs, int maxLeft, int maxRight);

// ---- ORBmatcher::FindDirectProjection + Align2D over a candidate batch (direct_kernels.hip) ---------------------------------
struct DirectArgs {
    FrameSet cache;                // the image cache: slot s = frame s of this FrameSet
    const LevelGeom *geom;
    int nlevels;
    int curSlot;
    float curTcw[7];
    float fx, fy, cx, cy;
    float scale[kMaxLevels], invScale[kMaxLevels];
    float invLevelSigma2_1;        // mvInvLevelSigma2[1]
    int n;
    const int *refSlot;
    const float *refTcw7;          // n x 7 (quaternion x,y,z,w + translation)
    const ygzf_kp *refKp;
    const float *mpWorld;          // n x 3
    float *pxCurr;                 // n x 2, in/out
    int *searchLevel;
    uint8_t *success;
    uint8_t *patches;              // nullable: n x 100 (_patch_with_border, tests)
};
void launch_direct_projection(hipStream_t st, const DirectArgs &A);

// ---- sparse image alignment (align_kernels.hip) ----------------------------------------------------------------------
struct SiaLevel {
    const uint8_t *img;
    int w, h, pitch;
};
struct FiaArgs {                    // Frame::GetFeaturesInArea queries against one frame's keypoints
    const ygzf_kp *keys;
    int n;
    float minX, minY, gridInvW, gridInvH;
    int nq;
    const float *xyr;               // 3 per query
    const int *levels;              // 2 per query (minLevel, maxLevel) or null (-1, -1)
    int cap;
    int *outIdx, *outN;             // nq x cap indices (reference order), nq counts (uncapped)
};
size_t fia_lds_bytes(int n);
hipError_t launch_features_in_area(hipStream_t st, const FiaArgs &A);

struct SiaArgs {
    int ldsFeat;                    // feature slots of the dynamic LDS carve-up (float4 s_feat[ldsFeat] | float2 s_uv[ldsFeat] | float4 s_jac[2*ldsFeat])
    int stageOff, stageBytes;       // byte offset (from the start of dynamic LDS) and size of the staged current-image region
    int jacLds;                     // the point-only Jacobian terms are staged in LDS (sia_jac_in_lds(ldsFeat))
    const ygzf_kp *keys;            // ref keypoints, pair p at keys + p*kpStride
    const float *world;             // MapPoint world positions, 3 per keypoint
    const uint8_t *mpValid, *outlier;  // nullable
    long long kpStride;
    const int *nRef;                // per-pair count, or null -> n
    int n;
    const float *poses;             // per pair 14 floats: ref Tcw (qx qy qz qw tx ty tz), cur Tcw
    const SiaLevel *refLv, *curLv;  // per pair lvStride entries, indexed by pyramid level
    int lvStride;
    float invScale[kMaxLevels];
    float fx, fy, cx, cy;
    int maxLevel, minLevel, nIter;
    float eps;
    float *patchCache;              // kpStride*48 floats per pair (16-byte aligned): 12 planes of kpStride float4, plane 3*row + {patch, dx, dy}
    float *jacCache;                // unused (Jacobians are rebuilt from dx, dy each iteration)
    float *momCache;                // 4 floats per keypoint (16-byte aligned), pair p at + p*kpStride*4: gradient moments of the reference patch
    uint8_t *visible;               // kpStride per pair
    // perLevel: the reference patches of all levels come from k_sia_precompute (launch_sia_precompute) instead of k_sia_run's own per-level
    // phase: patchCache / momCache hold one block per level (level l at + (l - minLevel) * pcLevelStride / momLevelStride floats, pairs inside
    // a block as before), levelFlags one byte per (level, pair, keypoint): visible at this level or a coarser one
    int unitWorld;                  // the world point of keypoint i is ((x - cx) / fx, (y - cy) / fy, 1), not world[3 i ..]
    int perLevel;
    size_t pcLevelStride, momLevelStride, flagLevelStride;
    uint8_t *levelFlags;
    float *out;                     // 48 floats per pair: TCR[7], ret, iters, chi2, pad[2], H[36]
    long long *dbg;                 // nullable: phase clocks of pair 0 (debug)
};
size_t sia_lds_bytes(int maxFeatures);
bool sia_jac_in_lds(int maxFeatures);
size_t sia_stage_bytes(size_t featBytes, size_t largestLevelBytes);
hipError_t sia_prepare(size_t ldsBytes);
void launch_sia_precompute(hipStream_t st, const SiaArgs &A, int nPairs, int maxN);
void launch_sia(hipStream_t st, const SiaArgs &A, int nPairs, size_t ldsBytes);

}  // namespace ygzf
#endif
