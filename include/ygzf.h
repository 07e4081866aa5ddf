/* ygzf.h -- C ABI of libygzf: the MI355X (gfx950) implementation of ORB-YGZ-SLAM's per-frame hot path.
 *
 * This is the drop-in boundary: plain C, POD structs, raw pointers and sizes; no C++ or torch types cross it.
 * The C++ class shells in orb_ygz_slam_amd/csrc/host/ (ygz::ORBextractor, ygz::ORBmatcher, ygz::SparseImgAlign,
 * same public signatures as the reference) are thin wrappers over these entry points; INTEGRATION.md shows how the
 * reference's Tracking.cc links against them.  Every function cites the reference interface it replaces
 * (paths relative to the reference checkout gaoxiang12/ORB-YGZ-SLAM).
 *
 * Conventions: every function returns YGZF_OK (0) or a negative ygzf_status; no exception crosses the ABI; the
 * caller owns all host buffers; the context owns device buffers and its HIP stream.  A context is not thread-safe:
 * use one per thread (the reference runs one ORBextractor per eye on two std::threads, src/Frame.cc:728-731).
 * There is no CPU fallback: without a usable HIP device ygzf_create fails with YGZF_ERR_NO_DEVICE.
 */
#ifndef YGZF_H
#define YGZF_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum ygzf_status {
    YGZF_OK = 0,
    YGZF_ERR_INVALID = -1,     /* bad argument (null pointer, size out of range, capacity too small) */
    YGZF_ERR_NO_DEVICE = -2,   /* no HIP device / device index out of range */
    YGZF_ERR_HIP = -3,         /* a HIP runtime call failed; see ygzf_last_error */
    YGZF_ERR_UNSUPPORTED = -4, /* configuration outside what the kernels cover (e.g. level wider than 4096 px) */
    YGZF_ERR_STATE = -5        /* call order violated (e.g. fetch before extract) */
} ygzf_status;

typedef struct ygzf_ctx ygzf_ctx;

/* ORBextractor::ORBextractor(nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST)  include/ORBextractor.h:58-59,
 * src/ORBextractor.cc:412-470 */
typedef struct ygzf_extractor_cfg {
    int nfeatures;
    float scale_factor;
    int nlevels;
    int ini_th_fast;
    int min_th_fast;
    int cv_mode;               /* ygzf_cv_mode; 0 (what a zero-initialised tail gives) = the reference's own build */
} ygzf_extractor_cfg;

/* Which OpenCV generation's 8-bit cv::GaussianBlur(7x7, sigma 2) the descriptors are sampled from (src/ORBextractor.cc:1010, :1083).
 * The reference does not pin OpenCV (CMakeLists.txt:40-46) and this is the one primitive of the path whose arithmetic differs between
 * the versions it can be built against; keypoints, angles and pyramids are the same in all modes.
 *   YGZF_CV_LEGACY_SSE2  OpenCV 2.4.x / 3.0-3.3 on x86 (the reference's tested versions, built -march=native): fixed-point kernel
 *                        {18,34,49,55,49,34,18}/256 per axis, column pass in SSE2 floats -> exact quotient rounded half to EVEN on the
 *                        columns of the vector body [0, width & ~3), (sum + 2^15) >> 16 on the last width % 4 columns
 *   YGZF_CV_LEGACY_INT   the same kernel without the SSE2 column body: (sum + 2^15) >> 16 everywhere
 *   YGZF_CV_4            OpenCV >= 3.4.11 / 4.x bit-exact path: Q8.8 kernel {18,34,48,56,48,34,18}, (sum + 2^15) >> 16 */
typedef enum ygzf_cv_mode { YGZF_CV_LEGACY_SSE2 = 0, YGZF_CV_LEGACY_INT = 1, YGZF_CV_4 = 2 } ygzf_cv_mode;

/* cv::KeyPoint, bit-compatible (7 x 4 bytes): pt.x pt.y size angle response octave class_id */
typedef struct ygzf_kp {
    float x, y, size, angle, response;
    int32_t octave, class_id;
} ygzf_kp;

/* ---- context ---------------------------------------------------------------------------------------------- */
/* Creates an extractor context on HIP device `device` able to process up to max_batch frames of up to
 * max_width x max_height pixels per call. */
int ygzf_create(int device, const ygzf_extractor_cfg *cfg, int max_width, int max_height, int max_batch, ygzf_ctx **out);
void ygzf_destroy(ygzf_ctx *ctx);
/* Human-readable description of the last failure on `ctx` (or of the last failed ygzf_create when ctx == NULL). */
const char *ygzf_last_error(const ygzf_ctx *ctx);

/* The constructor's tables without a context or a device (host arithmetic only): what ORBextractor::ORBextractor fills before any image
 * arrives (src/ORBextractor.cc:419-445) -- Frame's constructors read them first thing (src/Frame.cc:119-125).  Arrays of cfg->nlevels
 * entries, any may be NULL. */
int ygzf_scale_tables_host(const ygzf_extractor_cfg *cfg, float *scale, float *inv_scale, float *sigma2, float *inv_sigma2, int *nfeat);
/* How ComputePyramid (src/ORBextractor.cc:1130-1150) of ONE w x h frame is launched, without a context or a device (host arithmetic only; for
 * inspection and for tests of the plan's invariants).  *n_strips = 0: one launch per level.  Otherwise the whole chain is one launch of n_strips
 * workgroups (k_pyr_strips): workgroup s produces rows [ca, cb) of level l in LDS -- for level 0: stages them -- and
 * writes the rows [wa, wb) of them it owns to the pyramid; rows[(s * nlevels + l) * 4 + {0,1,2,3}] = ca, cb, wa, wb (rows_cap entries available,
 * at least n_strips * nlevels * 4 are needed; pass rows = NULL to ask for n_strips first: at most 64).  level_wh[2 l], level_wh[2 l + 1] = size
 * of level l.  lds_bytes = LDS per workgroup.  lds_bytes, level_wh and rows may be NULL. */
int ygzf_pyramid_plan_host(const ygzf_extractor_cfg *cfg, int w, int h, int *n_strips, int *lds_bytes, int *level_wh, unsigned short *rows, int rows_cap);
/* The level the strips of that plan start from: 0 = the image (the whole chain is the one launch); b > 0 (images too large for that: 3840 x 2160): levels
 * 1 .. b come from one launch each and the strips stage level b instead of the image -- rows[..] of the levels below b are zero.  Negative: an error code. */
int ygzf_pyramid_plan_base_host(const ygzf_extractor_cfg *cfg, int w, int h);

/* ORBextractor::operator() on the image whose pyramid the context's previous call, ygzf_compute_pyramid, left on the device: FAST, octree,
 * orientation and descriptors without a second upload and a second pyramid (Frame's constructors call ComputePyramid and then the
 * extractor on the same image: src/Frame.cc:807-813, :332-348).  YGZF_ERR_STATE when any other image operation came in between. */
int ygzf_extract_resident(ygzf_ctx *ctx, ygzf_kp *kps, uint8_t *desc, int cap, int *n_out);

/* Free / total memory of the context's device after draining the context's stream (diagnostics: leak checks, sizing of resident batches). */
int ygzf_device_mem_info(ygzf_ctx *ctx, size_t *free_bytes, size_t *total_bytes);

/* Tuning knob of the cell loop of ComputeKeyPointsOctTree (src/ORBextractor.cc:747-781: FAST(iniThFAST), and FAST(minThFAST) where that
 * finds nothing).  The keypoints are the same under every plan; only the cost differs with the image content:
 *   YGZF_FAST_PLAN_AUTO       (default) chosen before every launch from statistics of the context's earlier launches
 *   YGZF_FAST_PLAN_ONE_PASS   one corner test at minThFAST per cell, every corner scored, both thresholds' 3x3 NMS from the same score map
 *   YGZF_FAST_PLAN_INI_FIRST  the reference's order: test and score at iniThFAST, a second test at minThFAST only in cells left empty
 * ygzf_get_fast_plan reports the plan the next launch will use (1 or 2). */
enum ygzf_fast_plan { YGZF_FAST_PLAN_AUTO = 0, YGZF_FAST_PLAN_ONE_PASS = 1, YGZF_FAST_PLAN_INI_FIRST = 2 };
int ygzf_set_fast_plan(ygzf_ctx *ctx, int plan);
int ygzf_get_fast_plan(const ygzf_ctx *ctx, int *plan);
/* Which form of the same cell loop runs (identical results; tests run both against the oracle):
 *   YGZF_FAST_KERNEL_AUTO              (default) the cell-table form wherever it applies (cells up to 41 px wide, i.e. every usual configuration)
 *   YGZF_FAST_KERNEL_REGISTER_STAGING  k_fast_quads: every wave derives its cell from the level geometry and stages the window through registers
 *   YGZF_FAST_KERNEL_CELL_TABLE        k_fast_tab: per-cell records precomputed with the geometry, window staged by LDS-DMA (falls back to
 *                                      REGISTER_STAGING for wider cells) */
enum ygzf_fast_kernel { YGZF_FAST_KERNEL_AUTO = 0, YGZF_FAST_KERNEL_REGISTER_STAGING = 1, YGZF_FAST_KERNEL_CELL_TABLE = 2 };
int ygzf_set_fast_kernel(ygzf_ctx *ctx, int kernel);
/* Statistics the cell loop left behind on its last sampled launch (every 16th 2x2 cell group reports): corner-bearing quads (4 pixels) per
 * pass-1 run of a cell, and pass-1 runs per cell -- 1.0 under ONE_PASS; under INI_FIRST 1 + the fraction of cells that were empty at iniThFAST and
 * ran the corner test a second time at minThFAST (src/ORBextractor.cc:765-768).  Any pointer may be NULL. */
int ygzf_get_fast_stats(const ygzf_ctx *ctx, float *corner_quads_per_pass, float *passes_per_cell);
/* Phase clocks of k_fast_tab (kernel 0) and k_describe (kernel 1): shader-clock cycles summed over the sampled waves (one in 64, the last 8192 samples) since the last
 * reset -- out16[0..7] per phase, out16[14] pass-1 runs (kernel 0), out16[15] waves.  Only a library built with -DYGZF_PHASE_CLOCK (python -m
 * orb_ygz_slam_amd.build --phase-clock -> lib_ab/libygzf_clk.so, tools/fast_phases.py) carries the stamps; the product build returns YGZF_ERR_INVALID. */
int ygzf_phase_clocks(ygzf_ctx *ctx, int kernel, unsigned long long *out16, int reset);
/* Extract-ahead (default off).  When on, ygzf_compute_pyramid queues the ORBSLAM_KEYPOINT extraction of the same image (FAST, octree,
 * orientation, descriptors) right behind its pyramid kernels and reads the levels back on a second stream, so that the extraction runs while the
 * levels cross the link and while the caller works on them -- ygz::Frame's constructor clones them (src/Frame.cc:807-813) before
 * Frame::ExtractORB (:332-348) asks for the keypoints.  A following ygzf_extract_resident then launches nothing and only collects the
 * results (same keypoints, same bytes).  The price: the extraction is computed for every pyramid, also when no ygzf_extract_resident
 * follows, and it counts as an extraction for the batch calls' "previous frame" (ygzf_match_batch_prev / ygzf_align_batch_prev). */
int ygzf_set_extract_ahead(ygzf_ctx *ctx, int on);
/* The carried "previous frame" (default on).  Every batch extraction first copies the last frame of the batch before it into slot 0 of the result arrays: the
 * Last frame of pair 0 in ygzf_match_batch_prev / ygzf_align_batch_prev (the reference: Tracking keeps mLastFrame, src/Tracking.cc:1262).  A caller that only
 * extracts (or only pairs stereo eyes) can switch it off: one launch less per extraction, and a host-frame call's upload no longer waits behind it.  While it is
 * off ygzf_match_batch_prev and ygzf_align_batch_prev return YGZF_ERR_STATE; switched on again, the next extraction carries the then-last frame as usual. */
int ygzf_set_carry_previous(ygzf_ctx *ctx, int on);

/* ORBextractor::GetLevels / GetScaleFactors / GetInverseScaleFactors / GetScaleSigmaSquares /
 * GetInverseScaleSigmaSquares (include/ORBextractor.h:84-106); arrays of nlevels floats, any may be NULL. */
int ygzf_get_levels(const ygzf_ctx *ctx);
int ygzf_get_scale_tables(const ygzf_ctx *ctx, float *scale, float *inv_scale, float *sigma2, float *inv_sigma2);
/* mnFeaturesPerLevel (src/ORBextractor.cc:434-445) */
int ygzf_get_features_per_level(const ygzf_ctx *ctx, int *nfeat);
/* Size of pyramid level `level` for a w x h input (src/ORBextractor.cc:1131-1132). */
int ygzf_level_size(const ygzf_ctx *ctx, int w, int h, int level, int *lw, int *lh);
/* Row pitch of level 0 on the device for images w pixels wide (w rounded up to 64).  Host frames laid out at this pitch (cv::Mat rows of a
 * wider allocation, or ygzf_alloc_host memory) are uploaded by ygzf_extract_batch_host / ygzf_mgpu_* as whole frames instead of row by row --
 * where the image enters the reference: src/Frame.cc:807-813 (ComputePyramid + ExtractORB on the caller's cv::Mat).  Any other pitch works too. */
int ygzf_host_row_pitch(int w);
/* Upper bound on keypoints returned per frame for a w x h input (sum over levels of N_l + 3, or 4*nIni). */
int ygzf_max_keypoints(const ygzf_ctx *ctx, int w, int h);

/* ---- ORBextractor::ComputePyramid(cv::Mat)  src/ORBextractor.cc:1129-1150 ----------------------------------------
 * Host image in; all nlevels levels are computed on the device and copied back tight (step == width) into
 * levels_out[l] (each at least lw*lh bytes), which is what Frame clones into mvImagePyramid (src/Frame.cc:810-813).
 * The 19-px border the reference adds is never read by the hot path and is not materialised. */
int ygzf_compute_pyramid(ygzf_ctx *ctx, const uint8_t *img, int w, int h, int stride, uint8_t *const *levels_out);

/* ---- ORBextractor::operator()(InputArray image, InputArray mask, vector<KeyPoint>&, OutputArray descriptors)
 *      src/ORBextractor.cc:970-1028 (mask ignored, as in the reference) ------------------------------------------
 * One host frame in, keypoints (level-major order) and N x 32 descriptors out.  cap = capacity of kps/desc in
 * keypoints (use ygzf_max_keypoints).  Empty image (img == NULL or w*h == 0) => *n_out = 0, YGZF_OK (the reference
 * returns silently, :972-973). */
int ygzf_extract(ygzf_ctx *ctx, const uint8_t *img, int w, int h, int stride, ygzf_kp *kps, uint8_t *desc, int cap, int *n_out);

/* Batched form of the same operator on DEVICE-resident frames: frame f starts at d_imgs + f*frame_stride, rows are
 * row_pitch bytes apart (base pointer, row_pitch and frame_stride must be multiples of 4: the kernels use aligned 32-bit loads).
 * Results stay on the device (chain into ygzf_match_*) until fetched.  Asynchronous on the
 * context stream; ygzf_sync / ygzf_batch_counts / ygzf_batch_fetch synchronise.
 * Lifetime: the context keeps d_imgs as level 0 of the batch -- ygzf_describe_keys, ygzf_stereo_batch, ygzf_align_batch_prev and
 * ygzf_batch_fetch_level(…, 0, …) read it later -- so the frames must stay valid and unchanged until the next ygzf_extract* call on this
 * context (or its destruction). */
int ygzf_extract_batch_device(ygzf_ctx *ctx, const uint8_t *d_imgs, int n_frames, int w, int h, int row_pitch, size_t frame_stride);
/* Same, host frames (H2D copy of each frame included).  The copy is asynchronous too: `imgs` must stay valid and unchanged until the next
 * synchronising call on this context (ygzf_sync, ygzf_batch_counts, ygzf_batch_fetch*, ...); page-locked memory gives the full PCIe rate. */
int ygzf_extract_batch_host(ygzf_ctx *ctx, const uint8_t *imgs, int n_frames, int w, int h, int row_pitch, size_t frame_stride);
/* Same from a list of frame pointers (the data pointers of an array of cv::Mat; a round-robin share of a larger clip): frames[f] = level 0 of
 * frame f, rows of row_pitch bytes.  One copy per frame on the context's stream, asynchronous when the memory is page-locked. */
int ygzf_extract_batch_host_frames(ygzf_ctx *ctx, const uint8_t *const *frames, int n_frames, int w, int h, int row_pitch);
int ygzf_batch_counts(ygzf_ctx *ctx, int *n_kp /* n_frames ints */);
int ygzf_batch_fetch(ygzf_ctx *ctx, int frame, ygzf_kp *kps, uint8_t *desc, int cap, int *n_out);
/* All frames of the batch with one synchronisation: kps / desc hold n_frames rows of `stride` (>= ygzf_max_keypoints) entries, frame f's
 * first n_kp[f] entries are valid.  Pass page-locked host memory (hipHostMalloc / hipHostRegister) for full PCIe rate. */
int ygzf_batch_fetch_all(ygzf_ctx *ctx, ygzf_kp *kps, uint8_t *desc, int *n_kp, int stride);
/* The same results as ONE block and ONE copy, for launches of a few frames (a Tracking frame, a stereo pair): host (page-locked for the link's rate)
 * receives [n_kp: n_frames ints | keypoint rows | descriptor rows], rows of *row_entries entries (ygzf_max_keypoints of the batch's image size), the
 * parts at offsets 0, *off_kps and *off_desc (256-byte aligned); *bytes_out = bytes written (may be NULL).  Three separate device-to-host copies
 * cost a one-frame call 20 us more. */
int ygzf_batch_fetch_packed(ygzf_ctx *ctx, void *host, size_t host_bytes, size_t *off_kps, size_t *off_desc, int *row_entries, size_t *bytes_out);
/* Copies pyramid level `level` of batch frame `frame` back to the host, tight. */
int ygzf_batch_fetch_level(ygzf_ctx *ctx, int frame, int level, uint8_t *out);
int ygzf_sync(ygzf_ctx *ctx);

/* ---- stage-level accessors used by the parity tests (tests/test_gpu_extract.py) -------------------------------------
 * FAST candidates of (frame, level) after the per-cell NMS / threshold fallback, in the reference's vToDistributeKeys
 * order (src/ORBextractor.cc:747-781): region coordinates (origin (16,16)), cell-major, raster inside a cell. */
int ygzf_batch_fetch_candidates(ygzf_ctx *ctx, int frame, int level, int *xs, int *ys, int *scores, int cap, int *n_out);
/* Keypoints of (frame, level) selected by the octree, level coordinates, octree list order. */
int ygzf_batch_fetch_level_keypoints(ygzf_ctx *ctx, int frame, int level, int *xs, int *ys, int *scores, int cap, int *n_out);

/* ---- ORBmatcher::DescriptorDistance(const cv::Mat&, const cv::Mat&)  src/ORBmatcher.cc:1507-1523 ------------------
 * Batched on the device: dist[i] = popcount(a[i] xor b[i]) over 256 bits, n pairs of 32-byte host descriptors. */
int ygzf_descriptor_distance(ygzf_ctx *ctx, const uint8_t *a, const uint8_t *b, int n, int *dist);

/* ---- ORBmatcher::SearchByProjection(Frame &CurrentFrame, const Frame &LastFrame, const float th, const bool bMono,
 *      bool checkLevel)  src/ORBmatcher.cc:1218-1350, together with the Frame grid it searches
 *      (Frame::AssignFeaturesToGrid / PosInGrid / GetFeaturesInArea, src/Frame.cc:314-330, 483-493, 424-481) ------------
 * Frame / MapPoint objects are passed as the plain arrays the function reads. */
typedef struct ygzf_camera {         /* Frame statics: fx fy cx cy mb mbf (src/Frame.cc:27-33) and the image bounds mnMinX.. */
    float fx, fy, cx, cy, mb, mbf;
    float min_x, min_y, max_x, max_y;
} ygzf_camera;

typedef struct ygzf_frame_view {     /* the slice of `Frame` the matcher reads */
    int n;                           /* Frame::N */
    const ygzf_kp *keys;             /* mvKeys */
    const uint8_t *desc;             /* mDescriptors, n x 32 */
    const float *u_right;            /* mvuRight, or NULL for monocular (all -1) */
    const float *scale_factors;      /* mvScaleFactors, or NULL to use the context's extractor tables */
    int nlevels;
} ygzf_frame_view;

/* One (Last, Cur) pair from host arrays.
 *   per Last keypoint i: mp_valid[i] (mvpMapPoints[i] != NULL; NULL array = all valid), outlier[i] (mvbOutlier; NULL = none),
 *   mp_has_obs[i] (pMP->Observations() > 0; NULL = all), mp_world (GetWorldPos, n x 3), mp_desc (GetDescriptor, n x 32);
 *   Rcw/tcw = CurrentFrame.mTcw, Rlw/tlw = LastFrame.mTcw (row-major 3x3 + 3).
 *   cur_owner (in/out, cur->n bytes): state of CurrentFrame.mvpMapPoints: 0 NULL, 1 set by a point with no observations,
 *   2 set by a point with Observations() > 0.  cur_match (out): index of the Last keypoint whose MapPoint was written to
 *   CurrentFrame.mvpMapPoints[i2]; -1 untouched; -2 written, then reset to NULL by the rotation-consistency check.
 *   *nmatches = the function's return value. */
int ygzf_search_by_projection_last(ygzf_ctx *ctx, const ygzf_frame_view *cur, const ygzf_camera *cam, int last_n, const ygzf_kp *last_keys,
                                   const uint8_t *mp_valid, const uint8_t *outlier, const uint8_t *mp_has_obs, const float *mp_world,
                                   const uint8_t *mp_desc, const float *Rcw, const float *tcw, const float *Rlw, const float *tlw, float th,
                                   int b_mono, int check_level, int check_orientation, uint8_t *cur_owner, int *cur_match, int *nmatches);

/* ---- Frame::isInFrustum(MapPoint *pMP, float viewingCosLimit)   src/Frame.cc:363-422 over a MapPoint batch, and its
 *      fusion with SearchByProjection(F, MapPoints) = the body of Tracking::SearchLocalPoints (src/Tracking.cc:1544-1593) ---------------
 * Per MapPoint i: world = GetWorldPos(), normal = GetNormal(), max_dist_inv / min_dist_inv = GetMax/MinDistanceInvariance(),
 * mf_max_distance = mfMaxDistance (numerator of MapPoint::PredictScale, src/MapPoint.cc:359-373), candidate[i] (NULL = all) = the point
 * is evaluated at all (not bad, not already matched in this frame).  Rcw / tcw / Ow = Frame::mRcw / mtcw / mOw.
 * Outputs = the fields isInFrustum writes: in_view = mbTrackInView, proj_x / proj_y / proj_xr = mTrackProjX/Y/XR, level =
 * mnTrackScaleLevel, view_cos = mTrackViewCos (defined where in_view).  PredictScale's ceil(logf(ratio) / log_scale_factor) is a step
 * function of ratio: the library tabulates its steps with the host's libm (ygzf_predict_scale_steps) and the device compares.
 * ygzf_search_local_points runs isInFrustum and then SearchByProjection(F, MapPoints, th, check_level) (nnratio as the matcher's) on
 * the device without a host round trip; the frustum outputs are optional (NULL = not copied back). */
typedef struct ygzf_frustum_in {
    const float *world, *normal;                 /* n x 3 each */
    const float *max_dist_inv, *min_dist_inv, *mf_max_distance;
    const uint8_t *candidate;                    /* nullable */
    float Rcw[9], tcw[3], Ow[3];
    float log_scale_factor;                      /* Frame::mfLogScaleFactor */
    float viewing_cos_limit;
} ygzf_frustum_in;
int ygzf_predict_scale_steps(float log_scale_factor, int nlevels, float *steps);   /* host only: steps[k] = smallest ratio with level >= k */
int ygzf_is_in_frustum_batch(ygzf_ctx *ctx, const ygzf_camera *cam, int nlevels, int n, const ygzf_frustum_in *in, uint8_t *in_view, float *proj_x,
                             float *proj_y, float *proj_xr, int *level, float *view_cos);
int ygzf_search_local_points(ygzf_ctx *ctx, const ygzf_frame_view *F, const ygzf_camera *cam, int n_mp, const ygzf_frustum_in *in,
                             const uint8_t *mp_has_obs, const uint8_t *mp_desc, float th, int check_level, float nnratio, uint8_t *owner, int *match,
                             int *nmatches, uint8_t *in_view, float *proj_x, float *proj_y, float *proj_xr, int *level, float *view_cos);

/* ---- Frame::GetFeaturesInArea(x, y, r, minLevel, maxLevel)   src/Frame.cc:424-481 over the grid of Frame::AssignFeaturesToGrid
 *      (:314-330, PosInGrid :483-493) as a direct query --------------------------------------------------------------------
 * keys = the frame's mvKeys, cam->min_x .. max_y = mnMinX .. mnMaxY.  Query q = xyr[3q .. 3q+2] with levels[2q], levels[2q+1] (levels NULL:
 * -1, -1 = no level test).  out_idx[q * cap ..] receives the indices the reference's vector holds, in its order (grid column, grid row,
 * keypoint index); out_n[q] = that vector's size -- when it exceeds cap only the first cap indices are stored. */
int ygzf_features_in_area(ygzf_ctx *ctx, const ygzf_camera *cam, int n_keys, const ygzf_kp *keys, int n_queries, const float *xyr,
                          const int *levels, int cap, int *out_idx, int *out_n);

/* ---- MapPoint::ComputeDistinctiveDescriptors()   src/MapPoint.cc:211-271 over a MapPoint batch -----------------------------
 * Point p owns the observation descriptors desc[obs_off[p] .. obs_off[p+1]) (32 bytes each, those of its non-bad KeyFrames in map order);
 * best_idx[p] = index within the point's observations of the descriptor with the least median Hamming distance to the others (first
 * minimum, median = sorted row element (size_t)(0.5*(N-1))), -1 for a point without observations.  Any number of observations per point (up to 256 in registers, beyond that through a distance histogram). */
int ygzf_distinctive_descriptors_batch(ygzf_ctx *ctx, int n_points, const int *obs_off, const uint8_t *desc, int *best_idx);

/* ---- ORBmatcher::SearchByProjection(Frame &CurrentFrame, KeyFrame *pKF, const set<MapPoint*> &sAlreadyFound, th, ORBdist)
 *      src/ORBmatcher.cc:1352-1469 (Tracking::Relocalization: th 10 / ORBdist 100, then th 3 / ORBdist 64) -----------------------
 * The O(N) scalar prologue stays on the host (:1371-1400: projection with CurrentFrame.mTcw, image-bounds and distance gates,
 * MapPoint::PredictScale with its logf) -- the class shell does it with the reference's own expressions.  Per KeyFrame MapPoint i:
 * valid[i] = passed every gate, proj_x/proj_y = (u, v), pred_level = nPredictedLevel, kf_angle = pKF->mvKeys[i].angle,
 * mp_desc = GetDescriptor().  The device does the window search (radius th*scale[level], levels level-1..level+1), the in-order
 * "slot already taken" resolution (:1419), the accept rule bestDist <= ORBdist and the rotation histogram (:1433-1464).
 * owner (in/out, cur->n bytes): nonzero <=> CurrentFrame.mvpMapPoints[i] != NULL (any MapPoint blocks); match / nmatches as above. */
int ygzf_search_by_projection_kf(ygzf_ctx *ctx, const ygzf_frame_view *cur, const ygzf_camera *cam, int n_mp, const uint8_t *valid,
                                 const float *proj_x, const float *proj_y, const int *pred_level, const float *kf_angle, const uint8_t *mp_desc,
                                 float th, int orb_dist, int check_orientation, uint8_t *owner, int *match, int *nmatches);

/* ---- ORBmatcher::SearchForInitialization(Frame &F1, Frame &F2, vector<cv::Point2f> &vbPrevMatched, vector<int> &vnMatches12, int windowSize)
 *      src/ORBmatcher.cc:375-478 (Tracking::MonocularInitialization, nnratio 0.9, windowSize 100) ------------------------------------
 * Level-0 keys of F1 against the windowSize window around prev_matched_xy[i] in F2 (level 0), recorded-distance candidate gate
 * (:414-415), accept bestDist <= TH_LOW && bestDist < nnratio * bestDist2, take-over of an already matched F2 key (:427-430), rotation
 * histogram.  matches12 (F1->n ints) = vnMatches12; prev_matched_xy (F1->n x 2, in/out) is updated like vbPrevMatched (:470-474). */
int ygzf_search_for_initialization(ygzf_ctx *ctx, const ygzf_frame_view *F1, const ygzf_frame_view *F2, const ygzf_camera *cam,
                                   float *prev_matched_xy, int window_size, float nnratio, int check_orientation, int *matches12, int *nmatches);

/* ---- ORBmatcher::SearchByBoW(KeyFrame *pKF, Frame &F, vector<MapPoint*> &vpMapPointMatches)   src/ORBmatcher.cc:155-263 ---------------
 * The DBoW2 FeatureVector merge-join (:169-247, equal node ids) stays with the caller -- the vocabulary and Frame::ComputeBoW are outside
 * this path -- and arrives as a joined node list: node k pairs the KeyFrame feature indices kf_idx[kf_off[k] .. kf_off[k+1]) with the
 * Frame feature indices f_idx[f_off[k] .. f_off[k+1]) (each offset array has n_nodes + 1 entries).  kf_valid[i] = the KeyFrame's MapPoint
 * i exists and is not bad.  The device runs the per-node brute force (best / second best, "slot already matched" chain, TH_LOW + ratio)
 * and the rotation histogram.  match (n_f ints): KeyFrame feature index whose MapPoint lands in vpMapPointMatches[i]; -1 none;
 * -2 culled by the rotation check.  At most 4096 Frame features per node. */
int ygzf_search_by_bow(ygzf_ctx *ctx, int n_nodes, const int *kf_off, const int *kf_idx, const int *f_off, const int *f_idx, int n_kf,
                       const uint8_t *kf_valid, const ygzf_kp *kf_keys, const uint8_t *kf_desc, int n_f, const ygzf_kp *f_keys, const uint8_t *f_desc,
                       float nnratio, int check_orientation, int *match, int *nmatches);

/* ---- ORBmatcher::SearchForTriangulation(KeyFrame *pKF1, KeyFrame *pKF2, Matrix3f &F12, vector<pair<size_t,size_t>> &vMatchedPairs,
 *      const bool bOnlyStereo)   src/ORBmatcher.cc:596-741, with CheckDistEpipolarLine :136-153 (LocalMapping::CreateNewMapPoints) -----------
 * Joined node list as for ygzf_search_by_bow: node k pairs the features idx1[off1[k] .. off1[k+1]) of KF1 with idx2[off2[k] .. off2[k+1])
 * of KF2.  kf1 / kf2: mvKeys (the reference reads the DISTORTED keys here), mDescriptors, mvuRight (NULL: monocular); kf2's scale_factors
 * = pKF2->mvScaleFactors (NULL: the context's tables), level_sigma2_2 = pKF2->mvLevelSigma2 (NULL: scale factor squared).
 * has_mp[i] != 0 <=> the KeyFrame holds a MapPoint in slot i (GetMapPoint(i) != NULL): such features are skipped on either side.
 * F12 row-major 3x3; Cw1 = pKF1->GetCameraCenter(), R2w (row-major) / t2w = pKF2->GetRotation() / GetTranslation(), cam2 = KF2's
 * fx fy cx cy: the epipole of :601-608 is computed from them.  The device runs the per-node brute force (TH_LOW, least distance, last of
 * equals), the epipole exclusion disc (:668-673), the epipolar-line gate (3.84 sigma2) and the rotation histogram.
 * match12 (kf1->n ints): matched KF2 feature; -1 none; -2 culled by the rotation check.  vMatchedPairs = {(i, match12[i]) : match12[i] >= 0}
 * in ascending i (:731-736); *nmatches = their number.  At most 65535 KF2 features per node. */
int ygzf_search_for_triangulation(ygzf_ctx *ctx, int n_nodes, const int *off1, const int *idx1, const int *off2, const int *idx2,
                                  const ygzf_frame_view *kf1, const uint8_t *has_mp1, const ygzf_frame_view *kf2, const uint8_t *has_mp2,
                                  const float *level_sigma2_2, const float *F12, const float *Cw1, const float *R2w, const float *t2w,
                                  const ygzf_camera *cam2, int only_stereo, int check_orientation, int *match12, int *nmatches);

/* ---- Frame::ComputeBoW()   src/Frame.cc:495-500 -> ORBVocabulary::transform(features, BowVector, FeatureVector, levelsup = 4), i.e.
 *      DBoW2::TemplatedVocabulary<FORB::TDescriptor, FORB>::transform   Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1151-1283 ---
 * The vocabulary tree lives on the device.  ygzf_vocabulary_set uploads it as the reference's loaders build it (loadFromTextFile
 * :1362-1447, load :1672-1715): node 0 is the root, parent[i] the parent of node i (parent[0] ignored), desc the n_nodes x 32 centroids;
 * the children of a node are its child nodes in ascending node id (the order m_nodes[pid].children.push_back(nid) produces), a node
 * without children is a word; depth_levels = m_L.  ygzf_bow_transform propagates n descriptors down the tree (nearest child in Hamming
 * distance at every level, first minimum wins, :1262-1273) and returns per descriptor the LEAF NODE id it lands in (the caller's
 * vocabulary object maps it to WordId and idf weight: m_nodes[leaf].word_id / .weight) and the node id at level m_L - levelsup (the
 * FeatureVector key; 0 = root when m_L - levelsup <= 0).  The BowVector / FeatureVector maps are assembled by the caller in feature order
 * (std::map containers of the reference; the class shell does it, host/ORBVocabularyDevice.h). */
int ygzf_vocabulary_set(ygzf_ctx *ctx, int n_nodes, int depth_levels, const int *parent, const uint8_t *desc);
int ygzf_bow_transform(ygzf_ctx *ctx, int n, const uint8_t *desc, int levelsup, int *leaf_node, int *level_node);

/* ---- ORBmatcher::SearchByProjection(Frame &F, const vector<MapPoint*> &vpMapPoints, const float th, bool checkLevel)
 *      src/ORBmatcher.cc:43-126 (Tracking::SearchLocalPoints, nnratio 0.8) ---------------------------------------------------
 * Per MapPoint i (fields set by Frame::isInFrustum, src/Frame.cc:413-419): track_in_view = mbTrackInView, is_bad = isBad()
 * (NULL = none), mp_has_obs = Observations() > 0 (NULL = all), proj_x/proj_y/proj_xr = mTrackProjX/Y/XR (proj_xr may be
 * NULL for monocular), view_cos = mTrackViewCos, scale_level = mnTrackScaleLevel, mp_desc = GetDescriptor().
 * owner (in/out) / match (out: index of the MapPoint written to F.mvpMapPoints[idx], -1 untouched) / *nmatches as above. */
int ygzf_search_by_projection_mappoints(ygzf_ctx *ctx, const ygzf_frame_view *F, const ygzf_camera *cam, int n_mp, const uint8_t *track_in_view,
                                        const uint8_t *is_bad, const uint8_t *mp_has_obs, const float *proj_x, const float *proj_y,
                                        const float *proj_xr, const float *view_cos, const int *scale_level, const uint8_t *mp_desc, float th,
                                        int check_level, float nnratio, uint8_t *owner, int *match, int *nmatches);

/* Batched, device-resident form chained behind ygzf_extract_batch_*: frame f of the batch is CurrentFrame, frame f-1 is
 * LastFrame (for f == 0: the last frame of the previous batch of this context, or an empty frame).  Identity relative
 * pose; every Last keypoint carries a MapPoint at its unit-depth back-projection ((x-cx)/fx, (y-cy)/fy, 1) whose
 * descriptor is the keypoint's own -- the synthetic scenario of the extract+match metric. */
int ygzf_match_batch_prev(ygzf_ctx *ctx, const ygzf_camera *cam, float th, int b_mono, int check_level, int check_orientation);
int ygzf_match_counts(ygzf_ctx *ctx, int *nmatches /* n_frames ints */);
/* Diagnostics: how many (Cur, Last) / (Frame, MapPoints) pairs of this context's matcher launches so far resolved their in-order ownership
 * (src/ORBmatcher.cc:1296-1345, :98-121) through the one-wave serial pass because the block-wide fixpoint ran out of rounds or list
 * extensions -- identical matches, but a crowded frame (th = 5 after relocalisation, thousands of features) silently loses the speed-up;
 * bench.py prints it.  Synchronises the context. */
int ygzf_match_fallbacks(ygzf_ctx *ctx, unsigned *pairs);
int ygzf_match_fetch(ygzf_ctx *ctx, int frame, int *cur_match, uint8_t *cur_owner, int cap);
/* All pairs of the last ygzf_match_batch_prev at once: row p of `match` (stride >= ygzf_max_keypoints ints per row) receives the match
 * array of pair p in full row length (entries past the frame's keypoint count are -1 or stale: read n_kp of them). */
int ygzf_match_fetch_all(ygzf_ctx *ctx, int *match, int stride);

/* ---- ygz::SparseImgAlign(max_level, min_level, n_iter = 10, GaussNewton).run(Frame *ref, Frame *cur, SE3f &TCR)
 *      include/SparseImageAlign.h:14-51, src/SparseImageAlign.cc:7-49 (+ NLLSSolver<6,SE3f>::optimizeGaussNewton,
 *      include/NLSSolver_impl.hpp:17-91) -------------------------------------------------------------------------------
 * Frames as plain arrays.  Tcw = Frame::mTcw as (qx qy qz qw tx ty tz) (Sophus SE3f storage order); levels[l] = tight
 * 8-bit image of pyramid level l (Frame::mvImagePyramid[l], step == cols); only levels min_level..max_level are read.
 * ref needs keys / mp_valid / outlier / mp_world; cur only Tcw and its pyramid.  inv_scale_factors = mvInvScaleFactors.
 * The reference indexes `iterations[level]` out of bounds for level > 5: n_iter applies to every level.
 * Outputs: TCR_out = T_cur_from_ref (same 7-float layout; untouched when ref->n == 0), *ret = n_meas_/16 (0 = failure,
 * src/Tracking.cc:2089), info[0] = Gauss-Newton linearisations run, info[1] = final chi2, H36 = the Hessian H_ of the
 * last iteration (getFisherInformation() = H36 / (5e-4*255*255)).  info and H36 may be NULL. */
typedef struct ygzf_sia_frame {
    int n;
    const ygzf_kp *keys;
    const uint8_t *mp_valid;       /* mvpMapPoints[i] != NULL && !isBad(); NULL = all valid */
    const uint8_t *outlier;        /* mvbOutlier; NULL = none */
    const float *mp_world;         /* n x 3 */
    float Tcw[7];
    int nlevels;
    const uint8_t *const *levels;
    const int *level_w, *level_h;
} ygzf_sia_frame;
int ygzf_sia_run(ygzf_ctx *ctx, const ygzf_sia_frame *ref, const ygzf_sia_frame *cur, const ygzf_camera *cam, const float *inv_scale_factors,
                 int max_level, int min_level, int n_iter, float *TCR_out, size_t *ret, float *info, float *H36);

/* Batched, device-resident form chained behind ygzf_extract_batch_* (BASELINE config 3: extract + match + align):
 * for batch frame f, ref = frame f-1 (f == 0: the last frame of the previous batch of this context; without one the entry
 * reports ret = 0), cur = frame f; features = the keypoints of ref with MapPoints at their unit-depth back-projection;
 * both poses identity.  Uses the pyramids already in HBM (levels >= 1: min_level >= 1, as Tracking's
 * SparseImgAlign(nLevels-1, 1)).  The first call makes the context keep the last pyramid across batches. */
int ygzf_align_batch_prev(ygzf_ctx *ctx, const ygzf_camera *cam, int max_level, int min_level, int n_iter);
int ygzf_align_fetch(ygzf_ctx *ctx, int frame, float *TCR_out, size_t *ret, float *info);

/* ---- ORBextractor::operator()(Frame*, vector<KeyPoint>&, OutputArray, DSO_KEYPOINT, leftEye)   src/ORBextractor.cc:1031-1127 -----
 * with ComputeKeyPointsDSOSingleLevel (:1275-1386: FAST-10 per grid cell at barrier 20 then 5, 20-px edge filter, occupancy of the
 * frame's existing keys, Shi-Tomasi score :1152-1187, 3 best per cell, grid-size retry loop) and the descriptors of existing + new keys.
 * The pyramid is computed from `img` (what the Frame constructor does before, src/Frame.cc:807-813).
 * keys: in  = the frame's n_existing keys (mvKeys: level-0 coordinates, octave = level they were found at),
 *       out = the same keys with IC_Angle recomputed (:1380-1383) followed by the new level-0 keys (size 7, response 0, octave 0).
 * desc: n_total x 32 bytes, row i belongs to keys[i] (:1101-1126).
 * grid_size: ORBextractor::mnGridSize, persistent across frames; pass a negative value for "not yet set" (:1298-1299).
 * Defined where the reference is not: score ties keep raster order; a NaN score ranks lowest; a frame without a single FAST corner
 * returns no new keys (the reference never leaves its loop); existing keys closer than 15 px to their level's border are rejected
 * (out-of-bounds reads in the reference). */
int ygzf_extract_dso(ygzf_ctx *ctx, const uint8_t *img, int w, int h, int stride, ygzf_kp *keys, int n_existing, int cap, uint8_t *desc,
                     int *grid_size, int *n_total);

/* ---- the same overload with method == FAST_KEYPOINT: ComputeKeyPointsFast   src/ORBextractor.cc:1045-1051, 1189-1273 -----------------
 * Per level Thirdparty/fast's FAST-10 (barrier iniThFAST) on the level minus a 20-px top / left margin, fast_corner_score_10,
 * fast_nonmax_3x3 (>=); one keypoint per 5x5-px cell of a level-0 grid: the corner with the largest Shi-Tomasi score over all levels
 * (the first one on ties); cells that hold one of the frame's own keys stay empty; IC_Angle, descriptors, pt *= scale[level].
 * keys in / out, desc, n_total as ygzf_extract_dso (new keys: size = (int)(31 * scale[level]), response = the Shi-Tomasi score).
 * The reference's author marks the function "has a bug which may corrupt the program, don't call it" (:1191); defined here where it is
 * undefined there: (i) it strips the margin on two sides only, so corners come within 3 px of the right / bottom borders, where IC_Angle and
 * the descriptor read outside the level -- those reads see BORDER_REFLECT_101 (what the extractor's own bordered pyramid holds);
 * (ii) a key or corner whose cell index falls outside the grid (image sizes that are no multiples of 5) is ignored; (iii) levels narrower
 * than 42 px or lower than 27 px are skipped. */
int ygzf_extract_fast_keypoint(ygzf_ctx *ctx, const uint8_t *img, int w, int h, int stride, ygzf_kp *keys, int n_existing, int cap, uint8_t *desc,
                               int *n_total);

/* ---- the same overload over the multi-level ComputeKeyPointsDSO   src/ORBextractor.cc:1388-1507 (its call is commented out at :1053) ----
 * Per level: grid of sqrt(h w / mnFeaturesPerLevel) px, libfast at iniThFAST then minThFAST per inner cell, 20-px edge filter, the
 * level-0-sized occupancy map indexed with level coordinates (:1466, kept), 2 best Shi-Tomasi corners per cell, every selected corner
 * occupies its pixel at once and for the rest of the call; retry with grid - 5 (down to 7) while the level has fewer than its share.
 * New keys: size 7, response 0, octave = level, pt *= scale[level].  grid_size: mnGridSize before / after (only the value after matters:
 * the function recomputes it per level).  Definitions as ygzf_extract_dso. */
int ygzf_extract_dso_multilevel(ygzf_ctx *ctx, const uint8_t *img, int w, int h, int stride, ygzf_kp *keys, int n_existing, int cap, uint8_t *desc,
                                int *grid_size, int *n_total);

/* Descriptors (and optionally IC_Angle) of keys that already exist in a frame -- the "existing ones" loop of the Frame overload,
 * src/ORBextractor.cc:1093-1106 -- on the pyramid of frame `frame` of the last extracted batch.  keys hold level-0 coordinates and the
 * octave they live on; the descriptor is taken at cvRound(pt * mvInvScaleFactor[octave]) on the blurred level.
 * recompute_angle = 0: the stored angle is used (ORBSLAM_KEYPOINT / FAST_KEYPOINT branches, :1096);
 * recompute_angle = 1: IC_Angle first (what ComputeKeyPointsDSOSingleLevel does to them, :1380-1383), returned in angles_out. */
int ygzf_describe_keys(ygzf_ctx *ctx, int frame, const ygzf_kp *keys, int n, int recompute_angle, float *angles_out, uint8_t *desc);

/* ---- Frame::ComputeStereoMatches()   src/Frame.cc:509-682 (the step right after stereo extraction) ------------------------
 * Per left keypoint: best Hamming match among the right keypoints whose row band covers its row (octave +-1, disparity range
 * [0, mbf/mb]), accepted below (TH_HIGH+TH_LOW)/2; 11x11 SAD of the centre-subtracted patches over +-5 px on the keypoint's pyramid
 * level of both eyes; parabola sub-pixel fit; mvuRight / mvDepth; finally matches with SAD >= 1.5*1.4*median are dropped.
 * u_right / depth: n_left floats each, -1 = no match.  Defined where the reference is not: an empty match list skips the median cut.
 * Host-array form (what the Frame member is bound to): both level-0 images, the pyramids are recomputed on the
 * device -- the extractors' host pyramids come from the same ComputePyramid.  The context must be created with max_batch >= 2. */
int ygzf_compute_stereo_matches(ygzf_ctx *ctx, const uint8_t *img_left, const uint8_t *img_right, int w, int h, int stride, int n_left,
                                const ygzf_kp *keys_left, const uint8_t *desc_left, int n_right, const ygzf_kp *keys_right, const uint8_t *desc_right,
                                float mb, float mbf, float *u_right, float *depth);
/* Batch-resident form: the last extracted batch holds (left, right) image pairs, frames 2p / 2p+1; keys, descriptors and pyramids stay
 * on the device.  ygzf_stereo_fetch copies pair p's mvuRight / mvDepth (as many as the left frame has keypoints). */
int ygzf_stereo_batch(ygzf_ctx *ctx, float mb, float mbf);
/* ONE (left, right) pair from host memory with its eyes on two contexts of one device, so that the right eye's upload runs beside the left eye's
 * kernels (a 3840x2160 pair is 16.6 MB on the link: a third of the call when both eyes go up before anything runs).  Results: the packed blocks of
 * ygzf_batch_fetch_packed for each eye (host_left / host_right, host_bytes each; the same offsets and row length for both), mvuRight / mvDepth of the
 * left keypoints in u_right / depth (ygzf_max_keypoints floats each).  Same bytes as ygzf_extract_batch_host(left, right) + ygzf_stereo_batch.
 * Page-locked frames and result buffers give the overlap; pageable ones work and serialise. */
int ygzf_stereo_pair_host(ygzf_ctx *left_ctx, ygzf_ctx *right_ctx, const uint8_t *left, const uint8_t *right, int w, int h, int row_pitch, float mb, float mbf,
                          void *host_left, void *host_right, size_t host_bytes, size_t *off_kps, size_t *off_desc, int *row_entries, float *u_right,
                          float *depth);
int ygzf_stereo_fetch(ygzf_ctx *ctx, int pair, float *u_right, float *depth, int cap);
/* all pairs at once: rows of `stride` (>= ygzf_max_keypoints) floats; row p's first n_kp[2 p] entries (left keypoints) are valid */
int ygzf_stereo_fetch_all(ygzf_ctx *ctx, float *u_right, float *depth, int stride);

/* ---- ORBmatcher::FindDirectProjection(KeyFrame *ref, Frame *curr, MapPoint *mp, Vector2f &px_curr, int &search_level)
 *      src/ORBmatcher.cc:1574-1602 with GetWarpAffineMatrix :1525-1548, GetBestSearchLevel / GetBilateralInterpUchar
 *      include/ORBmatcher.h:185-211, WarpAffine :1550-1572 and ygz::Align2D src/Align.cc:8-104    ----------------------------
 * Tracking::SearchLocalPointsDirect (src/Tracking.cc:2174-2326) calls it once per (MapPoint, observing KeyFrame) candidate; the device
 * form takes the independent candidates as one batch.  The images live in an HBM-resident cache: every KeyFrame (and the current frame)
 * is put once -- its pyramid is built on the device -- and referenced by slot afterwards.
 *   ygzf_image_cache_reserve : n_slots images of w x h (the extractor configuration of the context defines the pyramid)
 *   ygzf_image_cache_put     : upload a level-0 image into a slot and build its pyramid (asynchronous: `img` must stay valid until the
 *       next synchronising call on this context, e.g. ygzf_find_direct_projection_batch or ygzf_sync)
 *   ygzf_find_direct_projection_batch : candidate i = (ref_slot[i], ref_Tcw7[i] = ref->GetPose() as quaternion x,y,z,w + translation,
 *       ref_kp[i] = ref->mvKeys[mp->GetObservations()[ref]], mp_world[i] = mp->GetWorldPos()); cur_Tcw7 = curr->mTcw;
 *       px_curr (n x 2, in/out): initial guess (mTrackProjX/Y) -> refined pixel; search_level / success (the bool result) out;
 *       patches_with_border (nullable, n x 100): the warped 10x10 reference patches, for tests.
 * Results equal the CPU definition bit for bit (one thread per candidate keeps the reference's sequential accumulation order). */
int ygzf_image_cache_reserve(ygzf_ctx *ctx, int n_slots, int w, int h);
/* SparseImgAlign::run on two image-cache slots (pyramids already on the device: the slots' level-0 images and what the resize kernel built
 * from them -- the same bytes the extractor's pyramid holds).  `ref` supplies keys / mp_world / mp_valid / outlier / Tcw as in ygzf_sia_run;
 * its level arrays are not read.  Saves the upload of both host pyramids on every call (the reference frame of one call was the current
 * frame of the previous one). */
int ygzf_sia_run_cached(ygzf_ctx *ctx, int ref_slot, int cur_slot, const ygzf_sia_frame *ref, const float *cur_Tcw7, const ygzf_camera *cam,
                        const float *inv_scale_factors, int max_level, int min_level, int n_iter, float *TCR_out, size_t *ret, float *info,
                        float *H36);

int ygzf_image_cache_put(ygzf_ctx *ctx, int slot, const uint8_t *img, int w, int h, int stride);
/* The same from another context of the same device that still holds the image: its level 0 and pyramid (built by its last
 * ygzf_compute_pyramid / ygzf_extract / ygzf_extract_resident, no other image operation since) are copied device to device -- no upload, no
 * second pyramid.  A Frame's image reaches the device once, in ORBextractor::ComputePyramid (src/Frame.cc:807); SparseImgAlign::run and
 * FindDirectProjection take it from there.  YGZF_ERR_STATE when the source holds no image of the cache's size (the caller then uploads with
 * ygzf_image_cache_put); ygzf_has_resident_image asks without an error.  The copy is ordered behind the source's pending work and ahead of
 * its later work. */
int ygzf_image_cache_put_resident(ygzf_ctx *cache_ctx, int slot, ygzf_ctx *src_ctx);
int ygzf_has_resident_image(const ygzf_ctx *ctx, int w, int h);
int ygzf_find_direct_projection_batch(ygzf_ctx *ctx, const ygzf_camera *cam, int cur_slot, const float *cur_Tcw7, int n, const int *ref_slot,
                                      const float *ref_Tcw7, const ygzf_kp *ref_kp, const float *mp_world, float *px_curr, int *search_level,
                                      uint8_t *success, uint8_t *patches_with_border);

/* ---- Thirdparty/fast (Rosten FAST-10/16), replaced outright: fast::fast_corner_detect_10_sse2 + fast::fast_corner_score_10
 *      + fast::fast_nonmax_3x3  (Thirdparty/fast/include/fast/fast.h:19-29; called at src/ORBextractor.cc:1220-1235,
 *      :1330-1340, :1440-1450) on the window [x0,x0+w) x [y0,y0+h) of a host image -----------------------------------------
 * xy: corners in raster order as (x, y) int16 pairs relative to the window origin (fast::fast_xy); scores[i] = largest
 * barrier for which corner i is still a corner; nonmax_idx = indices into the corner list that survive the 3x3 non-maximum
 * suppression (suppressed if any 8-neighbour corner scores >= own).  scores / nonmax_idx / n_nonmax may be NULL.
 * Domain as libfast: x,y in [3,w-3) x [3,h-3); for w < 22 the reference falls back to its plain detector, which scans the
 * WHOLE window and reads 3 px around it, so such windows must lie 3 px inside the image (error otherwise); h < 7: nothing. */
int ygzf_fast10(ygzf_ctx *ctx, const uint8_t *img, int img_w, int img_h, int stride, int x0, int y0, int w, int h, int barrier, int16_t *xy,
                int *scores, int *nonmax_idx, int cap, int *n_corners, int *n_nonmax);

/* ---- multi-GPU: frames dealt over the devices of one node ----------------------------------------------------------------------
 * Frames are independent and SearchByProjection couples frame t only with t - 1, so the split is "one unit per GPU, round-robin": a unit =
 * `unit` consecutive frames that must stay together (1: single frames, 2: frame pairs / stereo pairs, k: clips of k frames); unit u is
 * processed by devices[u % n_devices].  One host thread + one context + page-locked staging per device slot; results come back in INPUT order;
 * no collective, no peer copy.  `devices` may name a device more than once (several slots on one GPU: how the 1-GPU test box exercises
 * the split); a device index that does not exist is YGZF_ERR_NO_DEVICE.
 * ygzf_mgpu_extract_match: ORBextractor::operator() of every frame and, when match / nmatches are given, SearchByProjection(frame i, frame
 * i - 1) inside every unit with the synthetic scenario of ygzf_match_batch_prev (first frame of a unit: nmatches = -1, match row = -1).
 * kps / desc / match hold n_frames rows of `stride` (>= ygzf_mgpu_keypoint_stride) entries; frame f's first n_kp[f] entries are valid.
 * ygzf_mgpu_extract_stereo: the frames are (left, right) pairs (frames 2p, 2p + 1; BASELINE.json's 3840x2160 stereo configuration): extraction of
 * both eyes and Frame::ComputeStereoMatches of every pair (ygzf_stereo_batch), a pair never leaves its device.  u_right / depth hold n_frames / 2
 * rows of `stride` floats, row p's first n_kp[2 p] entries valid (-1: no match).
 * Inside a device slot the frames go through in unit-aligned chunks on two alternating contexts: while one chunk is in its kernels the next
 * one's frames go up and the previous one's results come down.  Frames that already lie in page-locked host memory (hipHostMalloc /
 * hipHostRegister -- detected with hipPointerGetAttributes) are copied to the device from where they lie; pageable frames are gathered into the
 * slot's own page-locked staging first.  The slot threads are bound to the CPUs of their device's NUMA node when sysfs reports one
 * (YGZF_MGPU_NUMA=0 switches that off). */
typedef struct ygzf_mgpu ygzf_mgpu;
int ygzf_mgpu_create(const int *devices, int n_devices, const ygzf_extractor_cfg *cfg, int max_width, int max_height, int max_frames_per_device,
                     ygzf_mgpu **out);
void ygzf_mgpu_destroy(ygzf_mgpu *m);
const char *ygzf_mgpu_last_error(const ygzf_mgpu *m);
int ygzf_mgpu_device_count(const ygzf_mgpu *m);
int ygzf_mgpu_keypoint_stride(const ygzf_mgpu *m);
int ygzf_mgpu_slot_of_frame(const ygzf_mgpu *m, int frame, int unit);   /* which device slot processes this frame */
int ygzf_mgpu_extract_match(ygzf_mgpu *m, const uint8_t *frames, int n_frames, int w, int h, int row_pitch, size_t frame_stride, int unit,
                            const ygzf_camera *cam, float th, int b_mono, int check_level, int check_orientation, ygzf_kp *kps, uint8_t *desc,
                            int *n_kp, int stride, int *match, int *nmatches);
int ygzf_mgpu_extract_stereo(ygzf_mgpu *m, const uint8_t *frames, int n_frames, int w, int h, int row_pitch, size_t frame_stride, float mb, float mbf,
                             ygzf_kp *kps, uint8_t *desc, int *n_kp, int stride, float *u_right, float *depth);
/* Page-locked, device-visible host memory for frames (what ygzf_mgpu_* and ygzf_extract_batch_host* copy from at the link's full rate, without a
 * staging copy) for callers that do not link the HIP runtime themselves: hipHostMalloc / hipHostFree on the given device.  NULL on failure. */
void *ygzf_alloc_host(int device, size_t bytes);
/* Binds the CALLING thread to the CPUs of the NUMA node the device hangs on (/sys/bus/pci/devices/<bus id>/numa_node), intersected with the
 * affinity it already has -- what a one-process-per-GPU caller (bench.py --gpus N: BASELINE.json configs[3], configs[4]) does once per rank
 * BEFORE it page-locks its frame buffers, so that they are first touched on the socket the GPU's link ends on (ygzf_mgpu_* does the same for its
 * own slot threads and never touches the caller's).  Returns the number of CPUs bound to, 0 when there is no NUMA information or nothing to do,
 * YGZF_ERR_NO_DEVICE for a device that does not exist. */
int ygzf_bind_host_thread_to_device(int device);
void ygzf_free_host(void *p);
/* What the host's memory can feed (no GPU work): n_threads threads, bound to `device`'s NUMA node when device >= 0, stream-read page-locked
 * buffers of bytes_per_thread each for `seconds`; *gb_per_s = bytes read / time.  One GPU's transfers-included rate reads its frames at the
 * link's rate (~55 GB/s); an 8-GPU node wants eight times that from DRAM at once (bench.py reports both). */
int ygzf_host_stream_probe(int device, int n_threads, size_t bytes_per_thread, double seconds, double *gb_per_s);
int ygzf_mgpu_chunk_frames(const ygzf_mgpu *m);   /* frames per chunk inside a slot (units up to this size alternate between the slot's two contexts) */

/* ---- timing / profiling helpers for bench.py -------------------------------------------------------------------------
 * HIP events recorded on the context stream (the stream every kernel of this context is launched on). */
int ygzf_timer_start(ygzf_ctx *ctx);
int ygzf_timer_stop(ygzf_ctx *ctx, float *elapsed_ms);
/* Per-kernel accumulation: when enabled every kernel launch is bracketed by its own event pair. */
int ygzf_profile_enable(ygzf_ctx *ctx, int on);
/* Returns the number of kernel kinds; fills up to cap entries (name pointers are static strings). */
int ygzf_profile_read(ygzf_ctx *ctx, const char **names, float *total_ms, int *launches, int cap);
int ygzf_profile_reset(ygzf_ctx *ctx);
/* Raw hipStream_t of the context (as void*), for callers that record their own events. */
void *ygzf_stream(ygzf_ctx *ctx);

#ifdef __cplusplus
}
#endif
#endif /* YGZF_H */
