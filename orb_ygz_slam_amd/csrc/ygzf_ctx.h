// ygzf_ctx.h -- the context of libygzf and the small helpers every translation unit of the C ABI shares (product code, internal: nothing here
// is part of include/ygzf.h).  The entry points live in
//   ygzf_api.hip         context, geometry tables, buffers, the extractor's launch sequence, batch fetches, timers / profile
//   ygzf_api_match.hip   ORBmatcher: SearchByProjection / SearchByBoW / SearchForInitialization / SearchForTriangulation, the Frame grid, isInFrustum,
//                        distinctive descriptors, the vocabulary
//   ygzf_api_align.hip   SparseImgAlign (one pair, cached, resident batch), the image cache, FindDirectProjection
//   ygzf_api_detect.hip  Thirdparty/fast replacement, DSO / FAST_KEYPOINT detectors, descriptors of existing keys
//   ygzf_api_stereo.hip  Frame::ComputeStereoMatches
#ifndef YGZF_CTX_H
#define YGZF_CTX_H
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <mutex>

#include "kernels.h"

#define YGZF_HIDDEN __attribute__((visibility("hidden")))

namespace ygzf {
YGZF_HIDDEN int cv_round_host(double v);
enum KernelKind { KK_PYR = 0, KK_FAST, KK_OCTREE, KK_DESCRIBE, KK_HAMMING, KK_BACKPROJ, KK_MATCH, KK_SIA, KK_FAST10, KK_DSO, KK_STEREO, KK_DIRECT, KK_BOW, KK_FRUSTUM, KK_DISTINCTIVE, KK_BOWNODES, KK_GRID, KK_FASTQ, KK_TRI, KK_FASTP, KK_COUNT };
static const char *kKernelNames[KK_COUNT] = {"k_pyr_resize", "k_fast_tab", "k_octree", "k_describe", "k_hamming_pairs",
                                             "k_backproject_unit", "k_match_last", "k_sia_run", "k_f10_*", "k_dso_cells", "k_stereo_*", "k_direct_projection", "k_bow_descend", "k_frustum", "k_distinctive", "k_bow_nodes", "k_features_in_area", "k_fast_quads", "k_tri_nodes", "k_fast_tab_persist"};

struct Geometry {
    int w = 0, h = 0;
    std::vector<LevelGeom> lv;
    std::vector<int> xofs, yofs;
    std::vector<short> xalpha, ybeta;
    std::vector<PyrColRec> pyrCols;      // k_pyr_resize_tiled's tables (levels >= 1, LevelGeom::pyrCol / pyrRow / pyrTile index them)
    std::vector<PyrRowRec> pyrRows;
    std::vector<PyrTileRec> pyrTiles;
    long long pyrBytes = 0;   // per frame, levels >= 1
    std::vector<PyrStripPlan> pyrPlan;   // k_pyr_strips: one entry per strip (empty: this geometry takes one launch per level)
    int pyrStripOffA = 0, pyrStripOffB = 0;
    int pyrBase = 0;                     // the strips start from this level (0: the image; > 0: levels 1 .. pyrBase come from k_pyr_resize_tiled first)
    PyrStripLevel pyrLevels[kMaxLevels];
    size_t pyrStripLds = 0;
    int totalCells = 0, maxCellsPerLevel = 0;
    int totalGroups = 0, fastSmapRows = 3, fastWinPitch = 16, fastWinRows = 7, fastQuadCap = 4;   // 2x2 cell groups of k_fast_quads
    int fastWCellMax = 1;
    std::vector<FastCellRec> fastCells;   // [group * 4 + position]: k_fast_tab's per-cell records
    long long totalSlots = 0;
    long long candStride = 0;
    int kpStride = 0, kpCapMax = 0;
};

}  // namespace ygzf

using namespace ygzf;

struct ygzf_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    Tables tab;
    int maxW = 0, maxH = 0, maxBatch = 0;
    Geometry geo;
    std::string err;
    // device buffers (grow-only)
    struct Buf {
        void *p = nullptr;
        size_t bytes = 0;
    };
    Buf dGeom, dXofs, dXalpha, dYofs, dYbeta, dImg0, dPyr, dCellCnt, dSlots, dK0, dV0, dK1, dV1, dXY, dLvlXY, dLvlScore,
        dLvlCnt, dLvlBase, dLvlCand, dOutKp, dOutDesc, dOutCnt, dTmpA, dTmpB, dTmpC, dWorld, dOwner, dMatch, dNMatch, dPoses, dQp,
        dGen[12], dVoc[3], dBow[3], dSia[8], dProcOrder, dF10[6], dSpill, dCarryPyr, dAl[5], dDso[8], dSt[6], dStBins, dCacheImg, dCachePyr, dDir[8], dFr[6], dSplitCnt, dSplitX, dPyrPlan, dPyrCols, dPyrRows, dPyrTiles, dOctHist;
    int vocNodes = 0, vocLevels = 0;
    int cacheSlots = 0, cacheW = 0, cacheH = 0, cachePitch = 0;
    long long cachePyrBytes = 0;
    std::vector<unsigned char> cacheFilled;
    int lastStereoPairs = 0;
    bool alignCarry = false;      // keep the last frame's pyramid across batches (enabled by the first ygzf_align_batch_prev)
    bool carryPyrValid = false;
    int lastAlignPairs = 0;
    std::vector<unsigned char> alKey;   // cache key of the uploaded SiaLevel tables
    bool carryValid = false;
    int lastMatchPairs = 0;
    int identityPoses = 0;
    void *identityPosesPtr = nullptr;
    size_t octLds = 0;
    int octLdsCand = 0;
    bool octGlobalNodes = false;
    // histogram plan of the octree (levels with tens of thousands of candidates): launches of consecutive levels, each with its own LDS allotment
    struct OctGroup { int l0 = 0, n = 0, cap = 0, regionInts = 0, histBins = 0; size_t lds = 0; };
    std::vector<OctGroup> octGroups;
    OctGroup octSmall;                     // all levels in ONE histogram-plan launch: launches of a few frames (see run_extract)
    bool haveOctSmall = false;
    bool carryOff = false;     // ygzf_set_carry_previous(0): extractions do not carry the previous batch's last frame into slot 0 ...
    bool slot0Stale = false;   // ... and slot 0 no longer holds it: the batch matchers refuse until an extraction has carried again
    size_t octHistWords = 0;
    int octDoneTarget = 0;     // what the launches so far have brought every (level, frame) counter of dOctHist to   // layout dOctHist's counters were last cleared for (k_octree's helper workgroups)
    static constexpr int octSmallWgs = 128;   // launches of up to this many workgroups take it (752x480: 16 frames 0.211 against 0.221 ms, 64 frames 0.439 against 0.430)
    Buf dOctNodes;
    // FAST threshold plan (extract_kernels.hip, fast_cell): 0 = chosen per batch from the statistics the kernel leaves behind, 1 = one pass at
    // minTh, 2 = iniTh first.  Identical results either way.
    int fastPlan = 0;
    bool fastIniFirst = false;
    unsigned fastLaunches = 0;             // statistics are collected on the first launches and on every 8th one after that
    double fastExtraRounds = 0.0;          // score rounds beyond the first per cell, last measured by the one-pass plan
    double fastCornerQuadsPerRun = 0.0, fastRunsPerCell = 0.0;   // last sampled: corner-bearing quads per pass-1 run, pass-1 runs per cell (ygzf_get_fast_stats)
    Buf dFastStats;
    Buf dFastCells;                        // FastCellRec table of the current geometry (k_fast_tab)
    Buf dMatchStat;                        // one counter: pairs that fell back to the matcher's one-wave pass (ygzf_match_fallbacks)
    Buf dPack;                             // inputs + outputs of a one-frame entry point, one copy each way (PackedTransfer)
    int fastKernel = 0;                    // ygzf_fast_kernel: 0 chosen per geometry, 1 k_fast_quads (register staging), 2 k_fast_tab (cell table + LDS-DMA)
    unsigned *hFastStats = nullptr;        // page-locked mirror, refreshed by an asynchronous copy after every FAST launch
    Buf dFastCtr;                          // eight draw counters of k_fast_tab_persist (one per XCD), zeroed before every launch
    int cuCount = 256;
    Buf dResPack;                          // [counts | keypoint rows | descriptor rows] of a small launch, contiguous (ygzf_batch_fetch_packed)
    bool carryLaunched = false;            // k_carry_slot already queued for the extraction being set up (ahead of its upload)
    Buf dUpStage;                          // linear landing area of uploads that are re-pitched on the device (upload_rows)
    bool pyrHeld = false;                  // dImg0 / dPyr frame 0 hold ONE image (pyrHeldW x pyrHeldH) and its complete pyramid (ygzf_image_cache_put_resident)
    int pyrHeldW = 0, pyrHeldH = 0;
    hipEvent_t evShare = nullptr, evPyrDone = nullptr;
    bool evPyrDoneValid = false;
    bool pyrResident = false;              // dImg0 / dPyr frame 0 hold the image and pyramid of the last ygzf_compute_pyramid (pyrResW x pyrResH)
    int pyrResW = 0, pyrResH = 0;
    // one-frame uploads from pageable caller memory (a cv::Mat): the rows are copied into this page-locked, device-visible buffer by the host and
    // read from there by a kernel that writes them at the context's pitch -- the runtime's own pageable path (pin / stage / blit) cost ~50 us for a
    // 752x480 frame, this ~25.  evIn marks the moment the kernel has read the buffer (the next upload waits for it before overwriting).
    uint8_t *hIn = nullptr;                // (one tight frame: at most maxW x maxH bytes, apply_geometry refuses anything larger)
    void *hInDev = nullptr;
    size_t hInBytes = 0;
    hipEvent_t evIn = nullptr;
    bool evInPending = false;
    uint8_t *hStage = nullptr;             // page-locked staging for results that go back to pageable caller memory in many small pieces
    size_t hStageBytes = 0;
    void *hStageDev = nullptr;             // the same memory as the device addresses it (kernels write small results straight into it)
    // batch state
    int lastFrames = 0;
    FrameSet lastFs{};
    int img0Pitch = 0;
    // timing
    hipEvent_t tStart = nullptr, tStop = nullptr;
    // ygzf_set_extract_ahead: ygzf_compute_pyramid queues the extraction behind the pyramid and reads the levels back on a second stream
    // one-frame pyramid chain as a captured graph: seven dependent launches cost the host more than the kernels take (pyramid_chain)
    bool useGraphs = !debug_flag("nograph");
    int pyrStripFrames = (int) forced("pyr_strip_frames", 16);   // k_pyr_strips up to this many frames per launch (0: never)
    PyrChainGraph pyrGraph = {};
    const void *pyrGraphKey[6] = {nullptr};   // geometry tables + size the graph was built for
    bool extractAhead = false, aheadPending = false;
    hipStream_t streamCopy = nullptr;
    hipEvent_t evPyramid = nullptr;
    bool profile = false;
    // debugging aids, read once per context (never on the launch path): synchronise after every kernel and name it on stderr /
    // phase clocks of the octree, matcher and aligner kernels printed to stderr
    bool debugSync = debug_flag("sync");
    // the plans below are the library's own choice unless YGZF_FORCE pins them (tests: every plan against the oracle; ygzf_internal.h)
    // (per-call switches, read once when the context is created like the ones below)
    bool linkKernels = forced("fetch_kernel", 1) != 0, pyrLink = forced("pyr_link", 1) != 0;
    int uploadKernelFrames = (int) forced("upload_kernel_frames", 2);
    size_t uploadKernelBytes = (size_t) forced("upload_kernel_bytes", 16l << 20);
    int octHelpersForced = (int) forced("oct_helpers", -1), octHelperSpin = (int) forced("oct_helper_spin", 1 << 16);
    int fastPersistMode = (int) forced("fast_persist", -1), fastPersistWgs = (int) forced("fast_persist_wgs", 8);
    int matchSplit = (int) forced("match_split", 0);   // 0 automatic, 1 off, n workgroups per pair
    int matchFixedLanes = forced_is("match_lanes", "fixed");   // eight lanes per query whatever the launch
    int matchFence = (int) forced("match_fence", 0);   // 1: full fences around the matcher's hand-over
    int matchSerial = (int) forced("match_serial", 0);   // 1: the one-wave in-order pass instead of the fixpoint; 2: fixpoint that hands over at the first exhausted list (tests)
    bool siaPerLevel = forced("sia_precompute", 1) != 0;   // reference patches of all levels in a kernel of their own (0: inside k_sia_run, as until round 4)
    bool octDebug = debug_flag("oct"), matchDebug = debug_flag("match"), siaDebug = debug_flag("sia");
    struct Rec { int kind; hipEvent_t a, b; };
    std::vector<Rec> recs;
    std::vector<hipEvent_t> pool;
    float profMs[KK_COUNT] = {0};
    int profN[KK_COUNT] = {0};
};

// last failed ygzf_create of THIS thread (the reference constructs its left / right extractors from different threads)
extern YGZF_HIDDEN thread_local std::string g_create_err;

// Every error return goes through here.  Entry points queue asynchronous uploads from caller-owned or local host arrays and synchronise
// at their end: an early error return must not leave such a copy in flight behind a source that is about to disappear, so the stream is
// drained first (errors are not a fast path).
static int fail(ygzf_ctx *c, int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (c) {
        if (c->stream) (void) hipStreamSynchronize(c->stream);
        c->err = buf;
    } else {
        g_create_err = buf;
    }
    return code;
}

#define HIPCHECK(c, expr)                                                                                         \
    do {                                                                                                          \
        hipError_t _e = (expr);                                                                                   \
        if (_e != hipSuccess) return fail(c, YGZF_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), \
                                          __FILE__, __LINE__);                                                    \
    } while (0)

static int ensure(ygzf_ctx *c, ygzf_ctx::Buf &b, size_t bytes) {
    if (bytes <= b.bytes) return YGZF_OK;
    // a buffer that grows loses its contents: whatever ygzf_compute_pyramid left in the image / pyramid buffers is gone with them
    if (&b == &c->dImg0 || &b == &c->dPyr) { c->pyrResident = false; c->pyrHeld = false; }
    if (b.p) HIPCHECK(c, hipFree(b.p));
    b.p = nullptr;
    b.bytes = 0;
    HIPCHECK(c, hipMalloc(&b.p, bytes));
    b.bytes = bytes;
    return YGZF_OK;
}

static inline int align_up(int v, int a) { return (v + a - 1) / a * a; }

static int ensure_stage(ygzf_ctx *c, size_t bytes) {
    if (bytes <= c->hStageBytes) return YGZF_OK;
    if (c->hStage) HIPCHECK(c, hipHostFree(c->hStage));
    c->hStage = nullptr;
    c->hStageBytes = 0;
    HIPCHECK(c, hipHostMalloc((void **) &c->hStage, bytes, hipHostMallocMapped));
    c->hStageDev = nullptr;
    if (hipHostGetDevicePointer(&c->hStageDev, c->hStage, 0) != hipSuccess) { (void) hipGetLastError(); c->hStageDev = nullptr; }
    c->hStageBytes = bytes;
    return YGZF_OK;
}

// One-frame entry points hand over a dozen small host arrays and take a few back.  From pageable memory every hipMemcpyAsync is a staged copy
// of its own (10-20 us of runtime work apiece; twelve of them cost more than the kernel they feed): the arrays are packed into the context's
// page-locked staging area and cross the link as ONE copy each way.
// callers with more than this in flight keep their own copies (the staging area is page-locked memory); YGZF_FORCE=packed_max=<bytes> lowers it so
// that tests reach the large-transfer paths with ordinary frames
static const size_t kPackedMax = (size_t) forced("packed_max", 4l << 20);
struct PackedTransfer {
    ygzf_ctx *c;
    struct Seg { const void *src; void *dst; size_t bytes, off; };
    std::vector<Seg> in, out;
    size_t inBytes = 0, outBytes = 0;
    explicit PackedTransfer(ygzf_ctx *c_) : c(c_) {}
    static size_t al(size_t b) { return (b + 255) & ~(size_t) 255; }
    size_t add_in(const void *src, size_t bytes) { const size_t o = inBytes; in.push_back({src, nullptr, bytes, o}); inBytes += al(bytes); return o; }
    size_t add_out(void *dst, size_t bytes) { const size_t o = outBytes; out.push_back({nullptr, dst, bytes, o}); outBytes += al(bytes); return o; }
    // device layout: [inputs | outputs] in c->dPack; returns the base
    // A transfer beyond kPackedMax (a KeyFrame with tens of thousands of features) does not grow the page-locked staging area: its arrays cross
    // one by one from / to the caller's own memory -- same device layout, so the kernels' pointers do not care which way the bytes came.
    bool direct() const { return inBytes + outBytes > kPackedMax; }
    // the packed blocks cross the link by a kernel (the staging area as the device addresses it) unless YGZF_FORCE=fetch_kernel=0
    bool link_kernel() const { return c->hStageDev != nullptr && c->linkKernels; }
    int upload(uint8_t **dBase) {
        int rc;
        if ((rc = ensure(c, c->dPack, inBytes + outBytes + 256))) return rc;
        if (direct()) {
            for (const Seg &s : in)
                if (s.bytes) HIPCHECK(c, hipMemcpyAsync((uint8_t *) c->dPack.p + s.off, s.src, s.bytes, hipMemcpyHostToDevice, c->stream));
        } else {
            if ((rc = ensure_stage(c, inBytes + outBytes + 256))) return rc;
            for (const Seg &s : in)
                if (s.bytes) memcpy(c->hStage + s.off, s.src, s.bytes);
            if (inBytes && link_kernel()) { ygzf::launch_link_copy(c->stream, c->dPack.p, c->hStageDev, inBytes); HIPCHECK(c, hipGetLastError()); }
            else if (inBytes) HIPCHECK(c, hipMemcpyAsync(c->dPack.p, c->hStage, inBytes, hipMemcpyHostToDevice, c->stream));
        }
        *dBase = (uint8_t *) c->dPack.p;
        return YGZF_OK;
    }
    uint8_t *d_out(size_t off) const { return (uint8_t *) c->dPack.p + inBytes + off; }
    int download() {   // one copy back, then scattered to the caller's arrays; synchronises the stream
        if (direct()) {
            for (const Seg &s : out)
                if (s.bytes && s.dst) HIPCHECK(c, hipMemcpyAsync(s.dst, d_out(s.off), s.bytes, hipMemcpyDeviceToHost, c->stream));
            HIPCHECK(c, hipStreamSynchronize(c->stream));
            return YGZF_OK;
        }
        if (outBytes && link_kernel()) { ygzf::launch_link_copy(c->stream, (uint8_t *) c->hStageDev + inBytes, (uint8_t *) c->dPack.p + inBytes, outBytes); HIPCHECK(c, hipGetLastError()); }
        else if (outBytes) HIPCHECK(c, hipMemcpyAsync(c->hStage + inBytes, (uint8_t *) c->dPack.p + inBytes, outBytes, hipMemcpyDeviceToHost, c->stream));
        HIPCHECK(c, hipStreamSynchronize(c->stream));
        for (const Seg &s : out)
            if (s.bytes && s.dst) memcpy(s.dst, c->hStage + inBytes + s.off, s.bytes);
        return YGZF_OK;
    }
};

// ---- per-kernel event bracketing ------------------------------------------------------------------------------------
static hipEvent_t take_event(ygzf_ctx *c) {
    if (!c->pool.empty()) {
        hipEvent_t e = c->pool.back();
        c->pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    (void) hipEventCreate(&e);
    return e;
}
struct ProfScope {
    ygzf_ctx *c;
    int kind;
    hipEvent_t a = nullptr, b = nullptr;
    hipStream_t s;
    ProfScope(ygzf_ctx *c_, int k, hipStream_t s_ = nullptr) : c(c_), kind(k), s(s_ ? s_ : c_->stream) {
        if (c->profile) {
            a = take_event(c);
            b = take_event(c);
            (void) hipEventRecord(a, s);
        }
    }
    ~ProfScope() {
        if (c->debugSync) {
            fprintf(stderr, "[ygzf] %s ...", kKernelNames[kind]);
            const hipError_t e = hipStreamSynchronize(s);
            fprintf(stderr, " %s\n", hipGetErrorString(e));
        }
        if (c->profile) {
            (void) hipEventRecord(b, s);
            c->recs.push_back({kind, a, b});
        }
    }
};
static void drain_profile(ygzf_ctx *c) {
    if (c->recs.empty()) return;
    (void) hipStreamSynchronize(c->stream);
    for (auto &r : c->recs) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) {
            c->profMs[r.kind] += ms;
            c->profN[r.kind]++;
        }
        c->pool.push_back(r.a);
        c->pool.push_back(r.b);
    }
    c->recs.clear();
}

static inline ygzf::PyrTabs pyr_tabs(const ygzf_ctx *c) {
    return ygzf::PyrTabs{(const int *) c->dXofs.p, (const short *) c->dXalpha.p, (const int *) c->dYofs.p, (const short *) c->dYbeta.p,
                         (const ygzf::PyrColRec *) c->dPyrCols.p, (const ygzf::PyrRowRec *) c->dPyrRows.p, (const ygzf::PyrTileRec *) c->dPyrTiles.p};
}

// ---- defined in ygzf_api.hip, used by the other translation units ---------------------------------------------------------------------
YGZF_HIDDEN int build_geometry(ygzf_ctx *c, int w, int h, ygzf::Geometry &G);
YGZF_HIDDEN int apply_geometry(ygzf_ctx *c, int w, int h, int nFrames);
YGZF_HIDDEN int pyramid_chain(ygzf_ctx *c, const ygzf::FrameSet &fs, int nFrames);
YGZF_HIDDEN int stereo_across(ygzf_ctx *l, ygzf_ctx *r, float mb, float mbf);   // ygzf_api_stereo.hip
YGZF_HIDDEN int mark_pyramid_done(ygzf_ctx *c);
YGZF_HIDDEN int run_extract(ygzf_ctx *c, const ygzf::FrameSet &fs, int nFrames, bool pyramidReady = false, bool pyramidCarried = false);
YGZF_HIDDEN int upload_rows(ygzf_ctx *c, void *dst, size_t dstPitch, const uint8_t *src, size_t srcPitch, int w, size_t rows);
YGZF_HIDDEN int upload_frames(ygzf_ctx *c, const uint8_t *imgs, int nFrames, int w, int h, int row_pitch, size_t frame_stride, ygzf::FrameSet *fs);
#endif
