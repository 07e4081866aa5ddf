// ygzf_api.hip -- the C ABI of libygzf (include/ygzf.h): context, geometry tables, buffer management and the launch
// sequence of the extractor hot path.  Product code: no CPU fallback, nothing from oracle/ is included or linked.
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <mutex>

#include "kernels.h"

namespace ygzf {

int cv_round_host(double v) { return (int) std::nearbyint(v); }  // cvRound: round half to even

// ORBextractor::ORBextractor, reference src/ORBextractor.cc:412-470 (scale tables, per-level quota, umax)
void Tables::init(const ygzf_extractor_cfg &c) {
    cfg = c;
    const int L = c.nlevels;
    scale.assign(L, 1.f);
    sigma2.assign(L, 1.f);
    invScale.assign(L, 1.f);
    invSigma2.assign(L, 1.f);
    for (int i = 1; i < L; i++) {
        scale[i] = scale[i - 1] * c.scale_factor;
        sigma2[i] = scale[i] * scale[i];
    }
    for (int i = 0; i < L; i++) {
        invScale[i] = 1.0f / scale[i];
        invSigma2[i] = 1.0f / sigma2[i];
    }
    nFeat.assign(L, 0);
    const float factor = 1.0f / c.scale_factor;
    float want = c.nfeatures * (1 - factor) / (1 - (float) std::pow((double) factor, (double) L));
    int sum = 0;
    for (int l = 0; l < L - 1; l++) {
        nFeat[l] = cv_round_host(want);
        sum += nFeat[l];
        want *= factor;
    }
    nFeat[L - 1] = std::max(c.nfeatures - sum, 0);
    // circular patch row ends (:455-469)
    int v, v0;
    const int vmax = (int) std::floor(kHalfPatch * std::sqrt(2.f) / 2 + 1);
    const int vmin = (int) std::ceil(kHalfPatch * std::sqrt(2.f) / 2);
    const double hp2 = kHalfPatch * kHalfPatch;
    for (v = 0; v <= vmax; ++v) umax[v] = cv_round_host(std::sqrt(hp2 - v * v));
    for (v = kHalfPatch, v0 = 0; v >= vmin; --v) {
        while (umax[v0] == umax[v0 + 1]) ++v0;
        umax[v] = v0;
        ++v0;
    }
}

void Tables::levelSize(int w, int h, int level, int *lw, int *lh) const {  // :1131-1132
    const float s = invScale[level];
    *lw = cv_round_host((float) w * s);
    *lh = cv_round_host((float) h * s);
}

static inline short sat_short(float v) {
    int i = cv_round_host((double) v);
    return (short) (i < -32768 ? -32768 : i > 32767 ? 32767 : i);
}

enum KernelKind { KK_PYR = 0, KK_FAST, KK_OCTREE, KK_DESCRIBE, KK_HAMMING, KK_BACKPROJ, KK_MATCH, KK_SIA, KK_FAST10, KK_DSO, KK_STEREO, KK_DIRECT, KK_BOW, KK_FRUSTUM, KK_DISTINCTIVE, KK_BOWNODES, KK_GRID, KK_FASTQ, KK_TRI, KK_COUNT };
static const char *kKernelNames[KK_COUNT] = {"k_pyr_resize", "k_fast_tab", "k_octree", "k_describe", "k_hamming_pairs",
                                             "k_backproject_unit", "k_match_last", "k_sia_run", "k_f10_*", "k_dso_cells", "k_stereo_*", "k_direct_projection", "k_bow_descend", "k_frustum", "k_distinctive", "k_bow_nodes", "k_features_in_area", "k_fast_quads", "k_tri_nodes"};

struct Geometry {
    int w = 0, h = 0;
    std::vector<LevelGeom> lv;
    std::vector<int> xofs, yofs;
    std::vector<short> xalpha, ybeta;
    long long pyrBytes = 0;   // per frame, levels >= 1
    std::vector<PyrStripPlan> pyrPlan;   // k_pyr_strips: one entry per strip (empty: this geometry takes one launch per level)
    int pyrStripOffCol = 0, pyrStripOffA = 0, pyrStripOffB = 0;
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
        dGen[12], dVoc[3], dBow[3], dSia[8], dProcOrder, dF10[6], dSpill, dCarryPyr, dAl[5], dDso[8], dSt[6], dStBins, dCacheImg, dCachePyr, dDir[8], dFr[6], dSplitCnt, dSplitX, dPyrPlan;
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
    int octSmallWgs = getenv("YGZF_OCT_SMALL_WGS") ? atoi(getenv("YGZF_OCT_SMALL_WGS")) : 128;   // launches of up to this many workgroups take it (A/B runs; 752x480: 16 frames 0.211 against 0.221 ms, 64 frames 0.439 against 0.430)
    Buf dOctNodes;
    // FAST threshold plan (extract_kernels.hip, fast_cell): 0 = chosen per batch from the statistics the kernel leaves behind, 1 = one pass at
    // minTh, 2 = iniTh first.  Identical results either way.
    int fastPlan = 0;
    bool fastIniFirst = false;
    unsigned fastLaunches = 0;             // statistics are collected on the first launches and on every 8th one after that
    double fastExtraRounds = 0.0;          // score rounds beyond the first per cell, last measured by the one-pass plan
    Buf dFastStats;
    Buf dFastCells;                        // FastCellRec table of the current geometry (k_fast_tab)
    Buf dMatchStat;                        // one counter: pairs that fell back to the matcher's one-wave pass (ygzf_match_fallbacks)
    Buf dPack;                             // inputs + outputs of a one-frame entry point, one copy each way (PackedTransfer)
    int fastKernel = 0;                    // ygzf_fast_kernel: 0 chosen per geometry, 1 k_fast_quads (register staging), 2 k_fast_tab (cell table + LDS-DMA)
    unsigned *hFastStats = nullptr;        // page-locked mirror, refreshed by an asynchronous copy after every FAST launch
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
    // batch state
    int lastFrames = 0;
    FrameSet lastFs{};
    int img0Pitch = 0;
    // timing
    hipEvent_t tStart = nullptr, tStop = nullptr;
    // ygzf_set_extract_ahead: ygzf_compute_pyramid queues the extraction behind the pyramid and reads the levels back on a second stream
    // one-frame pyramid chain as a captured graph: seven dependent launches cost the host more than the kernels take (pyramid_chain)
    bool useGraphs = getenv("YGZF_NO_GRAPH") == nullptr;
    int pyrStripFrames = getenv("YGZF_PYR_STRIP_FRAMES") ? atoi(getenv("YGZF_PYR_STRIP_FRAMES")) : 16;   // k_pyr_strips up to this many frames per launch (0: never)
    PyrChainGraph pyrGraph = {};
    const void *pyrGraphKey[6] = {nullptr};   // geometry tables + size the graph was built for
    bool extractAhead = false, aheadPending = false;
    hipStream_t streamCopy = nullptr;
    hipEvent_t evPyramid = nullptr;
    // ygzf_set_stream_partition: the latency-bound kernels of the chain (k_octree, k_match_last) on a second stream restricted to a share of the
    // compute units, so that their long-lived, rarely-issuing workgroups do not take wave slots from the issue-bound kernels of other contexts
    hipStream_t streamFill = nullptr;
    int fillCUs = 0, mainMode = 0;
    hipEvent_t evHop[8] = {nullptr};
    unsigned hopSeq = 0;
    bool profile = false;
    // debugging aids, read once per context (never on the launch path): synchronise after every kernel and name it on stderr /
    // phase clocks of the octree, matcher and aligner kernels printed to stderr
    bool debugSync = getenv("YGZF_DEBUG_SYNC") != nullptr;
    int matchSplit = getenv("YGZF_MATCH_SPLIT") ? atoi(getenv("YGZF_MATCH_SPLIT")) : 0;   // 0 automatic, 1 off, n workgroups per pair (A/B runs)
    int matchFixedLanes = getenv("YGZF_MATCH_LANES") && !strcmp(getenv("YGZF_MATCH_LANES"), "fixed");   // (A/B runs)
    int matchFence = getenv("YGZF_MATCH_FENCE") ? atoi(getenv("YGZF_MATCH_FENCE")) : 0;   // 1: full fences around the matcher's hand-over (A/B runs)
    int matchSerial = getenv("YGZF_MATCH_SERIAL") ? atoi(getenv("YGZF_MATCH_SERIAL")) : 0;   // 1: the one-wave in-order pass instead of the fixpoint; 2: fixpoint that hands over at the first exhausted list (tests)
    bool siaPerLevel = !(getenv("YGZF_SIA_PRECOMPUTE") && atoi(getenv("YGZF_SIA_PRECOMPUTE")) == 0);   // reference patches of all levels in a kernel of their own (0: inside k_sia_run, as until round 4)
    bool octDebug = getenv("YGZF_OCT_DEBUG") != nullptr, matchDebug = getenv("YGZF_MATCH_DEBUG") != nullptr, siaDebug = getenv("YGZF_SIA_DEBUG") != nullptr;
    struct Rec { int kind; hipEvent_t a, b; };
    std::vector<Rec> recs;
    std::vector<hipEvent_t> pool;
    float profMs[KK_COUNT] = {0};
    int profN[KK_COUNT] = {0};
};

// last failed ygzf_create of THIS thread (the reference constructs its left / right extractors from different threads)
static thread_local std::string g_create_err;

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
    HIPCHECK(c, hipHostMalloc((void **) &c->hStage, bytes));
    c->hStageBytes = bytes;
    return YGZF_OK;
}

// One-frame entry points hand over a dozen small host arrays and take a few back.  From pageable memory every hipMemcpyAsync is a staged copy
// of its own (10-20 us of runtime work apiece; twelve of them cost more than the kernel they feed): the arrays are packed into the context's
// page-locked staging area and cross the link as ONE copy each way.
// callers with more than this in flight keep their own copies (the staging area is page-locked memory); YGZF_PACKED_MAX (bytes) lowers it so
// that tests reach the large-transfer paths with ordinary frames
static const size_t kPackedMax = getenv("YGZF_PACKED_MAX") ? (size_t) atoll(getenv("YGZF_PACKED_MAX")) : (size_t) (4u << 20);
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
            if (inBytes) HIPCHECK(c, hipMemcpyAsync(c->dPack.p, c->hStage, inBytes, hipMemcpyHostToDevice, c->stream));
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
        if (outBytes) HIPCHECK(c, hipMemcpyAsync(c->hStage + inBytes, (uint8_t *) c->dPack.p + inBytes, outBytes, hipMemcpyDeviceToHost, c->stream));
        HIPCHECK(c, hipStreamSynchronize(c->stream));
        for (const Seg &s : out)
            if (s.bytes && s.dst) memcpy(s.dst, c->hStage + inBytes + s.off, s.bytes);
        return YGZF_OK;
    }
};

// Per-(w,h) geometry: level sizes, resize coefficient tables (cv::resize INTER_LINEAR fixed point, restated from the
// OpenCV 2.4/3.2 algorithm: fx = (float)((dx+0.5)*scale - 0.5), 11-bit coefficients), FAST cell grid (:733-745),
// octree root layout (:537-557) and buffer offsets.
// Row ranges of k_pyr_strips (extract_kernels.hip): the last level is cut into S strips; walking up, a strip needs of level l the source
// rows of the rows it produces of level l + 1 (yofs is monotone: first row's sy .. last row's sy + 1, clamped as the kernel clamps) and
// owns the S-th part of every level.  S grows until the two LDS regions (even / odd levels) fit.
static void plan_pyr_strips(Geometry &G, int L) {
    G.pyrPlan.clear();
    if (L < 3) return;
    if (G.lv[0].h >= 65536) return;   // the strip plan packs row ranges into 16-bit fields: taller images keep one launch per level
    for (int l = 1; l < L; l++)
        if (G.lv[l].area2x || !G.lv[l].tiledOk || (G.lv[l].w + 3) / 4 > kPyrStripMaxThreads) return;
    // 752x480, one frame: 16 strips 148 us per resident extraction, 24 146, 32 144, 48 143 (the halo rows grow with the strip count: 1.23 x
    // the pixels of the pyramid at 16); eight frames: 16 strips 218 us, 32 215
    // (more strips when the LDS regions do not fit, fewer for images whose last level has fewer than 64 rows).
    const int candidates[] = {32, 48, 64, 16, 8};
    const int minS = getenv("YGZF_PYR_STRIPS") ? atoi(getenv("YGZF_PYR_STRIPS")) : 0;   // A/B runs: at least this many strips
    for (int S : candidates) {
        if (S < minS) continue;
        if (G.lv[L - 1].h < 2 * S) continue;
        std::vector<PyrStripPlan> plan(S);
        size_t bytes[2] = {0, 0};
        int maxRows = 0;
        for (int s = 0; s < S; s++) {
            int rows = 0, pca = 0, pcb = 0;   // the range produced of the level below the one in hand
            PyrStripPlan &p = plan[s];
            std::memset(&p, 0, sizeof p);
            for (int l = L - 1; l >= 0; l--) {
                const int hl = G.lv[l].h;
                int wa = (int) ((long long) hl * s / S), wb = (int) ((long long) hl * (s + 1) / S), ca = wa, cb = wb;
                if (l < L - 1) {
                    const LevelGeom &n = G.lv[l + 1];
                    const int na = std::min(std::max(G.yofs[n.ytab + pca], 0), hl - 1);
                    const int nb = std::min(std::max(G.yofs[n.ytab + pcb - 1] + 1, 0), hl - 1) + 1;
                    if (l == 0) { ca = na; cb = nb; wa = wb = 0; }
                    else { ca = std::min(wa, na); cb = std::max(wb, nb); }
                }
                p.lv[l] = make_uint2((unsigned) ca | ((unsigned) cb << 16), (unsigned) wa | ((unsigned) wb << 16));
                pca = ca; pcb = cb;
                bytes[l & 1] = std::max(bytes[l & 1], (size_t) (cb - ca) * pyr_strip_lds_pitch(G.lv[l].w));
                if (l >= 1) rows += cb - ca;
            }
            maxRows = std::max(maxRows, rows);
        }
        if (maxRows > kPyrStripMaxThreads) continue;   // one thread per row fills the row table
        size_t nCols = 0;
        for (int l = 1; l < L; l++) nCols += (size_t) G.lv[l].w;
        const size_t rowBytes = ((size_t) maxRows * 8 + 15) & ~(size_t) 15, head = rowBytes + ((nCols * 8 + 15) & ~(size_t) 15);
        const size_t a = (bytes[0] + 16 + 15) & ~(size_t) 15, b = (bytes[1] + 16 + 15) & ~(size_t) 15;   // hrow reads up to 11 bytes past a row's pixels
        if (head + a + b > 160 * 1024) continue;
        G.pyrPlan = plan;
        G.pyrStripOffCol = (int) rowBytes;
        G.pyrStripOffA = (int) head;
        G.pyrStripOffB = (int) (head + a);
        G.pyrStripLds = head + a + b;
        std::memset(G.pyrLevels, 0, sizeof G.pyrLevels);
        for (int l = 0; l < L; l++) {
            const LevelGeom &g = G.lv[l];
            G.pyrLevels[l] = PyrStripLevel{g.w, g.h, g.pitch, g.xtab, g.ytab, 0, (long long) g.off};
        }
        return;
    }
}

static int build_geometry(ygzf_ctx *c, int w, int h, Geometry &G) {
    const Tables &T = c->tab;
    const int L = T.cfg.nlevels;
    G = Geometry();
    G.w = w;
    G.h = h;
    G.lv.assign(L, LevelGeom());
    long long off = 0, slotBase = 0, candBase = 0;
    int cellBase = 0, kpBase = 0, groupBase = 0;
    for (int l = 0; l < L; l++) {
        LevelGeom &g = G.lv[l];
        memset(&g, 0, sizeof g);
        T.levelSize(w, h, l, &g.w, &g.h);
        if (g.w < 1 || g.h < 1) return fail(c, YGZF_ERR_UNSUPPORTED, "pyramid level %d of a %dx%d image is empty", l, w, h);
        if (g.w > 4096 + 2 * kBorder || g.h > 4096 + 2 * kBorder)
            return fail(c, YGZF_ERR_UNSUPPORTED, "level %d is %dx%d: larger than 4128 px is not supported", l, g.w, g.h);
        g.pitch = align_up(g.w, 64);
        g.scale = T.scale[l];
        g.kpSize = (float) (int) (kPatchSize * T.scale[l]);  // :789 int truncation of a float product
        g.nFeat = T.nFeat[l];
        if (l > 0) {
            g.off = off;
            off += (long long) g.pitch * g.h;
            const LevelGeom &s = G.lv[l - 1];
            g.area2x = (s.w == 2 * g.w && s.h == 2 * g.h);
            g.xtab = (int) G.xofs.size();
            g.ytab = (int) G.yofs.size();
            const double scale_x = 1. / ((double) g.w / s.w), scale_y = 1. / ((double) g.h / s.h);
            for (int dx = 0; dx < g.w; dx++) {
                float fx = (float) ((dx + 0.5) * scale_x - 0.5);
                int sx = (int) std::floor(fx);
                fx -= sx;
                if (sx < 0) { fx = 0; sx = 0; }
                if (sx >= s.w - 1) { fx = 0; sx = s.w - 1; }
                G.xofs.push_back(sx);
                G.xalpha.push_back(sat_short((1.f - fx) * 2048));
                G.xalpha.push_back(sat_short(fx * 2048));
            }
            for (int dy = 0; dy < g.h; dy++) {
                float fy = (float) ((dy + 0.5) * scale_y - 0.5);
                int sy = (int) std::floor(fy);
                fy -= sy;
                G.yofs.push_back(sy);
                G.ybeta.push_back(sat_short((1.f - fy) * 2048));
                G.ybeta.push_back(sat_short(fy * 2048));
            }
            // the LDS-tiled resize kernel stages <= 44 rows x 328 bytes of source per 256 x 32 output tile: true for the usual scale
            // factors (<= ~1.27); steeper pyramids take the per-pixel kernel
            g.tiledOk = 1;
            for (int x0 = 0; x0 < g.w && g.tiledOk; x0 += 256) {
                const int xl = std::min(x0 + 256, g.w) - 1;
                const int sxa = G.xofs[g.xtab + x0] & ~3, sxb = std::min(G.xofs[g.xtab + xl] + 1, s.w - 1);
                if ((sxb - sxa) / 4 + 1 > 328 / 4) g.tiledOk = 0;
            }
            for (int y0 = 0; y0 < g.h && g.tiledOk; y0 += 32) {
                const int yl = std::min(y0 + 32, g.h) - 1;
                const int sya = std::min(std::max(G.yofs[g.ytab + y0], 0), s.h - 1), syb = std::min(std::max(G.yofs[g.ytab + yl] + 1, 0), s.h - 1);
                if (syb - sya + 1 > 44) g.tiledOk = 0;
            }
        }
        // FAST cells
        g.maxBorderX = g.w - kEdgeThreshold + 3;
        g.maxBorderY = g.h - kEdgeThreshold + 3;
        const float width = (float) (g.maxBorderX - kBorder), height = (float) (g.maxBorderY - kBorder);
        g.regW = g.maxBorderX - kBorder;
        g.regH = g.maxBorderY - kBorder;
        int nCols = width > 0 ? (int) (width / 30.f) : 0, nRows = height > 0 ? (int) (height / 30.f) : 0;
        if (nCols < 1 || nRows < 1) nCols = nRows = 0;  // reference: division by zero (UB) -> defined as "no keypoints"
        g.nCols = nCols;
        g.nRows = nRows;
        g.cellBase = cellBase;
        g.groupBase = groupBase;
        g.slotBase = slotBase;
        g.candBase = candBase;
        g.kpBase = kpBase;
        if (nCols) {
            g.wCell = (int) std::ceil(width / nCols);
            g.hCell = (int) std::ceil(height / nRows);
            g.slotCap = ((g.wCell + 1) / 2) * ((g.hCell + 1) / 2);
            const int nc = nCols * nRows;
            cellBase += nc;
            groupBase += ((nCols + 1) / 2) * ((nRows + 1) / 2);
            G.fastSmapRows = std::max(G.fastSmapRows, g.hCell + 2);
            G.fastWinPitch = std::max(G.fastWinPitch, (g.wCell + 6 + 1 + 3) / 4 * 4);   // per-wave window of k_fast_quads: column 0 unused
            G.fastWinRows = std::max(G.fastWinRows, g.hCell + 6);
            G.fastQuadCap = std::max(G.fastQuadCap, ((g.wCell + 3) / 4 * g.hCell + 3) / 4 * 4);
            G.fastWCellMax = std::max(G.fastWCellMax, g.wCell);
            slotBase += (long long) nc * g.slotCap;
            g.candCap = nc * g.slotCap;
            candBase += g.candCap;
            G.maxCellsPerLevel = std::max(G.maxCellsPerLevel, nc);
            // octree roots
            int nIni = (int) std::round((float) g.regW / (float) g.regH);
            if (nIni < 1) nIni = 1;  // reference: UB for tall-narrow regions; defined as one root
            g.nIni = nIni;
            g.hX = (float) g.regW / nIni;
            int rootW = 1;
            for (int i = 0; i < nIni; i++)
                rootW = std::max(rootW, (int) (g.hX * (float) (i + 1)) - (int) (g.hX * (float) i));
            int D = 1;
            while ((1 << D) < std::max(std::max(rootW, g.regH), 2)) D++;
            g.depth = D + 1;
            int rootBits = 0;
            while ((1 << rootBits) < nIni) rootBits++;
            g.keyBits = rootBits + 2 * g.depth;
            if (g.keyBits > 32) return fail(c, YGZF_ERR_UNSUPPORTED, "octree path key needs %d bits at level %d", g.keyBits, l);
            g.kpCap = std::max(g.nFeat + 3, 4 * nIni) + 1;
            // k_octree's work records carry the list position in 16 bits (score | position << 8 | level << 24)
            if (g.kpCap > 65535) return fail(c, YGZF_ERR_UNSUPPORTED, "level %d would keep up to %d keypoints (at most 65535 per level)", l, g.kpCap);
        } else {
            g.nIni = 1;
            g.hX = 1.f;
            g.kpCap = 0;
        }
        kpBase += g.kpCap;
        G.kpCapMax = std::max(G.kpCapMax, g.kpCap);
    }
    G.pyrBytes = off;
    plan_pyr_strips(G, L);
    G.totalCells = cellBase;
    G.totalGroups = groupBase;
    // per-cell records of k_fast_tab: the cell loop of ComputeKeyPointsOctTree (src/ORBextractor.cc:747-764) evaluated once per geometry
    G.fastCells.assign((size_t) groupBase * 4, FastCellRec{0, 0, 0, 0, 0, 0, 0, 0});
    for (int l = 0; l < L; l++) {
        const LevelGeom &g = G.lv[l];
        if (!g.nCols) continue;
        const int gCols = (g.nCols + 1) / 2;
        for (int ci = 0; ci < g.nRows; ci++)
            for (int cj = 0; cj < g.nCols; cj++) {
                FastCellRec &r = G.fastCells[(size_t) (g.groupBase + (ci / 2) * gCols + cj / 2) * 4 + (ci & 1) * 2 + (cj & 1)];
                const int cidx = ci * g.nCols + cj;
                const int iniX = kBorder + cj * g.wCell, iniY = kBorder + ci * g.hCell;
                const int maxX = std::min(iniX + g.wCell + 6, g.maxBorderX), maxY = std::min(iniY + g.hCell + 6, g.maxBorderY);
                const bool skip = (iniX >= g.maxBorderX - 6) || (iniY >= g.maxBorderY - 3);   // :751, :759
                const int dw = maxX - iniX - 6, dh = maxY - iniY - 6;
                const bool run = !skip && dw > 0 && dh > 0;
                r.xy = (unsigned) (iniX - 1) | ((unsigned) iniY << 16);
                r.geo = (unsigned) (l ? g.pitch : 0) | ((unsigned) (run ? dw : 0) << 16) | ((unsigned) (run ? dh : 0) << 24);
                r.flags = (unsigned) (run ? maxY - iniY : 0) | ((kFastCellExists | (run ? kFastCellRun : 0u)) << 8) | ((unsigned) l << 16);
                r.cell = (unsigned) (g.cellBase + cidx);
                r.slot = (unsigned) (g.slotBase + (long long) cidx * g.slotCap);
                r.off = (unsigned) g.off;
            }
    }
    G.totalSlots = slotBase;
    G.candStride = candBase;
    G.kpStride = kpBase;
    if (G.totalCells > 0 && (size_t) G.candStride > 0xFFFFFEull)
        return fail(c, YGZF_ERR_UNSUPPORTED, "more than 2^24 candidate slots per frame");
    return YGZF_OK;
}

static int apply_geometry(ygzf_ctx *c, int w, int h, int nFrames) {
    if (w < 1 || h < 1) return fail(c, YGZF_ERR_INVALID, "image size %dx%d", w, h);
    if (w > c->maxW || h > c->maxH) return fail(c, YGZF_ERR_INVALID, "image %dx%d exceeds the context maximum %dx%d", w, h, c->maxW, c->maxH);
    if (nFrames < 1 || nFrames > c->maxBatch) return fail(c, YGZF_ERR_INVALID, "batch of %d frames (context maximum %d)", nFrames, c->maxBatch);
    const int L = c->tab.cfg.nlevels;
    if (c->geo.w != w || c->geo.h != h) {
        Geometry G;
        int rc = build_geometry(c, w, h, G);
        if (rc) return rc;
        // c->geo is committed only after every fallible step below has succeeded: a failed attempt leaves (w, h) unset, so that a retry
        // runs the whole setup again instead of continuing on partial device tables
        c->geo.w = c->geo.h = 0;
        c->pyrResident = false;
        c->pyrHeld = false;
        c->aheadPending = false;
        c->carryValid = false;
        c->lastFrames = 0;
        HIPCHECK(c, hipStreamSynchronize(c->stream));
        rc = ensure(c, c->dGeom, sizeof(LevelGeom) * L);
        if (rc) return rc;
        HIPCHECK(c, hipMemcpy(c->dGeom.p, G.lv.data(), sizeof(LevelGeom) * L, hipMemcpyHostToDevice));
        struct { ygzf_ctx::Buf *b; const void *src; size_t bytes; } up[4] = {
            {&c->dXofs, G.xofs.data(), G.xofs.size() * sizeof(int)},
            {&c->dXalpha, G.xalpha.data(), G.xalpha.size() * sizeof(short)},
            {&c->dYofs, G.yofs.data(), G.yofs.size() * sizeof(int)},
            {&c->dYbeta, G.ybeta.data(), G.ybeta.size() * sizeof(short)}};
        for (auto &u : up) {
            rc = ensure(c, *u.b, std::max<size_t>(u.bytes, 16));
            if (rc) return rc;
            if (u.bytes) HIPCHECK(c, hipMemcpy(u.b->p, u.src, u.bytes, hipMemcpyHostToDevice));
        }
        if (!G.pyrPlan.empty()) {
            rc = ensure(c, c->dPyrPlan, sizeof G.pyrLevels + G.pyrPlan.size() * sizeof(PyrStripPlan));   // level table, then one plan per strip
            if (rc) return rc;
            HIPCHECK(c, hipMemcpy(c->dPyrPlan.p, G.pyrLevels, sizeof G.pyrLevels, hipMemcpyHostToDevice));
            HIPCHECK(c, hipMemcpy((char *) c->dPyrPlan.p + sizeof G.pyrLevels, G.pyrPlan.data(), G.pyrPlan.size() * sizeof(PyrStripPlan), hipMemcpyHostToDevice));
            if (pyr_strips_prepare(G.pyrStripLds) != hipSuccess) {   // no such LDS allotment on this device: one launch per level
                (void) hipGetLastError();
                G.pyrPlan.clear();
            }
        }
        rc = ensure(c, c->dFastCells, std::max<size_t>(G.fastCells.size() * sizeof(FastCellRec), 32));
        if (rc) return rc;
        if (!G.fastCells.empty()) HIPCHECK(c, hipMemcpy(c->dFastCells.p, G.fastCells.data(), G.fastCells.size() * sizeof(FastCellRec), hipMemcpyHostToDevice));
        // LDS-resident candidate sort buffers: as many as keep two workgroups per CU (<= ~78 KB each)
        {
            // per-list-position arrays (19 x cap ints) stay in LDS while one workgroup fits the CU; very large per-level feature budgets
            // move them to a global arena
            c->octGlobalNodes = octree_lds_bytes(G.maxCellsPerLevel, G.kpCapMax, 0, false) > 142 * 1024;
            const size_t fixed = octree_lds_bytes(G.maxCellsPerLevel, G.kpCapMax, 0, c->octGlobalNodes);
            const size_t budget = 71 * 1024;   // two workgroups per CU beside 9 KB of static LDS each (radix histogram)
            c->octLdsCand = fixed + 16 * 256 < budget ? (int) ((budget - fixed) / 16) : 0;
            if (c->octLdsCand > 8192) c->octLdsCand = 8192;
        }
        c->octLds = octree_lds_bytes(G.maxCellsPerLevel, G.kpCapMax, c->octLdsCand, c->octGlobalNodes);
        // Levels too large for LDS-resident candidate buffers (1920x1080 / 4000 and up) take the histogram plan: no sort, tree passes in LDS
        // (extract_kernels.hip, k_octree<.., kHist>).  Consecutive levels are launched together while their list caps stay within a factor of two of
        // the group's first level, so that the small levels do not reserve the LDS of the large ones.
        c->octGroups.clear();
        // (YGZF_OCT_PLAN=hist / sort forces one plan, YGZF_OCT_HIST_BINS bounds the histogram: the tests run every geometry through both plans
        // and through the fall-back from one to the other)
        const char *planEnv = getenv("YGZF_OCT_PLAN");
        const bool wantHist = planEnv && !strcmp(planEnv, "hist") ? true : planEnv && !strcmp(planEnv, "sort") ? false : (c->octGlobalNodes || c->octLdsCand == 0);
        const int binsEnv = getenv("YGZF_OCT_HIST_BINS") ? atoi(getenv("YGZF_OCT_HIST_BINS")) : 0;
        if (wantHist) {
            bool ok = true;
            for (int l = 0; l < L && ok;) {
                ygzf_ctx::OctGroup grp;
                grp.l0 = l;
                int cells = 0, tabs = 0;
                for (; l < L && (grp.n == 0 || 2 * G.lv[l].kpCap > G.lv[grp.l0].kpCap); l++, grp.n++) {
                    const LevelGeom &g = G.lv[l];
                    const int nc = g.nCols * g.nRows;
                    grp.cap = std::max(grp.cap, g.kpCap);
                    cells = std::max(cells, nc + 1);
                    if (nc > 0)
                        tabs = std::max(tabs, nc + 1 + std::max(g.regW, g.nCols * g.wCell) + 9 + std::max(g.regH, g.nRows * g.hCell) + 9 + nc);
                }
                if (grp.cap < 1) grp.cap = 1;
                const size_t budget = 150 * 1024;
                grp.histBins = binsEnv >= 4 && binsEnv <= 8192 ? binsEnv : 8192;
                grp.regionInts = std::max(std::max(19 * grp.cap, cells), tabs);
                if (octree_hist_lds_bytes(grp.regionInts, grp.histBins) > budget) grp.regionInts = std::max(19 * grp.cap, cells);   // keys without the tables
                while (grp.histBins > 1024 && !binsEnv && octree_hist_lds_bytes(grp.regionInts, grp.histBins) > budget) grp.histBins /= 2;
                grp.lds = octree_hist_lds_bytes(grp.regionInts, grp.histBins);
                if (grp.lds > budget) ok = false;
                c->octGroups.push_back(grp);
            }
            if (!ok) c->octGroups.clear();
        }
        // A launch of ONE frame is eight workgroups whose duration is a level's: there the histogram plan (no sort: 11 of the sort plan's 40 us
        // per level; its tree passes 12 us against 20) wins even where the candidates would fit LDS -- if all levels go in ONE launch, sized for
        // the largest (with a handful of workgroups nobody else wants the LDS).  Launches of up to 128 workgroups take it (YGZF_OCT_SMALL_WGS; YGZF_OCT_PLAN=sort: never).
        c->haveOctSmall = false;
        if (!(planEnv && !strcmp(planEnv, "sort"))) {
            ygzf_ctx::OctGroup grp;
            grp.l0 = 0;
            grp.n = L;
            int cells = 0, tabs = 0;
            for (int l = 0; l < L; l++) {
                const LevelGeom &g = G.lv[l];
                const int nc = g.nCols * g.nRows;
                grp.cap = std::max(grp.cap, g.kpCap);
                cells = std::max(cells, nc + 1);
                if (nc > 0) tabs = std::max(tabs, nc + 1 + std::max(g.regW, g.nCols * g.wCell) + 9 + std::max(g.regH, g.nRows * g.hCell) + 9 + nc);
            }
            if (grp.cap < 1) grp.cap = 1;
            const size_t budget = 150 * 1024;
            // bins: sixteen per node the largest level may end with (a tree of N leaves rarely splits below the depth that offers 16 N cells; if it
            // does, the level restarts on the sorting path) -- the prefix sum over 8192 bins was 3.3 of a level's 29 us, over 2048 it is 1
            int want = 1024;
            while (want < 16 * grp.cap && want < 8192) want *= 2;
            grp.histBins = binsEnv >= 4 && binsEnv <= 8192 ? binsEnv : want;
            grp.regionInts = std::max(std::max(19 * grp.cap, cells), tabs);
            if (octree_hist_lds_bytes(grp.regionInts, grp.histBins) > budget) grp.regionInts = std::max(19 * grp.cap, cells);
            while (grp.histBins > 1024 && !binsEnv && octree_hist_lds_bytes(grp.regionInts, grp.histBins) > budget) grp.histBins /= 2;
            grp.lds = octree_hist_lds_bytes(grp.regionInts, grp.histBins);
            if (grp.lds <= budget) {
                c->octSmall = grp;
                c->haveOctSmall = true;
                HIPCHECK(c, octree_prepare(0, false, true));
            }
        }
        if (c->octGroups.empty()) {
            if (c->octLds > 150 * 1024)
                return fail(c, YGZF_ERR_UNSUPPORTED, "octree kernel needs %zu bytes of LDS (cells/level %d, list cap %d)", c->octLds,
                            G.maxCellsPerLevel, G.kpCapMax);
            HIPCHECK(c, octree_prepare(c->octLds, c->octGlobalNodes, false));
        } else {
            c->octGlobalNodes = false;
            HIPCHECK(c, octree_prepare(0, false, true));
        }
        c->geo = G;
    }
    const Geometry &G = c->geo;
    const size_t B = (size_t) nFrames;
    int rc = 0;
    if ((rc = ensure(c, c->dPyr, std::max<size_t>(B * G.pyrBytes, 16) + 256))) return rc;
    if ((rc = ensure(c, c->dCellCnt, std::max<size_t>(B * G.totalCells * sizeof(unsigned short), 16)))) return rc;
    if ((rc = ensure(c, c->dSlots, std::max<size_t>(B * G.totalSlots * sizeof(unsigned), 16)))) return rc;
    const size_t cb = std::max<size_t>(B * G.candStride * sizeof(unsigned), 16);
    if ((rc = ensure(c, c->dK0, cb)) || (rc = ensure(c, c->dV0, cb)) || (rc = ensure(c, c->dK1, cb)) || (rc = ensure(c, c->dV1, cb)) ||
        (rc = ensure(c, c->dXY, cb)))
        return rc;
    if (c->octGlobalNodes && (rc = ensure(c, c->dOctNodes, B * L * 19 * (size_t) G.kpCapMax * sizeof(int) + 64))) return rc;
    const size_t kp = std::max<size_t>(B * G.kpStride, 16);
    const size_t kp1 = std::max<size_t>((B + 1) * G.kpStride, 16);  // + carry slot
    void *oldCnt = c->dOutCnt.p, *oldKp = c->dOutKp.p;
    if ((rc = ensure(c, c->dLvlXY, kp * sizeof(unsigned))) || (rc = ensure(c, c->dLvlScore, kp)) ||
        (rc = ensure(c, c->dLvlCnt, B * L * sizeof(int))) || (rc = ensure(c, c->dLvlBase, B * kMaxLevels * sizeof(int))) || (rc = ensure(c, c->dProcOrder, kp * sizeof(uint2))) || (rc = ensure(c, c->dLvlCand, B * L * sizeof(int))) ||
        (rc = ensure(c, c->dOutKp, kp1 * sizeof(ygzf_kp))) || (rc = ensure(c, c->dOutDesc, kp1 * 32)) ||
        (rc = ensure(c, c->dOutCnt, (B + 1) * sizeof(int))))
        return rc;
    if (oldCnt != c->dOutCnt.p || oldKp != c->dOutKp.p) c->carryValid = false;
    return YGZF_OK;
}

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
// The filler stream (ygzf_set_stream_partition): fill_begin makes it wait for everything queued on the context's stream so far and returns it
// (the context's own stream when there is no partition); fill_end makes the context's stream wait for what was queued on it.  Every hop is
// closed before an entry point returns, so that synchronising c->stream still drains the whole context.
static hipStream_t fill_begin(ygzf_ctx *c) {
    if (!c->streamFill) return c->stream;
    hipEvent_t e = c->evHop[c->hopSeq++ & 7];
    (void) hipEventRecord(e, c->stream);
    (void) hipStreamWaitEvent(c->streamFill, e, 0);
    return c->streamFill;
}
static void fill_end(ygzf_ctx *c) {
    if (!c->streamFill) return;
    hipEvent_t e = c->evHop[c->hopSeq++ & 7];
    (void) hipEventRecord(e, c->streamFill);
    (void) hipStreamWaitEvent(c->stream, e, 0);
}
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

// The launch sequence of ORBextractor::operator()(image...) for a batch resident on the device.
// Pyramid levels 1 .. L-1 of the frames in fs.  For ONE frame the chain is seven (nlevels - 1) small dependent kernels whose launches take the
// host longer than the kernels run; it is captured once per (buffers, geometry) into a graph and replayed with one call.
static int pyramid_chain(ygzf_ctx *c, const FrameSet &fs, int nFrames) {
    const Geometry &G = c->geo;
    const int L = c->tab.cfg.nlevels;
    const LevelGeom *dGeom = (const LevelGeom *) c->dGeom.p;
    auto launch_all = [&](bool prof) {
        for (int l = 1; l < L; l++) {
            if (prof) {
                ProfScope ps(c, KK_PYR);
                launch_pyr_resize(c->stream, fs, dGeom, G.lv[l], l, nFrames, (const int *) c->dXofs.p, (const short *) c->dXalpha.p,
                                  (const int *) c->dYofs.p, (const short *) c->dYbeta.p);
            } else
                launch_pyr_resize(c->stream, fs, dGeom, G.lv[l], l, nFrames, (const int *) c->dXofs.p, (const short *) c->dXalpha.p,
                                  (const int *) c->dYofs.p, (const short *) c->dYbeta.p);
        }
    };
    if (!G.pyrPlan.empty() && nFrames <= c->pyrStripFrames) {   // a few frames: the whole chain in one launch
        ProfScope ps(c, KK_PYR);
        launch_pyr_strips(c->stream, fs, L, (const PyrStripPlan *) ((const char *) c->dPyrPlan.p + sizeof G.pyrLevels), (const PyrStripLevel *) c->dPyrPlan.p,
                          (int) G.pyrPlan.size(), G.pyrStripOffCol, G.pyrStripOffA, G.pyrStripOffB, G.pyrStripLds, nFrames,
                          (const int *) c->dXofs.p, (const short *) c->dXalpha.p, (const int *) c->dYofs.p, (const short *) c->dYbeta.p);
        return YGZF_OK;
    }
    if (nFrames != 1 || L < 3 || !c->useGraphs || c->profile || c->debugSync) {
        launch_all(true);
        return YGZF_OK;
    }
    const void *key[6] = {c->dGeom.p, c->dXofs.p, c->dXalpha.p, c->dYofs.p, c->dYbeta.p,
                          (const void *) (((uintptr_t) G.w << 32) | (uintptr_t) (unsigned) G.h)};
    if (!c->pyrGraph.exec || memcmp(key, c->pyrGraphKey, sizeof key) != 0) {
        pyr_chain_graph_destroy(&c->pyrGraph);
        const hipError_t e = pyr_chain_graph_build(&c->pyrGraph, fs, dGeom, G.lv.data(), L, (const int *) c->dXofs.p, (const short *) c->dXalpha.p,
                                                   (const int *) c->dYofs.p, (const short *) c->dYbeta.p);
        if (e != hipSuccess) {   // no graphs on this runtime: plain launches from now on
            (void) hipGetLastError();
            c->useGraphs = false;
            launch_all(true);
            return YGZF_OK;
        }
        memcpy(c->pyrGraphKey, key, sizeof key);
    } else if (memcmp(&c->pyrGraph.fs, &fs, sizeof fs) != 0) {
        HIPCHECK(c, pyr_chain_graph_retarget(&c->pyrGraph, fs));
    }
    HIPCHECK(c, hipGraphLaunch(c->pyrGraph.exec, c->stream));
    return YGZF_OK;
}

// marks "the pyramid of frame 0 is complete" on the context's stream: what another context waits for before it copies that pyramid
// (ygzf_image_cache_put_resident) -- not the end of the stream, where an extraction queued ahead may still be running
static int mark_pyramid_done(ygzf_ctx *c) {
    if (!c->evPyrDone) HIPCHECK(c, hipEventCreateWithFlags(&c->evPyrDone, hipEventDisableTiming));
    HIPCHECK(c, hipEventRecord(c->evPyrDone, c->stream));
    c->evPyrDoneValid = true;
    return YGZF_OK;
}

// pyramidReady: dPyr already holds the pyramid of the frame to extract (ygzf_compute_pyramid / ygzf_extract_resident) -- the previous
// extraction's pyramid is gone from it, so the pyramid carry of ygzf_align_batch_prev is valid only when the caller copied it out BEFORE
// overwriting dPyr (pyramidCarried; ygzf_compute_pyramid in extract-ahead mode does).
static int run_extract(ygzf_ctx *c, const FrameSet &fs, int nFrames, bool pyramidReady = false, bool pyramidCarried = false) {
    const Geometry &G = c->geo;
    c->pyrResident = false;
    c->aheadPending = false;
    const int L = c->tab.cfg.nlevels;
    const LevelGeom *dGeom = (const LevelGeom *) c->dGeom.p;
    // slot 0 of the output arrays carries the last frame of the previous batch (Last frame of pair 0 in ygzf_match_batch_prev)
    ygzf_kp *outKp = (ygzf_kp *) c->dOutKp.p;
    uint8_t *outDesc = (uint8_t *) c->dOutDesc.p;
    int *outCnt = (int *) c->dOutCnt.p;
    launch_carry_slot(c->stream, outKp, outDesc, outCnt, (c->carryValid && c->lastFrames > 0 && G.kpStride > 0) ? (long long) c->lastFrames : 0, G.kpStride);
    outKp += G.kpStride;
    outDesc += (size_t) G.kpStride * 32;
    outCnt += 1;
    if (c->alignCarry && G.pyrBytes > 0) {   // the previous batch's last pyramid is the reference of pair 0 in ygzf_align_batch_prev
        int rc2 = ensure(c, c->dCarryPyr, (size_t) G.pyrBytes + 256);
        if (rc2) return rc2;
        c->carryPyrValid = c->carryValid && c->lastFrames > 0 && (!pyramidReady || pyramidCarried);
        if (c->carryPyrValid && !pyramidReady)
            HIPCHECK(c, hipMemcpyAsync(c->dCarryPyr.p, (uint8_t *) c->dPyr.p + (size_t) (c->lastFrames - 1) * G.pyrBytes, (size_t) G.pyrBytes,
                                       hipMemcpyDeviceToDevice, c->stream));
    }
    if (!pyramidReady) {
        int rcP = pyramid_chain(c, fs, nFrames);
        if (rcP || (rcP = mark_pyramid_done(c))) return rcP;
    }
    if (G.totalCells > 0) {
        int groupBase[kMaxLevels];
        for (int l = 0; l < L; l++) groupBase[l] = G.lv[l].groupBase;
        // threshold plan of this launch: from the statistics of an earlier launch of this context (whatever the asynchronous copy has
        // delivered by now; a stale or half-written snapshot only affects speed -- both plans return the same keypoints)
        {
            unsigned cells = 0, usedMin = 0, extra = 0, plan = 0;
            const volatile unsigned *hs = c->hFastStats;   // written by the copy engine, possibly right now
            for (int k = 0; k < 64; k++) {
                cells += hs[4 * k]; usedMin += hs[4 * k + 1]; extra += hs[4 * k + 2]; plan |= hs[4 * k + 3];
            }
            if (cells >= 32 && (plan == 1 || plan == 2)) {
                if (plan == 1) c->fastExtraRounds = (double) extra / cells;   // only the one-pass plan sees every FAST(minTh) corner
                // a score round costs ~180 vector instructions per wave, a second pass 1 ~700: iniTh first pays when the rounds it saves
                // outweigh the second passes of the cells that are empty at iniTh
                c->fastIniFirst = c->fastExtraRounds * 180.0 > ((double) usedMin / cells) * 700.0;
            }
        }
        const bool iniFirst = c->fastPlan == 2 || (c->fastPlan == 0 && c->fastIniFirst);
        const bool collect = c->fastPlan == 0 && (c->fastLaunches < 4 || (c->fastLaunches & 7) == 0);   // content drifts slowly: sample it
        c->fastLaunches++;
        if (collect) {
            int rcS = ensure(c, c->dFastStats, kFastStatWords * sizeof(unsigned));
            if (rcS) return rcS;
            HIPCHECK(c, hipMemsetAsync(c->dFastStats.p, 0, kFastStatWords * sizeof(unsigned), c->stream));
        }
        {
            // which form of the cell loop: the table-driven one (per-cell records, LDS-DMA staging) wherever its 48-byte window pitch holds the
            // widest cell and the pyramid offsets / frame sizes fit its 32-bit / 16-bit record fields
            const bool tab = c->fastKernel != YGZF_FAST_KERNEL_REGISTER_STAGING && G.fastWCellMax <= kFastTabMaxCell && G.pyrBytes < (1ll << 32) &&
                             G.w < 65536 && G.h < 65536 && G.totalSlots < (1ll << 32);
            ProfScope ps(c, tab ? KK_FAST : KK_FASTQ);
            if (tab)
                launch_fast_tab(c->stream, fs, (const FastCellRec *) c->dFastCells.p, c->tab.cfg.ini_th_fast, c->tab.cfg.min_th_fast,
                                (unsigned short *) c->dCellCnt.p, (unsigned *) c->dSlots.p, G.totalCells, G.totalSlots, G.totalGroups, G.fastSmapRows,
                                nFrames, G.fastWinRows, G.fastQuadCap, iniFirst, collect ? (unsigned *) c->dFastStats.p : nullptr);
            else
                launch_fast_cells(c->stream, fs, dGeom, L, c->tab.cfg.ini_th_fast, c->tab.cfg.min_th_fast,
                                  (unsigned short *) c->dCellCnt.p, (unsigned *) c->dSlots.p, G.totalCells, G.totalSlots, G.totalGroups,
                                  G.fastSmapRows, nFrames, G.fastWinPitch, G.fastWinRows, G.fastQuadCap, groupBase, iniFirst, collect ? (unsigned *) c->dFastStats.p : nullptr);
        }
        if (collect) HIPCHECK(c, hipMemcpyAsync(c->hFastStats, c->dFastStats.p, kFastStatWords * sizeof(unsigned), hipMemcpyDeviceToHost, c->stream));
        long long *odbg = nullptr;
        if (c->octDebug) {
            int rc2 = ensure(c, c->dTmpA, kOctDbgWords * sizeof(long long));
            if (rc2) return rc2;
            odbg = (long long *) c->dTmpA.p;
            HIPCHECK(c, hipMemsetAsync(odbg, 0, kOctDbgWords * sizeof(long long), c->stream));
        }
        {
            hipStream_t so = fill_begin(c);
            {
            ProfScope ps(c, KK_OCTREE, so);
            const bool small = c->haveOctSmall && nFrames * L <= c->octSmallWgs;
            if (small) {
                const auto &grp = c->octSmall;
                launch_octree(so, dGeom, L, grp.l0, grp.n, (const unsigned short *) c->dCellCnt.p, (const unsigned *) c->dSlots.p,
                              G.totalCells, G.totalSlots, (unsigned *) c->dK0.p, (unsigned *) c->dV0.p, (unsigned *) c->dK1.p,
                              (unsigned *) c->dV1.p, (unsigned *) c->dXY.p, G.candStride, (unsigned *) c->dLvlXY.p,
                              (unsigned char *) c->dLvlScore.p, (int *) c->dLvlCnt.p, (int *) c->dLvlCand.p,
                              (uint2 *) c->dProcOrder.p, G.kpStride, grp.cap, 0, grp.lds, nFrames, odbg, nullptr, grp.regionInts, grp.histBins);
            } else if (c->octGroups.empty())
                launch_octree(so, dGeom, L, 0, L, (const unsigned short *) c->dCellCnt.p, (const unsigned *) c->dSlots.p,
                              G.totalCells, G.totalSlots, (unsigned *) c->dK0.p, (unsigned *) c->dV0.p, (unsigned *) c->dK1.p,
                              (unsigned *) c->dV1.p, (unsigned *) c->dXY.p, G.candStride, (unsigned *) c->dLvlXY.p,
                              (unsigned char *) c->dLvlScore.p, (int *) c->dLvlCnt.p, (int *) c->dLvlCand.p,
                              (uint2 *) c->dProcOrder.p, G.kpStride, G.kpCapMax, c->octLdsCand, c->octLds, nFrames, odbg,
                              c->octGlobalNodes ? (int *) c->dOctNodes.p : nullptr, 0, 0);
            else
                for (const auto &grp : c->octGroups)
                    launch_octree(so, dGeom, L, grp.l0, grp.n, (const unsigned short *) c->dCellCnt.p, (const unsigned *) c->dSlots.p,
                                  G.totalCells, G.totalSlots, (unsigned *) c->dK0.p, (unsigned *) c->dV0.p, (unsigned *) c->dK1.p,
                                  (unsigned *) c->dV1.p, (unsigned *) c->dXY.p, G.candStride, (unsigned *) c->dLvlXY.p,
                                  (unsigned char *) c->dLvlScore.p, (int *) c->dLvlCnt.p, (int *) c->dLvlCand.p,
                                  (uint2 *) c->dProcOrder.p, G.kpStride, grp.cap, 0, grp.lds, nFrames, odbg, nullptr, grp.regionInts, grp.histBins);
            }
            fill_end(c);
        }
        if (odbg) {
            long long st[kOctDbgWords];
            HIPCHECK(c, hipStreamSynchronize(c->stream));
            HIPCHECK(c, hipMemcpy(st, odbg, sizeof st, hipMemcpyDeviceToHost));
            for (int l = 0; l < L; l++)
                fprintf(stderr, "[ygzf octree lvl %d, 10ns ticks] prefix %lld keys %lld sort %lld bfs %lld final %lld  M=%lld n=%lld\n", l, st[l * 8 + 1] - st[l * 8],
                        st[l * 8 + 2] - st[l * 8 + 1], st[l * 8 + 3] - st[l * 8 + 2], st[l * 8 + 4] - st[l * 8 + 3], st[l * 8 + 5] - st[l * 8 + 4], st[l * 8 + 6], st[l * 8 + 7]);
            if (!c->octGroups.empty()) {
                fprintf(stderr, "[ygzf octree histogram plan: %zu launches;", c->octGroups.size());
                for (const auto &grp : c->octGroups) fprintf(stderr, " levels %d-%d cap %d bins %d lds %zu;", grp.l0, grp.l0 + grp.n - 1, grp.cap, grp.histBins, grp.lds);
                fprintf(stderr, " workgroups of %d frames that fell back to the sort, per level:", nFrames);
                for (int l = 0; l < L; l++) fprintf(stderr, " %lld", st[16 * 8 + l]);
                fprintf(stderr, "]\n");
            }
        }
        {
            ProfScope ps(c, KK_DESCRIBE);
            launch_describe(c->stream, fs, dGeom, L, (const int *) c->dLvlCnt.p, (int *) c->dLvlBase.p, (const uint2 *) c->dProcOrder.p, G.kpStride,
                            outKp, outDesc, outCnt, G.kpStride, nFrames, c->tab.cfg.cv_mode);
        }
    } else {
        HIPCHECK(c, hipMemsetAsync(outCnt, 0, sizeof(int) * nFrames, c->stream));
        HIPCHECK(c, hipMemsetAsync(c->dLvlCnt.p, 0, sizeof(int) * nFrames * L, c->stream));
        HIPCHECK(c, hipMemsetAsync(c->dLvlCand.p, 0, sizeof(int) * nFrames * L, c->stream));
    }
    HIPCHECK(c, hipGetLastError());
    c->pyrHeld = fs.img0 == (const uint8_t *) c->dImg0.p && fs.pyr == (uint8_t *) c->dPyr.p;   // frame 0 of the context's own buffers
    c->pyrHeldW = G.w;
    c->pyrHeldH = G.h;
    c->lastFrames = nFrames;
    c->lastFs = fs;
    c->carryValid = true;
    c->lastMatchPairs = 0;
    c->lastAlignPairs = 0;
    c->lastStereoPairs = 0;
    return YGZF_OK;
}

// `rows` rows of `w` bytes from host memory (row pitch srcPitch) into a pitched device image.  The copy engine executes a pitched copy whose
// width or pitches are not multiples of 4 row by row (measured: 2.6-3.3 ms for one 1241x376 or 641x479 frame); such images travel as ONE
// linear copy into a landing buffer and are re-pitched by a kernel.
static int upload_rows(ygzf_ctx *c, void *dst, size_t dstPitch, const uint8_t *src, size_t srcPitch, int w, size_t rows) {
    if (rows == 0 || w <= 0) return YGZF_OK;
    if ((w & 3) == 0 && (srcPitch & 3) == 0 && (dstPitch & 3) == 0 && ((uintptr_t) src & 3) == 0) {
        HIPCHECK(c, hipMemcpy2DAsync(dst, dstPitch, src, srcPitch, (size_t) w, rows, hipMemcpyHostToDevice, c->stream));
        return YGZF_OK;
    }
    const size_t bytes = (rows - 1) * srcPitch + (size_t) w;
    int rc = ensure(c, c->dUpStage, bytes + 64);
    if (rc) return rc;
    HIPCHECK(c, hipMemcpyAsync(c->dUpStage.p, src, bytes, hipMemcpyHostToDevice, c->stream));
    launch_repitch_rows(c->stream, (const uint8_t *) c->dUpStage.p, srcPitch, (uint8_t *) dst, dstPitch, w, rows);
    HIPCHECK(c, hipGetLastError());
    return YGZF_OK;
}

static int upload_frames(ygzf_ctx *c, const uint8_t *imgs, int nFrames, int w, int h, int row_pitch, size_t frame_stride,
                         FrameSet *fs) {
    c->pyrResident = false;
    c->aheadPending = false;
    c->pyrHeld = false;
    const int pitch = align_up(w, 64);
    int rc = ensure(c, c->dImg0, (size_t) nFrames * pitch * h);
    if (rc) return rc;
    if (nFrames == 1 && !getenv("YGZF_NO_STAGED_UPLOAD")) {
        hipPointerAttribute_t at;
        memset(&at, 0, sizeof at);
        bool pageable = true;
        if (hipPointerGetAttributes(&at, imgs) == hipSuccess) pageable = at.type != hipMemoryTypeHost && at.type != hipMemoryTypeDevice && at.type != hipMemoryTypeManaged;
        else (void) hipGetLastError();
        if (pageable) {
            const size_t bytes = (size_t) w * h;
            if (bytes > c->hInBytes) {
                if (c->evInPending) { HIPCHECK(c, hipEventSynchronize(c->evIn)); c->evInPending = false; }
                if (c->hIn) HIPCHECK(c, hipHostFree(c->hIn));
                c->hIn = nullptr;
                c->hInBytes = 0;
                HIPCHECK(c, hipHostMalloc((void **) &c->hIn, bytes + 64, hipHostMallocMapped));
                HIPCHECK(c, hipHostGetDevicePointer(&c->hInDev, c->hIn, 0));
                c->hInBytes = bytes;
            }
            if (!c->evIn) HIPCHECK(c, hipEventCreateWithFlags(&c->evIn, hipEventDisableTiming));
            if (c->evInPending) { HIPCHECK(c, hipEventSynchronize(c->evIn)); c->evInPending = false; }   // the previous upload's kernel has read the buffer
            if (row_pitch == w) memcpy(c->hIn, imgs, bytes);
            else for (int y = 0; y < h; y++) memcpy(c->hIn + (size_t) y * w, imgs + (size_t) y * row_pitch, (size_t) w);
            HostFrameList L;
            L.addr[0] = (unsigned long long) (uintptr_t) c->hInDev;
            launch_gather_host_frames(c->stream, L, 1, (size_t) w, (uint8_t *) c->dImg0.p, (size_t) pitch, (size_t) pitch * h, w, h);
            HIPCHECK(c, hipGetLastError());
            HIPCHECK(c, hipEventRecord(c->evIn, c->stream));
            c->evInPending = true;
            fs->img0 = (const uint8_t *) c->dImg0.p;
            fs->img0_stride = (long long) pitch * h;
            fs->img0_pitch = pitch;
            fs->pyr = (uint8_t *) c->dPyr.p;
            fs->pyr_stride = c->geo.pyrBytes;
            return YGZF_OK;
        }
    }
    // Host frames that already carry the device's row pitch (ygzf_host_row_pitch): the copy engine pays per ROW of a pitched copy (752-byte rows:
    // 48 GB/s where 1920-byte rows reach 55), so the rows handed to it are runs of k image rows -- (k - 1) x pitch + w bytes, k = the whole
    // frame by default -- and the padding bytes between image rows travel with them.
    static const int upK = getenv("YGZF_UPLOAD_K") ? atoi(getenv("YGZF_UPLOAD_K")) : 0;   // 0: whole frames; 1: image rows (as before); k: runs of k rows (k divides h) -- A/B runs
    if (row_pitch == pitch && pitch != w && upK != 1 && (frame_stride == 0 || frame_stride >= (size_t) pitch * h) && ((uintptr_t) imgs & 3) == 0 && (w & 3) == 0 &&
        (frame_stride & 3) == 0) {
        const size_t fsd = (size_t) pitch * h, fss = nFrames == 1 ? fsd : frame_stride;
        const int k = (upK > 1 && h % upK == 0) ? upK : h;
        if (k == h) {
            HIPCHECK(c, hipMemcpy2DAsync(c->dImg0.p, fsd, imgs, fss, (size_t) (h - 1) * pitch + w, (size_t) nFrames, hipMemcpyHostToDevice, c->stream));
        } else if (fss == fsd) {
            HIPCHECK(c, hipMemcpy2DAsync(c->dImg0.p, (size_t) k * pitch, imgs, (size_t) k * pitch, (size_t) (k - 1) * pitch + w, (size_t) nFrames * (h / k), hipMemcpyHostToDevice, c->stream));
        } else {
            for (int f = 0; f < nFrames; f++)
                HIPCHECK(c, hipMemcpy2DAsync((uint8_t *) c->dImg0.p + f * fsd, (size_t) k * pitch, imgs + f * fss, (size_t) k * pitch, (size_t) (k - 1) * pitch + w, (size_t) (h / k), hipMemcpyHostToDevice, c->stream));
        }
    } else if (nFrames == 1 || frame_stride == (size_t) row_pitch * h) {   // frames back to back: one copy of nFrames * h rows
        if ((rc = upload_rows(c, c->dImg0.p, (size_t) pitch, imgs, (size_t) row_pitch, w, (size_t) nFrames * h))) return rc;
    } else {
        for (int f = 0; f < nFrames; f++) {
            // (the landing buffer is reused: stream order keeps frame f's re-pitch ahead of frame f + 1's copy)
            if ((rc = upload_rows(c, (uint8_t *) c->dImg0.p + (size_t) f * pitch * h, (size_t) pitch, imgs + f * frame_stride, (size_t) row_pitch, w, (size_t) h)))
                return rc;
        }
    }
    fs->img0 = (const uint8_t *) c->dImg0.p;
    fs->img0_stride = (long long) pitch * h;
    fs->img0_pitch = pitch;
    fs->pyr = (uint8_t *) c->dPyr.p;
    fs->pyr_stride = c->geo.pyrBytes;
    return YGZF_OK;
}

extern "C" {

int ygzf_create(int device, const ygzf_extractor_cfg *cfg, int max_width, int max_height, int max_batch, ygzf_ctx **out) {
    if (!cfg || !out) return fail(nullptr, YGZF_ERR_INVALID, "null argument");
    *out = nullptr;
    if (cfg->nlevels < 1 || cfg->nlevels > kMaxLevels || cfg->nfeatures < 0 || !(cfg->scale_factor > 1.0f) || cfg->ini_th_fast < 0 ||
        cfg->min_th_fast < 0 || cfg->ini_th_fast > 255 || cfg->min_th_fast > 255 || cfg->min_th_fast > cfg->ini_th_fast)
        return fail(nullptr, YGZF_ERR_INVALID, "bad extractor configuration (nlevels 1..%d, scale_factor > 1, 0 <= minTh <= iniTh <= 255)",
                    kMaxLevels);
    if (cfg->cv_mode < YGZF_CV_LEGACY_SSE2 || cfg->cv_mode > YGZF_CV_4) return fail(nullptr, YGZF_ERR_INVALID, "cv_mode %d (0 legacy SSE2, 1 legacy integer, 2 OpenCV >= 3.4.11 / 4.x)", cfg->cv_mode);
    if (max_width < 1 || max_height < 1 || max_batch < 1) return fail(nullptr, YGZF_ERR_INVALID, "bad maximum sizes");
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev < 1)
        return fail(nullptr, YGZF_ERR_NO_DEVICE, "no HIP device available (%s): libygzf has no CPU fallback",
                    e == hipSuccess ? "device count 0" : hipGetErrorString(e));
    if (device < 0 || device >= ndev) return fail(nullptr, YGZF_ERR_NO_DEVICE, "device %d out of range (0..%d)", device, ndev - 1);
    ygzf_ctx *c = new ygzf_ctx();
    c->device = device;
    c->maxW = max_width;
    c->maxH = max_height;
    c->maxBatch = max_batch;
    c->tab.init(*cfg);
    auto bail = [&](int rc) {
        g_create_err = c->err;   // thread-local
        ygzf_destroy(c);
        return rc;
    };
#define CK(expr)                                                                                                   \
    do {                                                                                                           \
        hipError_t _e = (expr);                                                                                    \
        if (_e != hipSuccess) {                                                                                    \
            fail(c, YGZF_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(_e));                                  \
            return bail(YGZF_ERR_HIP);                                                                             \
        }                                                                                                          \
    } while (0)
    CK(hipSetDevice(device));
    CK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    CK(hipEventCreate(&c->tStart));
    CK(hipEventCreate(&c->tStop));
    CK(upload_constants(c->tab.umax));
    CK(hipHostMalloc((void **) &c->hFastStats, kFastStatWords * sizeof(unsigned)));
    memset(c->hFastStats, 0, kFastStatWords * sizeof(unsigned));
    if (const char *e = getenv("YGZF_FAST_KERNEL")) c->fastKernel = atoi(e) == 1 ? 1 : atoi(e) == 2 ? 2 : 0;
#undef CK
    // validate the largest configuration up front (and size the buffers once)
    int rc = apply_geometry(c, max_width, max_height, max_batch);
    if (rc) return bail(rc);
    if (const char *e = getenv("YGZF_FILL_CUS")) {   // A/B runs: the stream partition of every context of the process (ygzf_set_stream_partition)
        const int mm = getenv("YGZF_MAIN_MODE") ? atoi(getenv("YGZF_MAIN_MODE")) : 0;
        if ((atoi(e) != 0 || mm != 0) && (rc = ygzf_set_stream_partition(c, atoi(e), mm))) return bail(rc);
    }
    *out = c;
    return YGZF_OK;
}

void ygzf_destroy(ygzf_ctx *c) {
    if (!c) return;
    (void) hipSetDevice(c->device);
    if (c->stream) (void) hipStreamSynchronize(c->stream);
    ygzf_ctx::Buf *bufs[] = {&c->dGeom, &c->dXofs, &c->dXalpha, &c->dYofs, &c->dYbeta, &c->dImg0, &c->dPyr, &c->dCellCnt, &c->dSlots,
                             &c->dK0, &c->dV0, &c->dK1, &c->dV1, &c->dXY, &c->dLvlXY, &c->dLvlScore, &c->dLvlCnt, &c->dLvlBase, &c->dLvlCand,
                             &c->dOutKp, &c->dOutDesc, &c->dOutCnt, &c->dTmpA, &c->dTmpB, &c->dTmpC, &c->dWorld, &c->dOwner, &c->dMatch,
                             &c->dNMatch, &c->dPoses, &c->dQp, &c->dProcOrder, &c->dSpill, &c->dCarryPyr, &c->dOctNodes};
    for (auto *b : bufs)
        if (b->p) (void) hipFree(b->p);
    for (auto &b : c->dGen)
        if (b.p) (void) hipFree(b.p);
    for (auto &b : c->dSia)
        if (b.p) (void) hipFree(b.p);
    for (auto &b : c->dVoc)
        if (b.p) (void) hipFree(b.p);
    for (auto &b : c->dBow)
        if (b.p) (void) hipFree(b.p);
    for (auto &b : c->dF10)
        if (b.p) (void) hipFree(b.p);
    for (auto &b : c->dAl)
        if (b.p) (void) hipFree(b.p);
    for (auto &b : c->dDso)
        if (b.p) (void) hipFree(b.p);
    for (auto &b : c->dSt)
        if (b.p) (void) hipFree(b.p);
    if (c->dStBins.p) (void) hipFree(c->dStBins.p);
    for (auto &b : c->dDir)
        if (b.p) (void) hipFree(b.p);
    for (auto &b : c->dFr)
        if (b.p) (void) hipFree(b.p);
    if (c->dCacheImg.p) (void) hipFree(c->dCacheImg.p);
    if (c->dCachePyr.p) (void) hipFree(c->dCachePyr.p);
    if (c->dFastStats.p) (void) hipFree(c->dFastStats.p);
    if (c->dFastCells.p) (void) hipFree(c->dFastCells.p);
    if (c->dPack.p) (void) hipFree(c->dPack.p);
    if (c->dMatchStat.p) (void) hipFree(c->dMatchStat.p);
    if (c->dSplitCnt.p) (void) hipFree(c->dSplitCnt.p);
    if (c->dSplitX.p) (void) hipFree(c->dSplitX.p);
    if (c->dUpStage.p) (void) hipFree(c->dUpStage.p);
    if (c->hFastStats) (void) hipHostFree(c->hFastStats);
    if (c->hStage) (void) hipHostFree(c->hStage);
    if (c->hIn) (void) hipHostFree(c->hIn);
    if (c->evIn) (void) hipEventDestroy(c->evIn);
    for (auto &r : c->recs) { (void) hipEventDestroy(r.a); (void) hipEventDestroy(r.b); }
    for (auto e : c->pool) (void) hipEventDestroy(e);
    pyr_chain_graph_destroy(&c->pyrGraph);
    if (c->evShare) (void) hipEventDestroy(c->evShare);
    if (c->evPyrDone) (void) hipEventDestroy(c->evPyrDone);
    if (c->evPyramid) (void) hipEventDestroy(c->evPyramid);
    if (c->streamCopy) (void) hipStreamDestroy(c->streamCopy);
    if (c->streamFill) (void) hipStreamDestroy(c->streamFill);
    for (auto e : c->evHop)
        if (e) (void) hipEventDestroy(e);
    if (c->tStart) (void) hipEventDestroy(c->tStart);
    if (c->tStop) (void) hipEventDestroy(c->tStop);
    if (c->stream) (void) hipStreamDestroy(c->stream);
    delete c;
}

const char *ygzf_last_error(const ygzf_ctx *c) {
    if (c) return c->err.c_str();
    return g_create_err.c_str();
}

int ygzf_device_mem_info(ygzf_ctx *c, size_t *free_bytes, size_t *total_bytes) {
    if (!c || !free_bytes || !total_bytes) return fail(c, YGZF_ERR_INVALID, "null argument");
    HIPCHECK(c, hipSetDevice(c->device));
    HIPCHECK(c, hipStreamSynchronize(c->stream));
    HIPCHECK(c, hipMemGetInfo(free_bytes, total_bytes));
    return YGZF_OK;
}

int ygzf_set_fast_plan(ygzf_ctx *c, int plan) {
    if (!c) return YGZF_ERR_INVALID;
    if (plan < YGZF_FAST_PLAN_AUTO || plan > YGZF_FAST_PLAN_INI_FIRST) return fail(c, YGZF_ERR_INVALID, "FAST plan %d (0 auto, 1 one pass, 2 iniTh first)", plan);
    c->fastPlan = plan;
    return YGZF_OK;
}

int ygzf_set_fast_kernel(ygzf_ctx *c, int kernel) {
    if (!c) return YGZF_ERR_INVALID;
    if (kernel < YGZF_FAST_KERNEL_AUTO || kernel > YGZF_FAST_KERNEL_CELL_TABLE) return fail(c, YGZF_ERR_INVALID, "FAST kernel %d (0 auto, 1 register staging, 2 cell table + LDS-DMA)", kernel);
    c->fastKernel = kernel;
    return YGZF_OK;
}

int ygzf_set_extract_ahead(ygzf_ctx *c, int on) {
    if (!c) return YGZF_ERR_INVALID;
    HIPCHECK(c, hipSetDevice(c->device));
    if (on && !c->streamCopy) {
        HIPCHECK(c, hipStreamCreateWithFlags(&c->streamCopy, hipStreamNonBlocking));
        HIPCHECK(c, hipEventCreateWithFlags(&c->evPyramid, hipEventDisableTiming));
    }
    c->extractAhead = on != 0;
    if (!on) c->aheadPending = false;
    return YGZF_OK;
}

int ygzf_set_stream_partition(ygzf_ctx *c, int fill_cus, int main_mode) {
    if (!c) return YGZF_ERR_INVALID;
    if (main_mode < 0 || main_mode > 1) return fail(c, YGZF_ERR_INVALID, "main_mode %d (0 all compute units, 1 the complement of the filler's share)", main_mode);
    HIPCHECK(c, hipSetDevice(c->device));
    hipDeviceProp_t prop;
    HIPCHECK(c, hipGetDeviceProperties(&prop, c->device));
    const int nCU = prop.multiProcessorCount;
    if (fill_cus < -1 || fill_cus >= nCU) return fail(c, YGZF_ERR_INVALID, "fill_cus %d (-1 unmasked second stream, 0 off, 1..%d compute units)", fill_cus, nCU - 1);
    if (main_mode == 1 && fill_cus <= 0) return fail(c, YGZF_ERR_INVALID, "main_mode 1 needs a filler share");
    HIPCHECK(c, hipStreamSynchronize(c->stream));
    if (c->streamFill) {
        HIPCHECK(c, hipStreamSynchronize(c->streamFill));
        HIPCHECK(c, hipStreamDestroy(c->streamFill));
        c->streamFill = nullptr;
    }
    // mask bit i is compute unit i / 8 of XCD i % 8 (the driver deals the bits round-robin over the XCDs and, inside one, over its shader
    // engines): the first n bits are n / 8 compute units of every XCD
    const int words = (nCU + 31) / 32;
    std::vector<uint32_t> fill((size_t) words, 0u), rest((size_t) words, 0u);
    for (int i = 0; i < nCU; i++) (i < fill_cus ? fill : rest)[(size_t) i / 32] |= 1u << (i % 32);
    if (c->mainMode != main_mode || (main_mode == 1 && c->fillCUs != fill_cus)) {   // the context's own stream changes its share
        hipStream_t ns = nullptr;
        if (main_mode == 1) HIPCHECK(c, hipExtStreamCreateWithCUMask(&ns, (uint32_t) words, rest.data()));
        else HIPCHECK(c, hipStreamCreateWithFlags(&ns, hipStreamNonBlocking));
        HIPCHECK(c, hipStreamDestroy(c->stream));
        c->stream = ns;
    }
    if (fill_cus > 0) HIPCHECK(c, hipExtStreamCreateWithCUMask(&c->streamFill, (uint32_t) words, fill.data()));
    else if (fill_cus < 0) HIPCHECK(c, hipStreamCreateWithFlags(&c->streamFill, hipStreamNonBlocking));
    if (c->streamFill)
        for (auto &e : c->evHop)
            if (!e) HIPCHECK(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    c->fillCUs = fill_cus;
    c->mainMode = main_mode;
    return YGZF_OK;
}

int ygzf_get_fast_plan(const ygzf_ctx *c, int *plan) {
    if (!c || !plan) return YGZF_ERR_INVALID;
    *plan = c->fastPlan == YGZF_FAST_PLAN_AUTO ? (c->fastIniFirst ? YGZF_FAST_PLAN_INI_FIRST : YGZF_FAST_PLAN_ONE_PASS) : c->fastPlan;
    return YGZF_OK;
}

int ygzf_scale_tables_host(const ygzf_extractor_cfg *cfg, float *scale, float *inv_scale, float *sigma2, float *inv_sigma2, int *nfeat) {
    if (!cfg || cfg->nlevels < 1 || cfg->nlevels > kMaxLevels || cfg->nfeatures < 0 || !(cfg->scale_factor > 1.0f)) return YGZF_ERR_INVALID;
    Tables T;
    T.init(*cfg);
    const size_t n = sizeof(float) * cfg->nlevels;
    if (scale) memcpy(scale, T.scale.data(), n);
    if (inv_scale) memcpy(inv_scale, T.invScale.data(), n);
    if (sigma2) memcpy(sigma2, T.sigma2.data(), n);
    if (inv_sigma2) memcpy(inv_sigma2, T.invSigma2.data(), n);
    if (nfeat) memcpy(nfeat, T.nFeat.data(), sizeof(int) * cfg->nlevels);
    return YGZF_OK;
}

int ygzf_pyramid_plan_host(const ygzf_extractor_cfg *cfg, int w, int h, int *n_strips, int *lds_bytes, int *level_wh, unsigned short *rows, int rows_cap) {
    if (!cfg || cfg->nlevels < 1 || cfg->nlevels > kMaxLevels || cfg->nfeatures < 0 || !(cfg->scale_factor > 1.0f) || w < 1 || h < 1 || !n_strips)
        return YGZF_ERR_INVALID;
    ygzf_ctx tmp;   // never touches a device: carries the tables and receives the error text
    tmp.tab.init(*cfg);
    Geometry G;
    const int rc = build_geometry(&tmp, w, h, G);
    if (rc) return rc;
    const int L = cfg->nlevels, S = (int) G.pyrPlan.size();
    *n_strips = S;
    if (lds_bytes) *lds_bytes = (int) G.pyrStripLds;
    if (level_wh)
        for (int l = 0; l < L; l++) { level_wh[2 * l] = G.lv[l].w; level_wh[2 * l + 1] = G.lv[l].h; }
    if (rows) {
        if (rows_cap < S * L * 4) return YGZF_ERR_INVALID;
        for (int s = 0; s < S; s++)
            for (int l = 0; l < L; l++) {
                const uint2 v = G.pyrPlan[s].lv[l];
                unsigned short *o = rows + ((size_t) s * L + l) * 4;
                o[0] = (unsigned short) (v.x & 0xFFFFu); o[1] = (unsigned short) (v.x >> 16);
                o[2] = (unsigned short) (v.y & 0xFFFFu); o[3] = (unsigned short) (v.y >> 16);
            }
    }
    return YGZF_OK;
}

int ygzf_get_levels(const ygzf_ctx *c) { return c ? c->tab.cfg.nlevels : YGZF_ERR_INVALID; }

int ygzf_get_scale_tables(const ygzf_ctx *c, float *scale, float *inv_scale, float *sigma2, float *inv_sigma2) {
    if (!c) return YGZF_ERR_INVALID;
    const size_t n = sizeof(float) * c->tab.cfg.nlevels;
    if (scale) memcpy(scale, c->tab.scale.data(), n);
    if (inv_scale) memcpy(inv_scale, c->tab.invScale.data(), n);
    if (sigma2) memcpy(sigma2, c->tab.sigma2.data(), n);
    if (inv_sigma2) memcpy(inv_sigma2, c->tab.invSigma2.data(), n);
    return YGZF_OK;
}

int ygzf_get_features_per_level(const ygzf_ctx *c, int *nfeat) {
    if (!c || !nfeat) return YGZF_ERR_INVALID;
    memcpy(nfeat, c->tab.nFeat.data(), sizeof(int) * c->tab.cfg.nlevels);
    return YGZF_OK;
}

int ygzf_level_size(const ygzf_ctx *c, int w, int h, int level, int *lw, int *lh) {
    if (!c || !lw || !lh || level < 0 || level >= c->tab.cfg.nlevels) return YGZF_ERR_INVALID;
    c->tab.levelSize(w, h, level, lw, lh);
    return YGZF_OK;
}

int ygzf_host_row_pitch(int w) { return w > 0 ? align_up(w, 64) : YGZF_ERR_INVALID; }

int ygzf_max_keypoints(const ygzf_ctx *c_, int w, int h) {
    ygzf_ctx *c = const_cast<ygzf_ctx *>(c_);
    if (!c) return YGZF_ERR_INVALID;
    if (c->geo.w == w && c->geo.h == h) return c->geo.kpStride;
    Geometry G;
    int rc = build_geometry(c, w, h, G);
    return rc ? rc : G.kpStride;
}

int ygzf_sync(ygzf_ctx *c) {
    if (!c) return YGZF_ERR_INVALID;
    HIPCHECK(c, hipStreamSynchronize(c->stream));
    return YGZF_OK;
}

int ygzf_compute_pyramid(ygzf_ctx *c, const uint8_t *img, int w, int h, int stride, uint8_t *const *levels_out) {
    if (!c || !img || !levels_out) return fail(c, YGZF_ERR_INVALID, "null argument");
    HIPCHECK(c, hipSetDevice(c->device));
    int rc = apply_geometry(c, w, h, 1);
    if (rc) return rc;
    FrameSet fs;
    // extract-ahead counts as an extraction for ygzf_align_batch_prev: the previous extraction's last pyramid (the reference image of pair 0)
    // leaves dPyr before this frame's pyramid is written over it
    bool pyramidCarried = false;
    if (c->extractAhead && c->streamCopy && c->alignCarry && c->geo.pyrBytes > 0 && c->carryValid && c->lastFrames > 0) {
        if ((rc = ensure(c, c->dCarryPyr, (size_t) c->geo.pyrBytes + 256))) return rc;
        HIPCHECK(c, hipMemcpyAsync(c->dCarryPyr.p, (uint8_t *) c->dPyr.p + (size_t) (c->lastFrames - 1) * c->geo.pyrBytes, (size_t) c->geo.pyrBytes,
                                   hipMemcpyDeviceToDevice, c->stream));
        pyramidCarried = true;
    }
    if ((rc = upload_frames(c, img, 1, w, h, stride, 0, &fs))) return rc;
    const Geometry &G = c->geo;
    const int L = c->tab.cfg.nlevels;
    if ((rc = pyramid_chain(c, fs, 1)) || (rc = mark_pyramid_done(c))) return rc;
    HIPCHECK(c, hipGetLastError());
    c->pyrHeld = true;
    c->pyrHeldW = w;
    c->pyrHeldH = h;
    // The levels go back tight (pitch = width) into caller memory that is pageable as a rule (cv::Mat buffers).  Eight pitched device-to-host
    // copies took 10-13 ms for a 752x480 pyramid (the copy engine works an odd-width pitched copy off row by row, into page-locked memory
    // as well): the levels are packed on the device and leave in one linear copy through the context's page-locked staging buffer.
    // Level 0 IS the caller's image: it is copied on the host while the other levels are on the link.
    unsigned offs[kMaxLevels + 1];
    offs[0] = offs[1] = 0;
    for (int l = 1; l < L; l++) offs[l + 1] = offs[l] + (unsigned) G.lv[l].w * (unsigned) G.lv[l].h;
    const size_t total = offs[L];
    if ((rc = ensure_stage(c, total + 64)) || (rc = ensure(c, c->dTmpC, total + 64))) return rc;
    launch_pack_levels(c->stream, fs, (const LevelGeom *) c->dGeom.p, 1, L, offs, (uint8_t *) c->dTmpC.p);
    HIPCHECK(c, hipGetLastError());
    // extract-ahead: FAST / octree / descriptors of this image are queued behind the pyramid now and run while the levels travel back on
    // the copy stream and the caller works on them (a Frame constructor clones them); ygzf_extract_resident then only collects the results.
    // (Measured alternatives: the copy on the context's own stream with an event behind it and the extraction queued after that event --
    // the event is reported 30 us later than on a stream that ends there; two half copies to overlap the host's copy-out -- slower too.)
    hipStream_t rd = c->stream;
    bool ahead = false;
    if (c->extractAhead && c->streamCopy) {
        HIPCHECK(c, hipEventRecord(c->evPyramid, c->stream));
        HIPCHECK(c, hipStreamWaitEvent(c->streamCopy, c->evPyramid, 0));
        rd = c->streamCopy;
        ahead = true;
    }
    if (total) HIPCHECK(c, hipMemcpyAsync(c->hStage, c->dTmpC.p, total, hipMemcpyDeviceToHost, rd));
    if (ahead && (rc = run_extract(c, fs, 1, true, pyramidCarried))) return rc;   // (queued after the copy so that the copy starts while these launches are issued)
    if (levels_out[0] != img || stride != w)
        for (int y = 0; y < h; y++) memcpy(levels_out[0] + (size_t) y * w, img + (size_t) y * stride, (size_t) w);
    HIPCHECK(c, hipStreamSynchronize(rd));
    for (int l = 1; l < L; l++) memcpy(levels_out[l], c->hStage + offs[l], offs[l + 1] - offs[l]);
    if (ahead) {
        c->pyrResident = true;
        c->aheadPending = true;
        c->pyrResW = w;
        c->pyrResH = h;
        return YGZF_OK;
    }
    c->lastFrames = 0;
    c->pyrResident = true;
    c->pyrResW = w;
    c->pyrResH = h;
    return YGZF_OK;
}

int ygzf_extract_batch_device(ygzf_ctx *c, const uint8_t *d_imgs, int n_frames, int w, int h, int row_pitch, size_t frame_stride) {
    if (!c || !d_imgs) return fail(c, YGZF_ERR_INVALID, "null argument");
    if (row_pitch < w) return fail(c, YGZF_ERR_INVALID, "row_pitch %d < width %d", row_pitch, w);
    if (((uintptr_t) d_imgs & 3) || (row_pitch & 3) || (frame_stride & 3))
        return fail(c, YGZF_ERR_UNSUPPORTED, "device frames must be 4-byte aligned (base pointer, row_pitch and frame_stride multiples of 4)");
    HIPCHECK(c, hipSetDevice(c->device));
    int rc = apply_geometry(c, w, h, n_frames);
    if (rc) return rc;
    FrameSet fs;
    fs.img0 = d_imgs;
    fs.img0_stride = (long long) frame_stride;
    fs.img0_pitch = row_pitch;
    fs.pyr = (uint8_t *) c->dPyr.p;
    fs.pyr_stride = c->geo.pyrBytes;
    return run_extract(c, fs, n_frames);
}

int ygzf_extract_batch_host(ygzf_ctx *c, const uint8_t *imgs, int n_frames, int w, int h, int row_pitch, size_t frame_stride) {
    if (!c || !imgs) return fail(c, YGZF_ERR_INVALID, "null argument");
    if (row_pitch < w) return fail(c, YGZF_ERR_INVALID, "row_pitch %d < width %d", row_pitch, w);
    HIPCHECK(c, hipSetDevice(c->device));
    int rc = apply_geometry(c, w, h, n_frames);
    if (rc) return rc;
    FrameSet fs;
    if ((rc = upload_frames(c, imgs, n_frames, w, h, row_pitch, frame_stride, &fs))) return rc;
    return run_extract(c, fs, n_frames);
}

// The same from a list of frame pointers (frames that do not sit at one stride: the rows of an array of cv::Mat, a round-robin share of a
// larger clip).  Each frame is one (asynchronous, when the memory is page-locked) copy on the context's stream; the call returns once the
// extraction is queued, like ygzf_extract_batch_host.
int ygzf_extract_batch_host_frames(ygzf_ctx *c, const uint8_t *const *frames, int n_frames, int w, int h, int row_pitch) {
    if (!c || !frames) return fail(c, YGZF_ERR_INVALID, "null argument");
    if (n_frames < 1) return fail(c, YGZF_ERR_INVALID, "no frames");
    if (row_pitch < w) return fail(c, YGZF_ERR_INVALID, "row_pitch %d < width %d", row_pitch, w);
    for (int f = 0; f < n_frames; f++)
        if (!frames[f]) return fail(c, YGZF_ERR_INVALID, "frame %d is null", f);
    HIPCHECK(c, hipSetDevice(c->device));
    int rc = apply_geometry(c, w, h, n_frames);
    if (rc) return rc;
    c->pyrResident = false;
    c->aheadPending = false;
    c->pyrHeld = false;
    const int pitch = align_up(w, 64);
    if ((rc = ensure(c, c->dImg0, (size_t) n_frames * pitch * h))) return rc;
    // frames in page-locked, device-visible host memory are read by a kernel over the link (one launch per 128 frames instead of one pitched copy per
    // frame); anything else goes through the copy engine frame by frame
    bool gathered = false;
    const char *modeEnv = getenv("YGZF_HOST_FRAMES_MODE");      // 1: the kernel reads the host frames itself; 2 (default): one linear copy per frame + one re-pitch launch; 0: pitched copies
    const int mode = modeEnv ? atoi(modeEnv) : 2;
    if (mode == 2) {
        // Into a tight staging buffer with the copy engine's fast paths, then ONE launch per 128 frames that lays the rows out at the context's
        // pitch.  Frames that come as runs of u back-to-back frames at one distance S (a device slot's round-robin share of a tight clip: units of u
        // frames, S = slots x u frames) go up as ONE two-dimensional copy whose "rows" are the runs; anything else as one linear copy per frame.
        const size_t frameBytes = (size_t) (h - 1) * row_pitch + w;
        const size_t tight = (size_t) h * row_pitch;
        int u = n_frames;
        for (int f = 1; f < n_frames; f++)
            if (frames[f] != frames[f - 1] + tight) { u = f; break; }
        const int runs = n_frames / u;
        bool regular = n_frames % u == 0;
        ptrdiff_t S = 0;
        if (regular && runs > 1) {
            S = frames[u] - frames[0];
            regular = S > 0 && (size_t) S >= (size_t) u * tight;
            for (int f = 0; f < n_frames && regular; f++) regular = frames[f] == frames[0] + (ptrdiff_t) (f / u) * S + (ptrdiff_t) (f % u) * (ptrdiff_t) tight;
        }
        if (regular && row_pitch == pitch && pitch != w && (w & 3) == 0 && ((uintptr_t) frames[0] & 3) == 0 && (S & 3) == 0) {
            // frames that carry the device's pitch (ygzf_host_row_pitch): the runs go straight to their place, no staging and no re-pitch launch
            const size_t width = (size_t) u * tight - (size_t) (pitch - w);
            HIPCHECK(c, hipMemcpy2DAsync(c->dImg0.p, (size_t) u * tight, frames[0], runs > 1 ? (size_t) S : (size_t) u * tight, width, (size_t) runs, hipMemcpyHostToDevice, c->stream));
            goto uploaded;
        }
        regular = regular && row_pitch == w;
        const size_t slot = regular ? tight : ((frameBytes + 255) & ~(size_t) 255);
        if ((rc = ensure(c, c->dUpStage, slot * n_frames + 64))) return rc;
        if (regular) {
            if (runs > 1) HIPCHECK(c, hipMemcpy2DAsync(c->dUpStage.p, (size_t) u * tight, frames[0], (size_t) S, (size_t) u * tight, (size_t) runs, hipMemcpyHostToDevice, c->stream));
            else HIPCHECK(c, hipMemcpyAsync(c->dUpStage.p, frames[0], (size_t) n_frames * tight, hipMemcpyHostToDevice, c->stream));
        }
        for (int f0 = 0; f0 < n_frames; f0 += kHostFrameListMax) {
            HostFrameList L;
            const int nf = std::min(kHostFrameListMax, n_frames - f0);
            for (int f = 0; f < nf; f++) {
                uint8_t *d = (uint8_t *) c->dUpStage.p + slot * (size_t) (f0 + f);
                if (!regular) HIPCHECK(c, hipMemcpyAsync(d, frames[f0 + f], frameBytes, hipMemcpyHostToDevice, c->stream));
                L.addr[f] = (unsigned long long) (uintptr_t) d;
            }
            launch_gather_host_frames(c->stream, L, nf, (size_t) row_pitch, (uint8_t *) c->dImg0.p + (size_t) f0 * pitch * h, (size_t) pitch, (size_t) pitch * h, w, h);
        }
        HIPCHECK(c, hipGetLastError());
        gathered = true;
    }
    if (mode == 1) {
        gathered = true;
        for (int f0 = 0; f0 < n_frames && gathered; f0 += kHostFrameListMax) {
            HostFrameList L;
            const int nf = std::min(kHostFrameListMax, n_frames - f0);
            for (int f = 0; f < nf && gathered; f++) {
                hipPointerAttribute_t at;
                memset(&at, 0, sizeof at);
                if (hipPointerGetAttributes(&at, frames[f0 + f]) != hipSuccess) { (void) hipGetLastError(); gathered = false; break; }
                if (at.type != hipMemoryTypeHost || !at.devicePointer) { gathered = false; break; }
                L.addr[f] = (unsigned long long) (uintptr_t) at.devicePointer;
            }
            if (!gathered) break;
            launch_gather_host_frames(c->stream, L, nf, (size_t) row_pitch, (uint8_t *) c->dImg0.p + (size_t) f0 * pitch * h, (size_t) pitch, (size_t) pitch * h, w, h);
        }
        if (gathered) HIPCHECK(c, hipGetLastError());
    }
    if (!gathered)
        for (int f = 0; f < n_frames; f++)
            if ((rc = upload_rows(c, (uint8_t *) c->dImg0.p + (size_t) f * pitch * h, (size_t) pitch, frames[f], (size_t) row_pitch, w, (size_t) h))) return rc;
uploaded:
    FrameSet fs;
    fs.img0 = (const uint8_t *) c->dImg0.p;
    fs.img0_stride = (long long) pitch * h;
    fs.img0_pitch = pitch;
    fs.pyr = (uint8_t *) c->dPyr.p;
    fs.pyr_stride = c->geo.pyrBytes;
    return run_extract(c, fs, n_frames);
}

int ygzf_batch_counts(ygzf_ctx *c, int *n_kp) {
    if (!c || !n_kp) return fail(c, YGZF_ERR_INVALID, "null argument");
    if (c->lastFrames < 1) return fail(c, YGZF_ERR_STATE, "no extracted batch");
    HIPCHECK(c, hipMemcpyAsync(n_kp, (int *) c->dOutCnt.p + 1, sizeof(int) * c->lastFrames, hipMemcpyDeviceToHost, c->stream));
    HIPCHECK(c, hipStreamSynchronize(c->stream));
    return YGZF_OK;
}

int ygzf_batch_fetch(ygzf_ctx *c, int frame, ygzf_kp *kps, uint8_t *desc, int cap, int *n_out) {
    if (!c || !n_out) return fail(c, YGZF_ERR_INVALID, "null argument");
    if (c->lastFrames < 1) return fail(c, YGZF_ERR_STATE, "no extracted batch");
    if (frame < 0 || frame >= c->lastFrames) return fail(c, YGZF_ERR_INVALID, "frame %d out of range", frame);
    int n = 0;
    HIPCHECK(c, hipMemcpyAsync(&n, (int *) c->dOutCnt.p + frame + 1, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIPCHECK(c, hipStreamSynchronize(c->stream));
    *n_out = n;
    if (n == 0) return YGZF_OK;
    if (n > cap || !kps || !desc) return fail(c, YGZF_ERR_INVALID, "capacity %d < %d keypoints", cap, n);
    const size_t base = (size_t) (frame + 1) * c->geo.kpStride;
    HIPCHECK(c, hipMemcpyAsync(kps, (ygzf_kp *) c->dOutKp.p + base, sizeof(ygzf_kp) * n, hipMemcpyDeviceToHost, c->stream));
    HIPCHECK(c, hipMemcpyAsync(desc, (uint8_t *) c->dOutDesc.p + base * 32, (size_t) 32 * n, hipMemcpyDeviceToHost, c->stream));
    HIPCHECK(c, hipStreamSynchronize(c->stream));
    return YGZF_OK;
}

int ygzf_batch_fetch_all(ygzf_ctx *c, ygzf_kp *kps, uint8_t *desc, int *n_kp, int stride) {
    if (!c || !kps || !desc || !n_kp) return fail(c, YGZF_ERR_INVALID, "null argument");
    if (c->lastFrames < 1) return fail(c, YGZF_ERR_STATE, "no extracted batch");
    if (stride < c->geo.kpStride) return fail(c, YGZF_ERR_INVALID, "stride %d < %d (ygzf_max_keypoints)", stride, c->geo.kpStride);
    const int B = c->lastFrames, ks = c->geo.kpStride;
    HIPCHECK(c, hipMemcpyAsync(n_kp, (int *) c->dOutCnt.p + 1, sizeof(int) * B, hipMemcpyDeviceToHost, c->stream));
    // slot f + 1 holds frame f; rows of `stride` keypoints on the host side, kpStride on the device side
    HIPCHECK(c, hipMemcpy2DAsync(kps, sizeof(ygzf_kp) * (size_t) stride, (ygzf_kp *) c->dOutKp.p + ks, sizeof(ygzf_kp) * (size_t) ks,
                                 sizeof(ygzf_kp) * (size_t) ks, B, hipMemcpyDeviceToHost, c->stream));
    HIPCHECK(c, hipMemcpy2DAsync(desc, 32 * (size_t) stride, (uint8_t *) c->dOutDesc.p + 32 * (size_t) ks, 32 * (size_t) ks, 32 * (size_t) ks, B,
                                 hipMemcpyDeviceToHost, c->stream));
    HIPCHECK(c, hipStreamSynchronize(c->stream));
    return YGZF_OK;
}

// Results of frame 0 of a one-frame extraction with ONE synchronisation: the count and the frame's whole keypoint / descriptor rows (kpStride
// entries, ~60 KB: a microsecond on the link) come back together through the page-locked staging area, the first n entries go on to the
// caller.  ygzf_batch_fetch reads the count, waits, then copies exactly n entries and waits again: two host wake-ups of 10-20 us each, which
// is what a Tracking thread's ORBextractor::operator() spent beside 74 us of kernels.
static int fetch_frame0_packed(ygzf_ctx *c, ygzf_kp *kps, uint8_t *desc, int cap, int *n_out) {
    const size_t ks = (size_t) c->geo.kpStride;
    if (ks == 0) { *n_out = 0; return YGZF_OK; }
    const size_t oK = 256, oD = oK + ((ks * sizeof(ygzf_kp) + 255) & ~(size_t) 255), total = oD + ks * 32;
    int rc = ensure_stage(c, total + 256);
    if (rc) return rc;
    HIPCHECK(c, hipMemcpyAsync(c->hStage, (int *) c->dOutCnt.p + 1, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIPCHECK(c, hipMemcpyAsync(c->hStage + oK, (ygzf_kp *) c->dOutKp.p + ks, ks * sizeof(ygzf_kp), hipMemcpyDeviceToHost, c->stream));   // slot 1 = frame 0
    HIPCHECK(c, hipMemcpyAsync(c->hStage + oD, (uint8_t *) c->dOutDesc.p + ks * 32, ks * 32, hipMemcpyDeviceToHost, c->stream));
    HIPCHECK(c, hipStreamSynchronize(c->stream));
    const int n = *(const int *) c->hStage;
    *n_out = n;
    if (n == 0) return YGZF_OK;
    if (n > cap || !kps || !desc) return fail(c, YGZF_ERR_INVALID, "capacity %d < %d keypoints", cap, n);
    memcpy(kps, c->hStage + oK, sizeof(ygzf_kp) * (size_t) n);
    memcpy(desc, c->hStage + oD, (size_t) 32 * n);
    return YGZF_OK;
}

int ygzf_extract(ygzf_ctx *c, const uint8_t *img, int w, int h, int stride, ygzf_kp *kps, uint8_t *desc, int cap, int *n_out) {
    if (!c || !n_out) return fail(c, YGZF_ERR_INVALID, "null argument");
    *n_out = 0;
    if (!img || w <= 0 || h <= 0) return YGZF_OK;  // reference: `if (_image.empty()) return;`
    int rc = ygzf_extract_batch_host(c, img, 1, w, h, stride, 0);
    if (rc) return rc;
    return fetch_frame0_packed(c, kps, desc, cap, n_out);
}

int ygzf_extract_resident(ygzf_ctx *c, ygzf_kp *kps, uint8_t *desc, int cap, int *n_out) {
    if (!c || !n_out) return fail(c, YGZF_ERR_INVALID, "null argument");
    *n_out = 0;
    if (!c->pyrResident) return fail(c, YGZF_ERR_STATE, "no resident pyramid (ygzf_compute_pyramid must be the context's previous image operation)");
    HIPCHECK(c, hipSetDevice(c->device));
    const int w = c->pyrResW, h = c->pyrResH;
    int rc = apply_geometry(c, w, h, 1);   // same geometry as the compute_pyramid call: no reallocation
    if (rc) return rc;
    FrameSet fs;
    fs.img0 = (const uint8_t *) c->dImg0.p;
    fs.img0_pitch = align_up(w, 64);
    fs.img0_stride = (long long) fs.img0_pitch * h;
    fs.pyr = (uint8_t *) c->dPyr.p;
    fs.pyr_stride = c->geo.pyrBytes;
    if (c->aheadPending) {   // queued by ygzf_compute_pyramid (ygzf_set_extract_ahead): nothing to launch
        c->aheadPending = false;
        c->pyrResident = false;
    } else if ((rc = run_extract(c, fs, 1, true))) return rc;
    return fetch_frame0_packed(c, kps, desc, cap, n_out);
}

int ygzf_batch_fetch_level(ygzf_ctx *c, int frame, int level, uint8_t *out) {
    if (!c || !out) return fail(c, YGZF_ERR_INVALID, "null argument");
    if (c->lastFrames < 1) return fail(c, YGZF_ERR_STATE, "no extracted batch");
    if (frame < 0 || frame >= c->lastFrames || level < 0 || level >= c->tab.cfg.nlevels) return fail(c, YGZF_ERR_INVALID, "bad frame/level");
    int pitch;
    const LevelGeom &g = c->geo.lv[level];
    const uint8_t *p = level_ptr(c->lastFs, g, level, frame, &pitch);
    // packed on the device, then one linear copy (a pitched copy of an odd-width level is executed row by row)
    const size_t bytes = (size_t) g.w * g.h;
    int rc = ensure(c, c->dTmpC, bytes + 64);
    if (rc) return rc;
    launch_repitch_rows(c->stream, p, (size_t) pitch, (uint8_t *) c->dTmpC.p, (size_t) g.w, g.w, (size_t) g.h);
    HIPCHECK(c, hipGetLastError());
    HIPCHECK(c, hipMemcpyAsync(out, c->dTmpC.p, bytes, hipMemcpyDeviceToHost, c->stream));
    HIPCHECK(c, hipStreamSynchronize(c->stream));
    return YGZF_OK;
}

int ygzf_batch_fetch_candidates(ygzf_ctx *c, int frame, int level, int *xs, int *ys, int *scores, int cap, int *n_out) {
    if (!c || !n_out) return fail(c, YGZF_ERR_INVALID, "null argument");
    if (c->lastFrames < 1) return fail(c, YGZF_ERR_STATE, "no extracted batch");
    if (frame < 0 || frame >= c->lastFrames || level < 0 || level >= c->tab.cfg.nlevels) return fail(c, YGZF_ERR_INVALID, "bad frame/level");
    const Geometry &G = c->geo;
    const LevelGeom &g = G.lv[level];
    const int nc = g.nCols * g.nRows;
    *n_out = 0;
    if (nc == 0) return YGZF_OK;
    std::vector<unsigned short> cnt(nc);
    std::vector<unsigned> sl((size_t) nc * g.slotCap);
    HIPCHECK(c, hipMemcpyAsync(cnt.data(), (unsigned short *) c->dCellCnt.p + (size_t) frame * G.totalCells + g.cellBase,
                               sizeof(unsigned short) * nc, hipMemcpyDeviceToHost, c->stream));
    HIPCHECK(c, hipMemcpyAsync(sl.data(), (unsigned *) c->dSlots.p + (size_t) frame * G.totalSlots + g.slotBase,
                               sizeof(unsigned) * sl.size(), hipMemcpyDeviceToHost, c->stream));
    HIPCHECK(c, hipStreamSynchronize(c->stream));
    int n = 0;
    for (int ci = 0; ci < nc; ci++) {
        const int i = ci / g.nCols, j = ci % g.nCols;
        for (int k = 0; k < cnt[ci]; k++, n++) {
            if (n < cap && xs && ys && scores) {
                const unsigned e = sl[(size_t) ci * g.slotCap + k];
                xs[n] = (int) (e & 255u) + j * g.wCell;
                ys[n] = (int) ((e >> 8) & 255u) + i * g.hCell;
                scores[n] = (int) (e >> 16);
            }
        }
    }
    *n_out = n;
    return n > cap ? fail(c, YGZF_ERR_INVALID, "capacity %d < %d candidates", cap, n) : YGZF_OK;
}

int ygzf_batch_fetch_level_keypoints(ygzf_ctx *c, int frame, int level, int *xs, int *ys, int *scores, int cap, int *n_out) {
    if (!c || !n_out) return fail(c, YGZF_ERR_INVALID, "null argument");
    if (c->lastFrames < 1) return fail(c, YGZF_ERR_STATE, "no extracted batch");
    const int L = c->tab.cfg.nlevels;
    if (frame < 0 || frame >= c->lastFrames || level < 0 || level >= L) return fail(c, YGZF_ERR_INVALID, "bad frame/level");
    const Geometry &G = c->geo;
    const LevelGeom &g = G.lv[level];
    int n = 0;
    HIPCHECK(c, hipMemcpyAsync(&n, (int *) c->dLvlCnt.p + frame * L + level, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIPCHECK(c, hipStreamSynchronize(c->stream));
    *n_out = n;
    if (n == 0) return YGZF_OK;
    if (n > cap) return fail(c, YGZF_ERR_INVALID, "capacity %d < %d keypoints", cap, n);
    std::vector<unsigned> xy(n);
    std::vector<unsigned char> sc(n);
    const size_t base = (size_t) frame * G.kpStride + g.kpBase;
    HIPCHECK(c, hipMemcpyAsync(xy.data(), (unsigned *) c->dLvlXY.p + base, sizeof(unsigned) * n, hipMemcpyDeviceToHost, c->stream));
    HIPCHECK(c, hipMemcpyAsync(sc.data(), (unsigned char *) c->dLvlScore.p + base, n, hipMemcpyDeviceToHost, c->stream));
    HIPCHECK(c, hipStreamSynchronize(c->stream));
    for (int i = 0; i < n; i++) {
        xs[i] = (int) (xy[i] & 0xFFFFu);
        ys[i] = (int) (xy[i] >> 16);
        scores[i] = sc[i];
    }
    return YGZF_OK;
}

int ygzf_descriptor_distance(ygzf_ctx *c, const uint8_t *a, const uint8_t *b, int n, int *dist) {
    if (!c || !a || !b || !dist || n < 0) return fail(c, YGZF_ERR_INVALID, "null argument");
    if (n == 0) return YGZF_OK;
    HIPCHECK(c, hipSetDevice(c->device));
    int rc;
    if ((size_t) n * 68 <= kPackedMax) {   // a frame's worth of pairs: one packed copy each way
        PackedTransfer P(c);
        const size_t iA = P.add_in(a, (size_t) n * 32), iB = P.add_in(b, (size_t) n * 32), oD = P.add_out(dist, (size_t) n * sizeof(int));
        uint8_t *d;
        if ((rc = P.upload(&d))) return rc;
        {
            ProfScope ps(c, KK_HAMMING);
            launch_hamming_pairs(c->stream, d + iA, d + iB, n, (int *) P.d_out(oD));
        }
        HIPCHECK(c, hipGetLastError());
        return P.download();
    }
    if ((rc = ensure(c, c->dTmpA, (size_t) n * 32)) || (rc = ensure(c, c->dTmpB, (size_t) n * 32)) ||
        (rc = ensure(c, c->dTmpC, (size_t) n * sizeof(int))))
        return rc;
    HIPCHECK(c, hipMemcpyAsync(c->dTmpA.p, a, (size_t) n * 32, hipMemcpyHostToDevice, c->stream));
    HIPCHECK(c, hipMemcpyAsync(c->dTmpB.p, b, (size_t) n * 32, hipMemcpyHostToDevice, c->stream));
    {
        ProfScope ps(c, KK_HAMMING);
        launch_hamming_pairs(c->stream, c->dTmpA.p, c->dTmpB.p, n, (int *) c->dTmpC.p);
    }
    HIPCHECK(c, hipGetLastError());
    HIPCHECK(c, hipMemcpyAsync(dist, c->dTmpC.p, (size_t) n * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIPCHECK(c, hipStreamSynchronize(c->stream));
    return YGZF_OK;
}

int ygzf_timer_start(ygzf_ctx *c) {
    if (!c) return YGZF_ERR_INVALID;
    HIPCHECK(c, hipEventRecord(c->tStart, c->stream));
    return YGZF_OK;
}

int ygzf_timer_stop(ygzf_ctx *c, float *elapsed_ms) {
    if (!c || !elapsed_ms) return YGZF_ERR_INVALID;
    HIPCHECK(c, hipEventRecord(c->tStop, c->stream));
    HIPCHECK(c, hipEventSynchronize(c->tStop));
    HIPCHECK(c, hipEventElapsedTime(elapsed_ms, c->tStart, c->tStop));
    return YGZF_OK;
}

int ygzf_profile_enable(ygzf_ctx *c, int on) {
    if (!c) return YGZF_ERR_INVALID;
    drain_profile(c);
    c->profile = on != 0;
    return YGZF_OK;
}

int ygzf_profile_read(ygzf_ctx *c, const char **names, float *total_ms, int *launches, int cap) {
    if (!c) return YGZF_ERR_INVALID;
    drain_profile(c);
    for (int k = 0; k < KK_COUNT && k < cap; k++) {
        if (names) names[k] = kKernelNames[k];
        if (total_ms) total_ms[k] = c->profMs[k];
        if (launches) launches[k] = c->profN[k];
    }
    return KK_COUNT;
}

int ygzf_profile_reset(ygzf_ctx *c) {
    if (!c) return YGZF_ERR_INVALID;
    drain_profile(c);
    for (int k = 0; k < KK_COUNT; k++) { c->profMs[k] = 0; c->profN[k] = 0; }
    return YGZF_OK;
}

void *ygzf_stream(ygzf_ctx *c) { return c ? (void *) c->stream : nullptr; }

// ---- matcher ----------------------------------------------------------------------------------------------------------
// The reference patches of all levels in a kernel of their own (k_sia_precompute) pay for launches of a few pairs -- one pair: 365 -> 305 us, the
// 71 us the pair's one workgroup spent on them become 10 us chip-wide -- and cost large ones: at 256 pairs the in-kernel form overlaps one pair's
// patches with another pair's solve on the same CU and its single cache stays L2-resident (971 against 1075 us per launch, 142 k against 134 k frames/s).
constexpr int kSiaPerLevelPairs = 32;

static void fill_camera(MatchArgs &A, const ygzf_camera *cam, const ygzf_ctx *c) {
    A.fx = cam->fx; A.fy = cam->fy; A.cx = cam->cx; A.cy = cam->cy; A.mb = cam->mb; A.mbf = cam->mbf;
    A.minX = cam->min_x; A.minY = cam->min_y; A.maxX = cam->max_x; A.maxY = cam->max_y;
    A.gridInvW = (float) 64 / (cam->max_x - cam->min_x);  // mfGridElementWidthInv, src/Frame.cc:302-303
    A.gridInvH = (float) 48 / (cam->max_y - cam->min_y);
    for (int l = 0; l < kMaxLevels; l++) A.scaleFactors[l] = l < c->tab.cfg.nlevels ? c->tab.scale[l] : 1.f;
}

static int plan_match_lds(ygzf_ctx *c, MatchArgs &A, int nPairs, size_t *ldsBytes) {
    const size_t budget = 156 * 1024;
    if (A.capCur > 65535) return fail(c, YGZF_ERR_UNSUPPORTED, "matcher supports at most 65535 keypoints per frame");
    A.qpInLds = 0;
    // plans in order of preference: everything in LDS; descriptors in global; + speculative lists in global; + misc arrays
    const struct { int desc, spill; } plans[] = {{1, 0}, {0, 0}, {0, kSpillSpec}, {0, kSpillSpec | kSpillMisc}};
    size_t b = 0, sp = 0;
    bool ok = false;
    for (const auto &pl : plans) {
        b = match_lds_bytes(A.capCur, A.capLast, pl.desc != 0, pl.spill, &sp, A.specDeep != 0);
        if (b <= budget) { A.descInLds = pl.desc; A.spill = pl.spill; ok = true; break; }
    }
    if (!ok) return fail(c, YGZF_ERR_UNSUPPORTED, "matcher needs %zu bytes of LDS for %d/%d keypoints", b, A.capCur, A.capLast);
    int rc = ensure(c, c->dQp, (size_t) nPairs * A.capLast * 32);
    if (rc) return rc;
    A.qpScratch = c->dQp.p;
    A.spillStride = (long long) sp;
    A.spillScratch = nullptr;
    if (sp) {
        if ((rc = ensure(c, c->dSpill, (size_t) nPairs * sp))) return rc;
        A.spillScratch = c->dSpill.p;
    }
    // few pairs in the launch (a Tracking thread matches ONE): spread each over several workgroups (kernels.h, MatchArgs::split)
    if (!c->dMatchStat.p) {
        if ((rc = ensure(c, c->dMatchStat, 64))) return rc;
        HIPCHECK(c, hipMemsetAsync(c->dMatchStat.p, 0, 64, c->stream));
    }
    A.serialFallbacks = (unsigned *) c->dMatchStat.p;
    A.serialOrder = c->matchSerial;
    A.handoverFence = c->matchFence;
    A.fixedLanes = c->matchFixedLanes;
    A.split = 1;
    A.splitCnt = nullptr;
    A.splitX = nullptr;
    if (!A.spill && A.capLast >= 128 && c->matchSplit != 1) {
        int sp2 = c->matchSplit > 1 ? c->matchSplit : 256 / (nPairs > 0 ? nPairs : 1);
        sp2 = std::min(sp2, c->matchSplit > 1 ? 128 : 64);   // (one pair: 64 workgroups of 16 queries, four waves each in the scan: 35.5 us against 41.7 at 8)
        if (sp2 > 1) {
            const size_t cntBytes = (size_t) nPairs * sizeof(int);
            if (c->dSplitCnt.bytes < cntBytes) {   // counters are zero between launches: a fresh buffer is cleared once
                if ((rc = ensure(c, c->dSplitCnt, cntBytes > 4096 ? cntBytes : 4096))) return rc;
                HIPCHECK(c, hipMemsetAsync(c->dSplitCnt.p, 0, c->dSplitCnt.bytes, c->stream));
            }
            if ((rc = ensure(c, c->dSplitX, (size_t) nPairs * A.capLast * kMatchSplitRec))) return rc;
            A.split = sp2;
            A.splitCnt = (int *) c->dSplitCnt.p;
            A.splitX = (unsigned char *) c->dSplitX.p;
        }
    }
    HIPCHECK(c, match_prepare(b));
    *ldsBytes = b;
    return YGZF_OK;
}

int ygzf_match_batch_prev(ygzf_ctx *c, const ygzf_camera *cam, float th, int b_mono, int check_level, int check_orientation) {
    if (!c || !cam) return fail(c, YGZF_ERR_INVALID, "null argument");
    if (c->lastFrames < 1) return fail(c, YGZF_ERR_STATE, "no extracted batch");
    HIPCHECK(c, hipSetDevice(c->device));
    const Geometry &G = c->geo;
    const int B = c->lastFrames;
    if (G.kpStride == 0) return fail(c, YGZF_ERR_STATE, "configuration yields no keypoints");
    int rc;
    if ((rc = ensure(c, c->dWorld, (size_t) (B + 1) * G.kpStride * 3 * sizeof(float))) ||
        (rc = ensure(c, c->dOwner, (size_t) B * G.kpStride)) || (rc = ensure(c, c->dMatch, (size_t) B * G.kpStride * sizeof(int))) ||
        (rc = ensure(c, c->dNMatch, (size_t) B * sizeof(int))) || (rc = ensure(c, c->dPoses, (size_t) B * 24 * sizeof(float))))
        return rc;
    // identity poses for every pair (uploaded once per buffer / batch size, so the steady state has no host sync)
    if (c->identityPoses < B || c->identityPosesPtr != c->dPoses.p) {
        std::vector<float> poses((size_t) B * 24, 0.f);
        for (int p = 0; p < B; p++) {
            float *q = &poses[(size_t) p * 24];
            q[0] = q[4] = q[8] = 1.f;
            q[12] = q[16] = q[20] = 1.f;
        }
        HIPCHECK(c, hipMemcpyAsync(c->dPoses.p, poses.data(), poses.size() * sizeof(float), hipMemcpyHostToDevice, c->stream));
        HIPCHECK(c, hipStreamSynchronize(c->stream));  // `poses` goes out of scope
        c->identityPoses = B;
        c->identityPosesPtr = c->dPoses.p;
    }
    const ygzf_kp *kp = (const ygzf_kp *) c->dOutKp.p;
    const uint8_t *desc = (const uint8_t *) c->dOutDesc.p;
    const int *cnt = (const int *) c->dOutCnt.p;
    MatchArgs A;
    memset(&A, 0, sizeof A);
    A.maxDist = 100;   // TH_HIGH
    A.unitWorld = 1;   // world point of a Last keypoint = its back-projection to depth 1, computed where it is used (a launch of its own until round 4)
    A.curKeys = kp + G.kpStride;            // pair p: Cur = slot p+1, Last = slot p
    A.curDesc = desc + (size_t) G.kpStride * 32;
    A.curURight = nullptr;
    A.curCnt = cnt;
    A.kpStrideCur = G.kpStride;
    A.cntStrideCur = 1;
    A.cntOffCur = 1;
    A.ownerIn = nullptr;
    A.lastKeys = kp;
    A.mpDesc = desc;
    A.world = (const float *) c->dWorld.p;
    A.lastCnt = cnt;
    A.kpStrideLast = G.kpStride;
    A.cntStrideLast = 1;
    A.cntOffLast = 0;
    A.poses = (const float *) c->dPoses.p;
    fill_camera(A, cam, c);
    A.th = th;
    A.bMono = b_mono != 0;
    A.checkLevel = check_level != 0;
    A.checkOri = check_orientation != 0;
    A.owner = (uint8_t *) c->dOwner.p;
    A.match = (int *) c->dMatch.p;
    A.nmatches = (int *) c->dNMatch.p;
    A.capCur = G.kpStride;
    A.capLast = G.kpStride;
    if (c->matchDebug) {
        if ((rc = ensure(c, c->dTmpC, (size_t) B * 8 * sizeof(long long)))) return rc;
        A.dbg = (long long *) c->dTmpC.p;
    }
    size_t lds;
    if ((rc = plan_match_lds(c, A, B, &lds))) return rc;
    {
        hipStream_t sm = fill_begin(c);
        {
            ProfScope ps(c, KK_MATCH, sm);
            launch_match_last(sm, A, B, lds);
        }
        fill_end(c);
    }
    HIPCHECK(c, hipGetLastError());
    if (A.dbg) {
        long long st[8];
        const int pp = B > 1 ? 1 : 0;
        HIPCHECK(c, hipMemcpy(st, A.dbg + 8 * pp, sizeof st, hipMemcpyDeviceToHost));
        fprintf(stderr, "[ygzf match pair %d, 100MHz ticks] grid %lld  proj %lld  spec %lld  seq %lld  tail %lld  rescans %lld of %lld queries\n", pp,
                st[1] - st[0], st[2] - st[1], st[3] - st[2], st[4] - st[3], st[5] - st[4], st[6], st[7]);
    }
    c->lastMatchPairs = B;
    return YGZF_OK;
}

int ygzf_match_fallbacks(ygzf_ctx *c, unsigned *pairs) {
    if (!c || !pairs) return fail(c, YGZF_ERR_INVALID, "null argument");
    *pairs = 0;
    if (!c->dMatchStat.p) return YGZF_OK;   // no matcher launch yet
    HIPCHECK(c, hipSetDevice(c->device));
    HIPCHECK(c, hipMemcpyAsync(pairs, c->dMatchStat.p, sizeof(unsigned), hipMemcpyDeviceToHost, c->stream));
    HIPCHECK(c, hipStreamSynchronize(c->stream));
    return YGZF_OK;
}

int ygzf_match_counts(ygzf_ctx *c, int *nmatches) {
    if (!c || !nmatches) return fail(c, YGZF_ERR_INVALID, "null argument");
    if (c->lastMatchPairs < 1) return fail(c, YGZF_ERR_STATE, "no matched batch");
    HIPCHECK(c, hipMemcpyAsync(nmatches, c->dNMatch.p, sizeof(int) * c->lastMatchPairs, hipMemcpyDeviceToHost, c->stream));
    HIPCHECK(c, hipStreamSynchronize(c->stream));
    return YGZF_OK;
}

int ygzf_match_fetch(ygzf_ctx *c, int frame, int *cur_match, uint8_t *cur_owner, int cap) {
    if (!c) return fail(c, YGZF_ERR_INVALID, "null argument");
    if (c->lastMatchPairs < 1) return fail(c, YGZF_ERR_STATE, "no matched batch");
    if (frame < 0 || frame >= c->lastMatchPairs) return fail(c, YGZF_ERR_INVALID, "frame %d out of range", frame);
    int n = 0;
    HIPCHECK(c, hipMemcpyAsync(&n, (int *) c->dOutCnt.p + frame + 1, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIPCHECK(c, hipStreamSynchronize(c->stream));
    if (n > cap) return fail(c, YGZF_ERR_INVALID, "capacity %d < %d keypoints", cap, n);
    const size_t base = (size_t) frame * c->geo.kpStride;
    if (cur_match && n) HIPCHECK(c, hipMemcpyAsync(cur_match, (int *) c->dMatch.p + base, sizeof(int) * n, hipMemcpyDeviceToHost, c->stream));
    if (cur_owner && n) HIPCHECK(c, hipMemcpyAsync(cur_owner, (uint8_t *) c->dOwner.p + base, n, hipMemcpyDeviceToHost, c->stream));
    HIPCHECK(c, hipStreamSynchronize(c->stream));
    return YGZF_OK;
}

int ygzf_match_fetch_all(ygzf_ctx *c, int *match, int stride) {
    if (!c || !match) return fail(c, YGZF_ERR_INVALID, "null argument");
    if (c->lastMatchPairs < 1) return fail(c, YGZF_ERR_STATE, "no matched batch");
    const int B = c->lastMatchPairs, ks = c->geo.kpStride;
    if (stride < ks) return fail(c, YGZF_ERR_INVALID, "stride %d < %d (ygzf_max_keypoints)", stride, ks);
    HIPCHECK(c, hipMemcpy2DAsync(match, sizeof(int) * (size_t) stride, c->dMatch.p, sizeof(int) * (size_t) ks, sizeof(int) * (size_t) ks, B,
                                 hipMemcpyDeviceToHost, c->stream));
    HIPCHECK(c, hipStreamSynchronize(c->stream));
    return YGZF_OK;
}

int ygzf_search_by_projection_last(ygzf_ctx *c, const ygzf_frame_view *cur, const ygzf_camera *cam, int last_n, const ygzf_kp *last_keys,
                                   const uint8_t *mp_valid, const uint8_t *outlier, const uint8_t *mp_has_obs, const float *mp_world,
                                   const uint8_t *mp_desc, const float *Rcw, const float *tcw, const float *Rlw, const float *tlw, float th,
                                   int b_mono, int check_level, int check_orientation, uint8_t *cur_owner, int *cur_match, int *nmatches) {
    if (!c || !cur || !cam || !nmatches || !Rcw || !tcw || !Rlw || !tlw) return fail(c, YGZF_ERR_INVALID, "null argument");
    *nmatches = 0;
    if (cur->n < 0 || last_n < 0) return fail(c, YGZF_ERR_INVALID, "negative count");
    if (cur->n == 0 || last_n == 0) {
        for (int i = 0; i < cur->n; i++) if (cur_match) cur_match[i] = -1;
        return YGZF_OK;
    }
    if (!cur->keys || !cur->desc || !last_keys || !mp_world || !mp_desc || !cur_match || !cur_owner)
        return fail(c, YGZF_ERR_INVALID, "null array");
    {   // the kernel indexes its scale-factor table with these octaves and stores Cur's as bytes
        const int nl = cur->scale_factors ? std::min(cur->nlevels, (int) kMaxLevels) : c->tab.cfg.nlevels;
        for (int i = 0; i < last_n; i++)
            if (last_keys[i].octave < 0 || last_keys[i].octave >= nl) return fail(c, YGZF_ERR_INVALID, "last_keys[%d].octave %d outside 0..%d", i, last_keys[i].octave, nl - 1);
        for (int i = 0; i < cur->n; i++)
            if (cur->keys[i].octave < 0 || cur->keys[i].octave >= nl) return fail(c, YGZF_ERR_INVALID, "cur keys[%d].octave %d outside 0..%d", i, cur->keys[i].octave, nl - 1);
    }
    HIPCHECK(c, hipSetDevice(c->device));
    const size_t nt = cur->n, nq = last_n;
    int counts[2] = {cur->n, last_n};
    float pose[24];
    memcpy(pose, Rcw, 36); memcpy(pose + 9, tcw, 12); memcpy(pose + 12, Rlw, 36); memcpy(pose + 21, tlw, 12);
    PackedTransfer P(c);
    const size_t oCurK = P.add_in(cur->keys, nt * sizeof(ygzf_kp)), oCurD = P.add_in(cur->desc, nt * 32), oUR = P.add_in(cur->u_right, cur->u_right ? nt * 4 : 0),
                 oOwn = P.add_in(cur_owner, nt), oLastK = P.add_in(last_keys, nq * sizeof(ygzf_kp)), oMpD = P.add_in(mp_desc, nq * 32),
                 oWorld = P.add_in(mp_world, nq * 12), oValid = P.add_in(mp_valid, mp_valid ? nq : 0), oOutl = P.add_in(outlier, outlier ? nq : 0),
                 oObs = P.add_in(mp_has_obs, mp_has_obs ? nq : 0), oCnt = P.add_in(counts, sizeof counts), oPose = P.add_in(pose, sizeof pose);
    const size_t rOwner = P.add_out(cur_owner, nt), rMatch = P.add_out(cur_match, nt * sizeof(int)), rN = P.add_out(nmatches, sizeof(int));
    int rc;
    uint8_t *dIn;
    if ((rc = P.upload(&dIn))) return rc;
    MatchArgs A;
    memset(&A, 0, sizeof A);
    A.maxDist = 100;   // TH_HIGH
    A.curKeys = (const ygzf_kp *) (dIn + oCurK);
    A.curDesc = dIn + oCurD;
    A.curURight = cur->u_right ? (const float *) (dIn + oUR) : nullptr;
    A.ownerIn = dIn + oOwn;
    A.curCnt = (const int *) (dIn + oCnt);
    A.kpStrideCur = (long long) nt;
    A.cntStrideCur = 0;
    A.cntOffCur = 0;
    A.lastKeys = (const ygzf_kp *) (dIn + oLastK);
    A.mpDesc = dIn + oMpD;
    A.world = (const float *) (dIn + oWorld);
    A.mpValid = mp_valid ? dIn + oValid : nullptr;
    A.outlier = outlier ? dIn + oOutl : nullptr;
    A.hasObs = mp_has_obs ? dIn + oObs : nullptr;
    A.lastCnt = (const int *) (dIn + oCnt);
    A.kpStrideLast = (long long) nq;
    A.cntStrideLast = 0;
    A.cntOffLast = 1;
    A.poses = (const float *) (dIn + oPose);
    fill_camera(A, cam, c);
    if (cur->scale_factors) for (int l = 0; l < kMaxLevels && l < cur->nlevels; l++) A.scaleFactors[l] = cur->scale_factors[l];
    A.th = th;
    A.bMono = b_mono != 0;
    A.checkLevel = check_level != 0;
    A.checkOri = check_orientation != 0;
    A.owner = P.d_out(rOwner);
    A.match = (int *) P.d_out(rMatch);
    A.nmatches = (int *) P.d_out(rN);
    A.capCur = (int) nt;
    A.capLast = (int) nq;
    size_t lds;
    if ((rc = plan_match_lds(c, A, 1, &lds))) return rc;
    if (c->matchDebug) {
        if ((rc = ensure(c, c->dTmpC, 8 * sizeof(long long)))) return rc;
        A.dbg = (long long *) c->dTmpC.p;
    }
    {
        ProfScope ps(c, KK_MATCH);
        launch_match_last(c->stream, A, 1, lds);
    }
    HIPCHECK(c, hipGetLastError());
    if ((rc = P.download())) return rc;
    if (A.dbg) {
        long long st[8];
        HIPCHECK(c, hipMemcpy(st, A.dbg, sizeof st, hipMemcpyDeviceToHost));
        fprintf(stderr, "[ygzf match (cur, last), 100MHz ticks] grid %lld  proj %lld  spec %lld  seq %lld  tail %lld  rescans %lld of %lld queries\n",
                st[1] - st[0], st[2] - st[1], st[3] - st[2], st[4] - st[3], st[5] - st[4], st[6], st[7]);
    }
    c->lastMatchPairs = 0;
    return YGZF_OK;
}

// ---- SparseImgAlign::run ------------------------------------------------------------------------------------------------
// ygzf_sia_run (pyramids from host memory) and ygzf_sia_run_cached (pyramids of two image-cache slots, already on the device).
// cached: ref_slot / cur_slot >= 0, cur carries the pose only (its level arrays are not read, ref's neither).
static FrameSet cache_frameset(const ygzf_ctx *c);
static int sia_run_impl(ygzf_ctx *c, const ygzf_sia_frame *ref, const ygzf_sia_frame *cur, int ref_slot, int cur_slot, const ygzf_camera *cam,
                        const float *inv_scale_factors, int max_level, int min_level, int n_iter, float *TCR_out, size_t *ret, float *info, float *H36) {
    if (!c || !ref || !cur || !cam || !inv_scale_factors || !TCR_out || !ret) return fail(c, YGZF_ERR_INVALID, "null argument");
    *ret = 0;
    const bool cached = ref_slot >= 0;
    if (cached) {
        if (c->cacheSlots <= 0) return fail(c, YGZF_ERR_STATE, "image cache not reserved");
        if (ref_slot >= c->cacheSlots || cur_slot < 0 || cur_slot >= c->cacheSlots || !c->cacheFilled[ref_slot] || !c->cacheFilled[cur_slot])
            return fail(c, YGZF_ERR_INVALID, "slot %d / %d is empty or outside the cache", ref_slot, cur_slot);
        if (c->geo.w != c->cacheW || c->geo.h != c->cacheH) {
            int rc0 = apply_geometry(c, c->cacheW, c->cacheH, 1);
            if (rc0) return rc0;
        }
        if (max_level < min_level || min_level < 0 || max_level >= c->tab.cfg.nlevels)
            return fail(c, YGZF_ERR_INVALID, "bad level range [%d,%d]", min_level, max_level);
    } else if (max_level < min_level || min_level < 0 || max_level >= kMaxLevels || max_level >= ref->nlevels || max_level >= cur->nlevels)
        return fail(c, YGZF_ERR_INVALID, "bad level range [%d,%d]", min_level, max_level);
    if (ref->n < 0) return fail(c, YGZF_ERR_INVALID, "negative count");
    // T_cur_from_ref for the empty case is still cur*ref^-1 in the reference only after the early return; :24-27 returns 0 at once
    if (ref->n == 0) {   // "SparseImgAlign: no features to track!" -> return 0, TCR untouched
        if (info) { info[0] = 0; info[1] = 0; }
        return YGZF_OK;
    }
    if (!ref->keys || !ref->mp_world) return fail(c, YGZF_ERR_INVALID, "null array");
    if (!cached && (!ref->levels || !cur->levels || !ref->level_w || !ref->level_h || !cur->level_w || !cur->level_h))
        return fail(c, YGZF_ERR_INVALID, "null array");
    HIPCHECK(c, hipSetDevice(c->device));
    const size_t N = ref->n;
    int rc;
    ygzf_ctx::Buf *S = c->dSia;   // 4 images, 6 caches; the small arrays cross the link as one packed copy each way (PackedTransfer)
    std::vector<SiaLevel> lv(2 * kMaxLevels);
    memset(lv.data(), 0, lv.size() * sizeof(SiaLevel));
    size_t largestCur = 0;   // the largest current-frame level that may be staged in LDS beside the feature tables
    if (cached) {
        // the slots hold level 0 and the pyramid the resize kernel built from it (what the extractor computed for the same image)
        for (int side = 0; side < 2; side++) {
            FrameSet fs = cache_frameset(c);
            const int slot = side ? cur_slot : ref_slot;
            fs.img0 += (long long) slot * fs.img0_stride;
            fs.pyr += (long long) slot * fs.pyr_stride;
            for (int l = min_level; l <= max_level; l++) {
                const LevelGeom &g = c->geo.lv[l];
                int pitch;
                SiaLevel &L = lv[side * kMaxLevels + l];
                L.img = level_ptr(fs, g, l, 0, &pitch);
                L.w = g.w; L.h = g.h; L.pitch = pitch;
                if (side) largestCur = std::max(largestCur, (size_t) pitch * g.h);
            }
        }
    } else {
        size_t imgBytes = 0;
        for (int l = min_level; l <= max_level; l++) {
            if (ref->level_w[l] < 1 || ref->level_h[l] < 1 || cur->level_w[l] < 1 || cur->level_h[l] < 1 || !ref->levels[l] || !cur->levels[l])
                return fail(c, YGZF_ERR_INVALID, "bad pyramid level %d", l);
            imgBytes += (size_t) ref->level_w[l] * ref->level_h[l] + (size_t) cur->level_w[l] * cur->level_h[l] + 128;
        }
        if ((rc = ensure(c, S[4], imgBytes))) return rc;
        size_t off = 0;
        for (int l = min_level; l <= max_level; l++) {
            for (int side = 0; side < 2; side++) {
                const ygzf_sia_frame *f = side ? cur : ref;
                const size_t b = (size_t) f->level_w[l] * f->level_h[l];
                uint8_t *d = (uint8_t *) S[4].p + off;
                HIPCHECK(c, hipMemcpyAsync(d, f->levels[l], b, hipMemcpyHostToDevice, c->stream));   // Frame clones are tight (step == cols)
                SiaLevel &L = lv[side * kMaxLevels + l];
                L.img = d; L.w = f->level_w[l]; L.h = f->level_h[l]; L.pitch = f->level_w[l];
                off += (b + 63) & ~(size_t) 63;
                if (side) largestCur = std::max(largestCur, b);
            }
        }
    }
    float poses[14];
    memcpy(poses, ref->Tcw, 28);
    memcpy(poses + 7, cur->Tcw, 28);
    float out[48];
    PackedTransfer P(c);
    const size_t oKeys = P.add_in(ref->keys, N * sizeof(ygzf_kp)), oWorld = P.add_in(ref->mp_world, N * 12), oValid = P.add_in(ref->mp_valid, ref->mp_valid ? N : 0),
                 oOutl = P.add_in(ref->outlier, ref->outlier ? N : 0), oPoses = P.add_in(poses, sizeof poses), oLv = P.add_in(lv.data(), lv.size() * sizeof(SiaLevel));
    const size_t rOut = P.add_out(out, sizeof out);
    uint8_t *dIn;
    const size_t nLv = (size_t) (max_level - min_level + 1);
    const bool perLevel = c->siaPerLevel;
    if ((rc = ensure(c, S[6], perLevel ? nLv * N * (52 * sizeof(float) + 1) + N + 64 : N * (16 + 96) * sizeof(float) + N + 64)) || (rc = P.upload(&dIn))) return rc;
    const SiaLevel *dLv = (const SiaLevel *) (dIn + oLv);
    SiaArgs A;
    memset(&A, 0, sizeof A);
    A.keys = (const ygzf_kp *) (dIn + oKeys);
    A.world = (const float *) (dIn + oWorld);
    A.mpValid = ref->mp_valid ? dIn + oValid : nullptr;
    A.outlier = ref->outlier ? dIn + oOutl : nullptr;
    A.kpStride = (long long) N;
    A.nRef = nullptr;
    A.n = (int) N;
    A.poses = (const float *) (dIn + oPoses);
    A.refLv = dLv;
    A.curLv = dLv + kMaxLevels;
    A.lvStride = 0;
    for (int l = 0; l < kMaxLevels; l++) A.invScale[l] = l <= max_level ? inv_scale_factors[l] : 1.f;
    A.fx = cam->fx; A.fy = cam->fy; A.cx = cam->cx; A.cy = cam->cy;
    A.maxLevel = max_level; A.minLevel = min_level; A.nIter = n_iter;
    A.eps = 0.000001f;   // src/SparseImageAlign.cc:17
    A.patchCache = (float *) S[6].p;
    A.jacCache = nullptr;
    A.visible = (uint8_t *) (A.patchCache + N * 48);
    A.momCache = A.patchCache + N * 64;   // (the buffer holds 112 floats per feature)
    if (perLevel) {   // [patch rows: nLv x N x 48 floats | moments: nLv x N x 4 floats | flags: nLv x N bytes | visible: N bytes]
        A.perLevel = 1;
        A.pcLevelStride = N * 48;
        A.momLevelStride = N * 4;
        A.flagLevelStride = N;
        A.momCache = A.patchCache + nLv * N * 48;
        A.levelFlags = (uint8_t *) (A.momCache + nLv * N * 4);
        A.visible = A.levelFlags + nLv * N;
    }
    A.out = (float *) P.d_out(rOut);
    {
        if (c->siaDebug) {
            if ((rc = ensure(c, c->dTmpB, 256))) return rc;
            HIPCHECK(c, hipMemsetAsync(c->dTmpB.p, 0, 256, c->stream));
            A.dbg = (long long *) c->dTmpB.p;
        }
        size_t sl = sia_lds_bytes((int) N);
        A.ldsFeat = (int) N;
        A.jacLds = sia_jac_in_lds((int) N) ? 1 : 0;
        if (sl > 150 * 1024) return fail(c, YGZF_ERR_UNSUPPORTED, "SparseImgAlign supports at most %d features", (int) (150 * 1024 / 24));
        {
            A.stageOff = (int) ((sl + 15) & ~(size_t) 15);
            A.stageBytes = (int) sia_stage_bytes((size_t) A.stageOff, largestCur);
            sl = (size_t) A.stageOff + (size_t) A.stageBytes;
        }
        HIPCHECK(c, sia_prepare(sl));
        ProfScope ps(c, KK_SIA);
        if (A.perLevel) launch_sia_precompute(c->stream, A, 1, (int) N);
        launch_sia(c->stream, A, 1, sl);
    }
    HIPCHECK(c, hipGetLastError());
    if ((rc = P.download())) return rc;
    if (A.dbg) {
        long long st[16];
        HIPCHECK(c, hipMemcpy(st, A.dbg, sizeof st, hipMemcpyDeviceToHost));
        fprintf(stderr, "[ygzf sia, 10ns ticks over %lld iterations] accumulate %lld (last wave %lld)  reduce %lld (wave sums %lld)  solve %lld  precompute(all levels) %lld\n", st[3], st[0], st[6], st[1], st[5], st[2], st[4]);
        fprintf(stderr, "[ygzf sia] ldlt %lld; accumulate per wave:", st[7]); for (int w = 0; w < 8; w++) fprintf(stderr, " %lld", st[8 + w]); fprintf(stderr, "\n");
    }
    memcpy(TCR_out, out, 28);
    *ret = (size_t) out[7];
    if (info) { info[0] = out[8]; info[1] = out[9]; }
    if (H36) memcpy(H36, out + 12, 36 * sizeof(float));
    return YGZF_OK;
}

int ygzf_sia_run(ygzf_ctx *c, const ygzf_sia_frame *ref, const ygzf_sia_frame *cur, const ygzf_camera *cam, const float *inv_scale_factors,
                 int max_level, int min_level, int n_iter, float *TCR_out, size_t *ret, float *info, float *H36) {
    return sia_run_impl(c, ref, cur, -1, -1, cam, inv_scale_factors, max_level, min_level, n_iter, TCR_out, ret, info, H36);
}

int ygzf_sia_run_cached(ygzf_ctx *c, int ref_slot, int cur_slot, const ygzf_sia_frame *ref, const float *cur_Tcw7, const ygzf_camera *cam,
                        const float *inv_scale_factors, int max_level, int min_level, int n_iter, float *TCR_out, size_t *ret, float *info,
                        float *H36) {
    if (!cur_Tcw7) return fail(c, YGZF_ERR_INVALID, "null argument");
    if (ref_slot < 0) return fail(c, YGZF_ERR_INVALID, "slot %d", ref_slot);
    ygzf_sia_frame cur;
    memset(&cur, 0, sizeof cur);
    memcpy(cur.Tcw, cur_Tcw7, 28);
    return sia_run_impl(c, ref, &cur, ref_slot, cur_slot, cam, inv_scale_factors, max_level, min_level, n_iter, TCR_out, ret, info, H36);
}

// ---- Thirdparty/fast replacement ------------------------------------------------------------------------------------------
int ygzf_fast10(ygzf_ctx *c, const uint8_t *img, int img_w, int img_h, int stride, int x0, int y0, int w, int h, int barrier, int16_t *xy,
                int *scores, int *nonmax_idx, int cap, int *n_corners, int *n_nonmax) {
    if (!c || !img || !n_corners) return fail(c, YGZF_ERR_INVALID, "null argument");
    *n_corners = 0;
    if (n_nonmax) *n_nonmax = 0;
    if (img_w < 1 || img_h < 1 || stride < img_w || w < 1 || h < 1 || x0 < 0 || y0 < 0 || x0 + w > img_w || y0 + h > img_h || cap < 0)
        return fail(c, YGZF_ERR_INVALID, "bad image / window geometry");
    // detection domain of fast_corner_detect_10_sse2 (faster_corner_10_sse.cpp:188-198)
    int dx0 = 3, dx1 = w - 3, dy0 = 3, dy1 = h - 3;
    if (w < 22) {          // falls back to the plain detector, which scans the whole window and reads 3 px around it
        dx0 = 0; dx1 = w; dy0 = 0; dy1 = h;
        if (x0 < 3 || y0 < 3 || x0 + w + 3 > img_w || y0 + h + 3 > img_h)
            return fail(c, YGZF_ERR_INVALID, "a window narrower than 22 px needs a 3-px margin inside the image (the reference reads it)");
    } else if (h < 7)
        return YGZF_OK;
    HIPCHECK(c, hipSetDevice(c->device));
    ygzf_ctx::Buf *B = c->dF10;
    const int pitch = align_up(img_w, 64);
    int rc;
    if ((rc = ensure(c, B[0], (size_t) pitch * img_h)) || (rc = ensure(c, B[1], (size_t) w * h * sizeof(short))) ||
        (rc = ensure(c, B[2], (size_t) (2 * h + 2) * sizeof(int))) || (rc = ensure(c, B[3], (size_t) std::max(cap, 1) * 2 * sizeof(short))) ||
        (rc = ensure(c, B[4], (size_t) std::max(cap, 1) * sizeof(int))) || (rc = ensure(c, B[5], (size_t) std::max(cap, 1) * sizeof(int))))
        return rc;
    if ((rc = upload_rows(c, B[0].p, (size_t) pitch, img, (size_t) stride, img_w, (size_t) img_h))) return rc;
    int *rowCnt = (int *) B[2].p, *rowKept = rowCnt + h, *totals = rowKept + h;
    {
        ProfScope ps(c, KK_FAST10);
        launch_fast10(c->stream, (const uint8_t *) B[0].p, pitch, x0, y0, w, h, dx0, dx1, dy0, dy1, barrier, (short *) B[1].p, rowCnt, rowKept,
                      totals, (short *) B[3].p, (int *) B[4].p, (int *) B[5].p, cap);
    }
    HIPCHECK(c, hipGetLastError());
    int tot[2];
    HIPCHECK(c, hipMemcpyAsync(tot, totals, sizeof tot, hipMemcpyDeviceToHost, c->stream));
    HIPCHECK(c, hipStreamSynchronize(c->stream));
    *n_corners = tot[0];
    if (n_nonmax) *n_nonmax = tot[1];
    if (tot[0] > cap) return fail(c, YGZF_ERR_INVALID, "capacity %d < %d corners", cap, tot[0]);
    if (tot[0] && xy) HIPCHECK(c, hipMemcpyAsync(xy, B[3].p, (size_t) tot[0] * 2 * sizeof(short), hipMemcpyDeviceToHost, c->stream));
    if (tot[0] && scores) HIPCHECK(c, hipMemcpyAsync(scores, B[4].p, (size_t) tot[0] * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    if (tot[1] && nonmax_idx) HIPCHECK(c, hipMemcpyAsync(nonmax_idx, B[5].p, (size_t) tot[1] * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIPCHECK(c, hipStreamSynchronize(c->stream));
    return YGZF_OK;
}

// Level position of an existing key as the reference rounds it: tmp.pt *= mvInvScaleFactor[octave] (float), then cvRound
static int key_level_position(ygzf_ctx *c, const ygzf_kp &k, int i, int *px, int *py) {
    const int L = c->tab.cfg.nlevels;
    if (k.octave < 0 || k.octave >= L) return fail(c, YGZF_ERR_INVALID, "key %d: octave %d out of range", i, k.octave);
    const float inv = c->tab.invScale[k.octave];
    const float lx = k.x * inv, ly = k.y * inv;
    *px = cv_round_host((double) lx);
    *py = cv_round_host((double) ly);
    const LevelGeom &g = c->geo.lv[k.octave];
    // IC_Angle and the rotated pattern read the 31x31 patch around the rounded position: outside the level that is an out-of-bounds
    // read in the reference (its levels have no border here)
    if (*px < kHalfPatch || *py < kHalfPatch || *px >= g.w - kHalfPatch || *py >= g.h - kHalfPatch)
        return fail(c, YGZF_ERR_INVALID, "key %d is closer than %d px to the border of level %d", i, kHalfPatch, k.octave);
    return YGZF_OK;
}

int ygzf_describe_keys(ygzf_ctx *c, int frame, const ygzf_kp *keys, int n, int recompute_angle, float *angles_out, uint8_t *desc) {
    if (!c || (n > 0 && (!keys || !desc))) return fail(c, YGZF_ERR_INVALID, "null argument");
    if (c->lastFrames < 1) return fail(c, YGZF_ERR_STATE, "no extracted batch");
    if (frame < 0 || frame >= c->lastFrames) return fail(c, YGZF_ERR_INVALID, "frame %d out of range", frame);
    if (n <= 0) return YGZF_OK;
    HIPCHECK(c, hipSetDevice(c->device));
    std::vector<int> list4((size_t) 4 * n);
    for (int i = 0; i < n; i++) {
        int px, py;
        int rc = key_level_position(c, keys[i], i, &px, &py);
        if (rc) return rc;
        list4[4 * i] = px;
        list4[4 * i + 1] = py;
        list4[4 * i + 2] = keys[i].octave | (recompute_angle ? 0 : 0x100);
        memcpy(&list4[4 * i + 3], &keys[i].angle, 4);
    }
    int rc;
    if ((rc = ensure(c, c->dDso[5], 16 * (size_t) n)) || (rc = ensure(c, c->dDso[7], 4 * (size_t) n)) || (rc = ensure(c, c->dTmpC, 32 * (size_t) n))) return rc;
    HIPCHECK(c, hipMemcpyAsync(c->dDso[5].p, list4.data(), 16 * (size_t) n, hipMemcpyHostToDevice, c->stream));
    {
        ProfScope ps(c, KK_DESCRIBE);
        launch_describe_list(c->stream, c->lastFs, (const LevelGeom *) c->dGeom.p, c->dDso[5].p, n, frame, (float *) c->dDso[7].p, (uint8_t *) c->dTmpC.p, c->tab.cfg.cv_mode);
    }
    HIPCHECK(c, hipGetLastError());
    if (angles_out) HIPCHECK(c, hipMemcpyAsync(angles_out, c->dDso[7].p, 4 * (size_t) n, hipMemcpyDeviceToHost, c->stream));
    HIPCHECK(c, hipMemcpyAsync(desc, c->dTmpC.p, 32 * (size_t) n, hipMemcpyDeviceToHost, c->stream));
    HIPCHECK(c, hipStreamSynchronize(c->stream));
    return YGZF_OK;
}

int ygzf_extract_dso(ygzf_ctx *c, const uint8_t *img, int w, int h, int stride, ygzf_kp *keys, int n_existing, int cap, uint8_t *desc,
                     int *grid_size, int *n_total) {
    if (!c || !img || !grid_size || !n_total || (n_existing > 0 && !keys)) return fail(c, YGZF_ERR_INVALID, "null argument");
    if (n_existing < 0 || n_existing > cap) return fail(c, YGZF_ERR_INVALID, "n_existing %d outside 0..cap", n_existing);
    if (w > 65535 || h > 65535) return fail(c, YGZF_ERR_UNSUPPORTED, "image larger than 65535 px");
    *n_total = n_existing;
    HIPCHECK(c, hipSetDevice(c->device));
    int rc = apply_geometry(c, w, h, 1);
    if (rc) return rc;
    FrameSet fs;
    if ((rc = upload_frames(c, img, 1, w, h, stride, 0, &fs))) return rc;
    const Geometry &G = c->geo;
    const int L = c->tab.cfg.nlevels, n = c->tab.cfg.nfeatures;
    const LevelGeom *dGeom = (const LevelGeom *) c->dGeom.p;
    for (int l = 1; l < L; l++) {   // Frame ctor: ComputeImagePyramid (src/Frame.cc:807-813)
        ProfScope ps(c, KK_PYR);
        launch_pyr_resize(c->stream, fs, dGeom, G.lv[l], l, 1, (const int *) c->dXofs.p, (const short *) c->dXalpha.p, (const int *) c->dYofs.p,
                          (const short *) c->dYbeta.p);
    }
    c->lastFrames = 0;
    c->carryValid = false;
    // existing keys: occupancy at cvRound(pt) on level 0 (:1286-1291), describe position cvRound(pt * invScale[octave]) (:1104-1112, :1380-1383)
    std::vector<unsigned> occXY(n_existing);
    std::vector<int> list4((size_t) 4 * n_existing);
    for (int i = 0; i < n_existing; i++) {
        const ygzf_kp &k = keys[i];
        int px, py;
        if ((rc = key_level_position(c, k, i, &px, &py))) return rc;
        const int ox = cv_round_host((double) k.x), oy = cv_round_host((double) k.y);
        if (ox < 0 || oy < 0 || ox >= w || oy >= h) return fail(c, YGZF_ERR_INVALID, "existing key %d lies outside the image", i);
        occXY[i] = (unsigned) ox | ((unsigned) oy << 16);
        list4[4 * i] = px; list4[4 * i + 1] = py; list4[4 * i + 2] = k.octave; list4[4 * i + 3] = 0;
    }
    int grid = *grid_size;
    if (grid < 0) grid = (int) std::sqrt(1.0 * h * w / (n > 0 ? n : 1));
    const int minGrid = 7;
    if (grid < 1) return fail(c, YGZF_ERR_INVALID, "grid size %d", grid);
    const int gmin = std::min(grid, minGrid);   // a start below 7 (many features on a small image) is used as it is; it only never shrinks further
    const int maxCells = (w / gmin) * (h / gmin) + 1;
    const size_t occWords = ((size_t) w * h + 31) / 32;
    ygzf_ctx::Buf &dOcc = c->dDso[0], &dOccXY = c->dDso[1], &dCellCnt = c->dDso[2], &dCellXY = c->dDso[3], &dTotal = c->dDso[4], &dList = c->dDso[5],
                  &dNewXY = c->dDso[6], &dAng = c->dDso[7];
    const size_t maxEntries = (size_t) n_existing + 3 * (size_t) maxCells;
    if ((rc = ensure(c, dOcc, occWords * 4)) || (rc = ensure(c, dOccXY, 4 * (size_t) (n_existing + 1))) || (rc = ensure(c, dCellCnt, 4 * (size_t) maxCells)) ||
        (rc = ensure(c, dCellXY, 12 * (size_t) maxCells)) || (rc = ensure(c, dTotal, 64)) || (rc = ensure(c, dList, 16 * maxEntries)) ||
        (rc = ensure(c, dNewXY, 4 * 3 * (size_t) maxCells)) || (rc = ensure(c, dAng, 4 * maxEntries)) || (rc = ensure(c, c->dTmpC, 32 * maxEntries)))
        return rc;
    HIPCHECK(c, hipMemsetAsync(dOcc.p, 0, occWords * 4, c->stream));
    if (n_existing > 0) {
        HIPCHECK(c, hipMemcpyAsync(dOccXY.p, occXY.data(), 4 * (size_t) n_existing, hipMemcpyHostToDevice, c->stream));
        HIPCHECK(c, hipMemcpyAsync(dList.p, list4.data(), 16 * (size_t) n_existing, hipMemcpyHostToDevice, c->stream));
        launch_dso_occ(c->stream, (const unsigned *) dOccXY.p, n_existing, w, h, (unsigned *) dOcc.p);
    }
    // the grid-size retry loop of :1301-1377; mnGridSize persists across frames through *grid_size
    int cnt = 0, nInner = 0;
    while (cnt < n) {
        if (cnt > 0) {
            grid -= 5;
            if (grid < minGrid) {
                grid = minGrid;
                break;   // the keypoints of the previous pass stand
            }
        }
        if (grid > kDsoMaxGrid) return fail(c, YGZF_ERR_UNSUPPORTED, "mnGridSize %d > %d (nfeatures too small for this image size)", grid, kDsoMaxGrid);
        if (grid < 1) return fail(c, YGZF_ERR_INVALID, "grid size %d", grid);
        const int nRows = h / grid, nCols = w / grid;
        nInner = (nRows > 2 && nCols > 2) ? (nRows - 2) * (nCols - 2) : 0;
        if (nInner > maxCells) return fail(c, YGZF_ERR_INVALID, "grid size %d: more cells than planned", grid);
        HIPCHECK(c, hipMemsetAsync(dTotal.p, 0, 4, c->stream));
        {
            ProfScope ps(c, KK_DSO);
            launch_dso_cells(c->stream, fs.img0, fs.img0_pitch, w, h, grid, nCols, nRows, (unsigned *) dOcc.p, (int *) dCellCnt.p, (unsigned *) dCellXY.p,
                             (int *) dTotal.p, 20, 5, 3, w, false);   // :1330 / :1337: both barriers hard-coded in the single-level detector
        }
        cnt = 0;
        if (nInner > 0) {
            HIPCHECK(c, hipMemcpyAsync(&cnt, dTotal.p, 4, hipMemcpyDeviceToHost, c->stream));
            HIPCHECK(c, hipStreamSynchronize(c->stream));
        }
        if (cnt == 0) break;   // the reference loops forever on a frame without a single corner; defined: no new keypoints
    }
    if (cnt > n) grid += 5;
    *grid_size = grid;
    const int total = n_existing + cnt;
    *n_total = total;
    if (total > cap) return fail(c, YGZF_ERR_INVALID, "capacity %d < %d keypoints", cap, total);
    if (total == 0) return YGZF_OK;
    if (!desc || !keys) return fail(c, YGZF_ERR_INVALID, "null output");
    if (cnt > 0) launch_dso_compact(c->stream, (const int *) dCellCnt.p, (const unsigned *) dCellXY.p, nInner, n_existing, dList.p, (unsigned *) dNewXY.p, 0);
    {
        ProfScope ps(c, KK_DESCRIBE);
        launch_describe_list(c->stream, fs, dGeom, dList.p, total, 0, (float *) dAng.p, (uint8_t *) c->dTmpC.p, c->tab.cfg.cv_mode);
    }
    HIPCHECK(c, hipGetLastError());
    std::vector<float> ang(total);
    std::vector<unsigned> nxy(cnt);
    HIPCHECK(c, hipMemcpyAsync(ang.data(), dAng.p, 4 * (size_t) total, hipMemcpyDeviceToHost, c->stream));
    if (cnt > 0) HIPCHECK(c, hipMemcpyAsync(nxy.data(), dNewXY.p, 4 * (size_t) cnt, hipMemcpyDeviceToHost, c->stream));
    HIPCHECK(c, hipMemcpyAsync(desc, c->dTmpC.p, 32 * (size_t) total, hipMemcpyDeviceToHost, c->stream));
    HIPCHECK(c, hipStreamSynchronize(c->stream));
    for (int i = 0; i < n_existing; i++) keys[i].angle = ang[i];
    for (int i = 0; i < cnt; i++) {   // :1356-1366
        ygzf_kp &k = keys[n_existing + i];
        k.x = (float) (nxy[i] & 0xFFFFu);
        k.y = (float) (nxy[i] >> 16);
        k.size = 7.f;
        k.angle = ang[n_existing + i];
        k.response = 0.f;
        k.octave = 0;
        k.class_id = -1;
    }
    return YGZF_OK;
}

// Shared by the two grid detectors below: image up, pyramid, the describe-list entries of the frame's own keys (re-oriented by both
// detectors: :1268-1271, :1503-1505).  list4 entries: (x, y at level coordinates, octave, unused).
static int grid_extract_begin(ygzf_ctx *c, const uint8_t *img, int w, int h, int stride, const ygzf_kp *keys, int n_existing, FrameSet *fs,
                              std::vector<int> *list4) {
    if (w > 65535 || h > 65535) return fail(c, YGZF_ERR_UNSUPPORTED, "image larger than 65535 px");
    HIPCHECK(c, hipSetDevice(c->device));
    int rc = apply_geometry(c, w, h, 1);
    if (rc) return rc;
    if ((rc = upload_frames(c, img, 1, w, h, stride, 0, fs))) return rc;
    const Geometry &G = c->geo;
    const int L = c->tab.cfg.nlevels;
    const LevelGeom *dGeom = (const LevelGeom *) c->dGeom.p;
    for (int l = 1; l < L; l++) {   // Frame ctor: ComputeImagePyramid (src/Frame.cc:807-813)
        ProfScope ps(c, KK_PYR);
        launch_pyr_resize(c->stream, *fs, dGeom, G.lv[l], l, 1, (const int *) c->dXofs.p, (const short *) c->dXalpha.p, (const int *) c->dYofs.p,
                          (const short *) c->dYbeta.p);
    }
    c->lastFrames = 0;
    c->carryValid = false;
    list4->assign((size_t) 4 * n_existing, 0);
    for (int i = 0; i < n_existing; i++) {
        int px, py;
        if ((rc = key_level_position(c, keys[i], i, &px, &py))) return rc;
        (*list4)[4 * i] = px; (*list4)[4 * i + 1] = py; (*list4)[4 * i + 2] = keys[i].octave; (*list4)[4 * i + 3] = 0;
    }
    return YGZF_OK;
}

// describe `total` list entries (the first n_existing are the frame's own keys) and read angles + descriptors back
static int grid_extract_describe(ygzf_ctx *c, const FrameSet &fs, void *dList, int total, std::vector<float> *ang, uint8_t *desc) {
    int rc;
    if ((rc = ensure(c, c->dDso[7], 4 * (size_t) total)) || (rc = ensure(c, c->dTmpC, 32 * (size_t) total))) return rc;
    {
        ProfScope ps(c, KK_DESCRIBE);
        launch_describe_list(c->stream, fs, (const LevelGeom *) c->dGeom.p, dList, total, 0, (float *) c->dDso[7].p, (uint8_t *) c->dTmpC.p, c->tab.cfg.cv_mode);
    }
    HIPCHECK(c, hipGetLastError());
    ang->resize(total);
    HIPCHECK(c, hipMemcpyAsync(ang->data(), c->dDso[7].p, 4 * (size_t) total, hipMemcpyDeviceToHost, c->stream));
    HIPCHECK(c, hipMemcpyAsync(desc, c->dTmpC.p, 32 * (size_t) total, hipMemcpyDeviceToHost, c->stream));
    HIPCHECK(c, hipStreamSynchronize(c->stream));
    return YGZF_OK;
}

int ygzf_extract_fast_keypoint(ygzf_ctx *c, const uint8_t *img, int w, int h, int stride, ygzf_kp *keys, int n_existing, int cap, uint8_t *desc, int *n_total) {
    if (!c || !img || !n_total || (n_existing > 0 && !keys)) return fail(c, YGZF_ERR_INVALID, "null argument");
    if (n_existing < 0 || n_existing > cap) return fail(c, YGZF_ERR_INVALID, "n_existing %d outside 0..cap", n_existing);
    *n_total = n_existing;
    FrameSet fs;
    std::vector<int> list4;
    int rc = grid_extract_begin(c, img, w, h, stride, keys, n_existing, &fs, &list4);
    if (rc) return rc;
    const Geometry &G = c->geo;
    const int L = c->tab.cfg.nlevels;
    // the 5-px occupancy grid of the frame's own keys (:1194-1204), built on the host in the reference's float arithmetic
    const int gridRows = h / 5, gridCols = w / 5;
    const long long nCells = (long long) gridRows * gridCols;
    if (nCells < 1) return fail(c, YGZF_ERR_UNSUPPORTED, "image smaller than one 5-px cell");
    std::vector<uint8_t> occ((size_t) nCells, 0);
    for (int i = 0; i < n_existing; i++) {
        const int gy = (int) (keys[i].y / 5), gx = (int) (keys[i].x / 5);
        const long long k = (long long) gy * gridCols + gx;
        if (k >= 0 && k < nCells) occ[(size_t) k] = 1;
    }
    // per-level corner lists (kept until the winners are known): xy / scores / nonmax at lvlOff[l], capacity = the level's detection window
    std::vector<long long> lvlOff(L + 1, 0);
    size_t maxWin = 1;
    int maxH = 1;
    for (int l = 0; l < L; l++) {
        const LevelGeom &g = G.lv[l];
        const long long win = (g.w >= 42 && g.h >= 27) ? (long long) (g.w - 20) * (g.h - 20) : 0;   // narrower / lower levels are skipped (defined; include/ygzf.h)
        // the vote key carries level << 24 | corner index (dso_kernels.hip, k_fgrid_vote): a level's corners are at most its window's pixels
        if (win >= (1ll << 24)) return fail(c, YGZF_ERR_UNSUPPORTED, "FAST_KEYPOINT: level %d has %lld candidate positions (at most 2^24 - 1 per level)", l, win);
        lvlOff[l + 1] = lvlOff[l] + win;
        maxWin = std::max(maxWin, (size_t) win);
        maxH = std::max(maxH, g.h);
    }
    const size_t nAll = (size_t) std::max<long long>(lvlOff[L], 1);
    ygzf_ctx::Buf *B = c->dF10;
    ygzf_ctx::Buf &dOcc = c->dDso[0], &dKey = c->dDso[1], &dCellXY = c->dDso[3], &dOff = c->dDso[4], &dList = c->dDso[5], &dTot = c->dDso[2];
    if ((rc = ensure(c, B[1], maxWin * sizeof(short))) || (rc = ensure(c, B[2], (size_t) (2 * maxH + 2) * sizeof(int))) || (rc = ensure(c, B[3], nAll * 2 * sizeof(short))) ||
        (rc = ensure(c, B[4], nAll * sizeof(int))) || (rc = ensure(c, B[5], nAll * sizeof(int))) || (rc = ensure(c, dOcc, (size_t) nCells)) ||
        (rc = ensure(c, dKey, 8 * (size_t) nCells)) || (rc = ensure(c, dCellXY, 4 * (size_t) nCells)) || (rc = ensure(c, dOff, 8 * (size_t) (L + 1))) ||
        (rc = ensure(c, dTot, 8 * (size_t) L + 8)))
        return rc;
    HIPCHECK(c, hipMemcpyAsync(dOcc.p, occ.data(), (size_t) nCells, hipMemcpyHostToDevice, c->stream));
    HIPCHECK(c, hipMemcpyAsync(dOff.p, lvlOff.data(), 8 * (size_t) (L + 1), hipMemcpyHostToDevice, c->stream));
    HIPCHECK(c, hipMemsetAsync(dKey.p, 0, 8 * (size_t) nCells, c->stream));
    HIPCHECK(c, hipMemsetAsync(dTot.p, 0, 8 * (size_t) L + 8, c->stream));
    for (int l = 0; l < L; l++) {
        const LevelGeom &g = G.lv[l];
        const long long win = lvlOff[l + 1] - lvlOff[l];
        if (win <= 0) continue;
        int pitch;
        const uint8_t *lp = level_ptr(fs, g, l, 0, &pitch);
        const int ww = g.w - 20, wh = g.h - 20;
        int *rowCnt = (int *) B[2].p, *rowKept = rowCnt + wh, *totals = (int *) dTot.p + 2 * l;
        short *xy = (short *) B[3].p + 2 * lvlOff[l];
        int *scores = (int *) B[4].p + lvlOff[l], *nm = (int *) B[5].p + lvlOff[l];
        {
            ProfScope ps(c, KK_FAST10);
            // fast_corner_detect_10_sse2 on the window that starts 20 px in (:1216-1226): domain [3, ww - 3) x [3, wh - 3), barrier iniThFAST; score and
            // >= non-maximum suppression as fast_corner_score_10 / fast_nonmax_3x3 (:1233-1236)
            launch_fast10(c->stream, lp, pitch, 20, 20, ww, wh, 3, ww - 3, 3, wh - 3, c->tab.cfg.ini_th_fast, (short *) B[1].p, rowCnt, rowKept, totals, xy, scores,
                          nm, (int) std::min<long long>(win, 0x7fffffff));
        }
        launch_fgrid_vote(c->stream, lp, pitch, g.w, g.h, l, c->tab.scale[l], xy, nm, totals, (int) std::min<long long>(win, 0x7fffffff), gridCols, nCells,
                          (const uint8_t *) dOcc.p, (unsigned long long *) dKey.p);
    }
    launch_fgrid_gather(c->stream, (const unsigned long long *) dKey.p, nCells, (const short *) B[3].p, (const int *) B[5].p, (const long long *) dOff.p,
                        (unsigned *) dCellXY.p);
    HIPCHECK(c, hipGetLastError());
    std::vector<unsigned long long> key((size_t) nCells);
    std::vector<unsigned> cxy((size_t) nCells);
    HIPCHECK(c, hipMemcpyAsync(key.data(), dKey.p, 8 * (size_t) nCells, hipMemcpyDeviceToHost, c->stream));
    HIPCHECK(c, hipMemcpyAsync(cxy.data(), dCellXY.p, 4 * (size_t) nCells, hipMemcpyDeviceToHost, c->stream));
    HIPCHECK(c, hipStreamSynchronize(c->stream));
    // allKeypoints[octave]: the cells in index order (:1260-1266), then level after level (:1104-1125)
    std::vector<std::vector<ygzf_kp>> all(L);
    for (long long k = 0; k < nCells; k++) {
        if (!key[(size_t) k]) continue;
        const unsigned order = 0xFFFFFFFFu - (unsigned) key[(size_t) k];
        const int level = (int) (order >> 24);
        ygzf_kp kp;
        kp.x = (float) (cxy[(size_t) k] & 0xFFFFu);
        kp.y = (float) (cxy[(size_t) k] >> 16);
        kp.size = (float) (int) (31 * c->tab.scale[level]);
        kp.angle = -1.f;
        const unsigned sb = (unsigned) (key[(size_t) k] >> 32);
        memcpy(&kp.response, &sb, 4);
        kp.octave = level;
        kp.class_id = -1;
        all[level].push_back(kp);
    }
    int total = n_existing;
    for (auto &v : all) total += (int) v.size();
    *n_total = total;
    if (total > cap) return fail(c, YGZF_ERR_INVALID, "capacity %d < %d keypoints", cap, total);
    if (total == 0) return YGZF_OK;
    if (!desc || !keys) return fail(c, YGZF_ERR_INVALID, "null output");
    int at = n_existing;
    for (int l = 0; l < L; l++)
        for (const ygzf_kp &kp : all[l]) {
            list4.push_back((int) kp.x); list4.push_back((int) kp.y); list4.push_back(l); list4.push_back(0);
            keys[at++] = kp;
        }
    if ((rc = ensure(c, dList, 16 * (size_t) total))) return rc;
    HIPCHECK(c, hipMemcpyAsync(dList.p, list4.data(), 16 * (size_t) total, hipMemcpyHostToDevice, c->stream));
    std::vector<float> ang;
    if ((rc = grid_extract_describe(c, fs, dList.p, total, &ang, desc))) return rc;
    for (int i = 0; i < total; i++) keys[i].angle = ang[i];
    for (int i = n_existing; i < total; i++)            // keypoint->pt *= scale for levels > 0 (:1116-1121)
        if (keys[i].octave != 0) { keys[i].x *= c->tab.scale[keys[i].octave]; keys[i].y *= c->tab.scale[keys[i].octave]; }
    return YGZF_OK;
}

int ygzf_extract_dso_multilevel(ygzf_ctx *c, const uint8_t *img, int w, int h, int stride, ygzf_kp *keys, int n_existing, int cap, uint8_t *desc,
                                int *grid_size, int *n_total) {
    if (!c || !img || !grid_size || !n_total || (n_existing > 0 && !keys)) return fail(c, YGZF_ERR_INVALID, "null argument");
    if (n_existing < 0 || n_existing > cap) return fail(c, YGZF_ERR_INVALID, "n_existing %d outside 0..cap", n_existing);
    *n_total = n_existing;
    FrameSet fs;
    std::vector<int> list4;
    int rc = grid_extract_begin(c, img, w, h, stride, keys, n_existing, &fs, &list4);
    if (rc) return rc;
    const Geometry &G = c->geo;
    const int L = c->tab.cfg.nlevels;
    std::vector<unsigned> occXY(n_existing);
    for (int i = 0; i < n_existing; i++) {
        const int ox = cv_round_host((double) keys[i].x), oy = cv_round_host((double) keys[i].y);
        if (ox < 0 || oy < 0 || ox >= w || oy >= h) return fail(c, YGZF_ERR_INVALID, "existing key %d lies outside the image", i);
        occXY[i] = (unsigned) ox | ((unsigned) oy << 16);
    }
    // capacity: a level can end with up to 2 keys per inner cell of its finest grid (7 px)
    long long maxNew = 0, maxCells = 1;
    for (int l = 0; l < L; l++) {
        const long long cells = (long long) (G.lv[l].w / 7) * (G.lv[l].h / 7) + 1;
        maxNew += 2 * cells;
        maxCells = std::max(maxCells, cells);
    }
    const size_t occWords = ((size_t) w * h + 31) / 32;
    ygzf_ctx::Buf &dOcc = c->dDso[0], &dOccXY = c->dDso[1], &dCellCnt = c->dDso[2], &dCellXY = c->dDso[3], &dTotal = c->dDso[4], &dList = c->dDso[5],
                  &dNewXY = c->dDso[6];
    const size_t maxEntries = (size_t) n_existing + (size_t) maxNew;
    if ((rc = ensure(c, dOcc, occWords * 4)) || (rc = ensure(c, dOccXY, 4 * (size_t) (n_existing + 1))) || (rc = ensure(c, dCellCnt, 4 * (size_t) maxCells)) ||
        (rc = ensure(c, dCellXY, 12 * (size_t) maxCells)) || (rc = ensure(c, dTotal, 64)) || (rc = ensure(c, dList, 16 * maxEntries)) ||
        (rc = ensure(c, dNewXY, 4 * (size_t) maxNew + 16)))
        return rc;
    HIPCHECK(c, hipMemsetAsync(dOcc.p, 0, occWords * 4, c->stream));
    if (n_existing > 0) {
        HIPCHECK(c, hipMemcpyAsync(dOccXY.p, occXY.data(), 4 * (size_t) n_existing, hipMemcpyHostToDevice, c->stream));
        HIPCHECK(c, hipMemcpyAsync(dList.p, list4.data(), 16 * (size_t) n_existing, hipMemcpyHostToDevice, c->stream));
        launch_dso_occ(c->stream, (const unsigned *) dOccXY.p, n_existing, w, h, (unsigned *) dOcc.p);
    }
    int grid = *grid_size, newTotal = 0;
    std::vector<int> lvlCount(L, 0);
    for (int l = 0; l < L; l++) {
        const LevelGeom &g = G.lv[l];
        const int n = c->tab.nFeat[l];
        if (n <= 0) continue;
        int pitch;
        const uint8_t *lp = level_ptr(fs, g, l, 0, &pitch);
        grid = (int) std::sqrt(1.0 * g.h * g.w / n);     // recomputed per level (:1407)
        if (grid < 1) grid = 1;
        int cnt = 0, nInner = 0;
        while (cnt < n) {                                // :1412-1494
            if (cnt > 0) {
                grid -= 5;
                if (grid < 7) { grid = 7; break; }       // the keypoints of the previous pass stand
            }
            if (grid > kDsoMaxGrid) return fail(c, YGZF_ERR_UNSUPPORTED, "mnGridSize %d > %d at level %d (too few features for this image size)", grid, kDsoMaxGrid, l);
            const int nRows = g.h / grid, nCols = g.w / grid;
            nInner = (nRows > 2 && nCols > 2) ? (nRows - 2) * (nCols - 2) : 0;
            if (nInner > maxCells) return fail(c, YGZF_ERR_INVALID, "grid size %d: more cells than planned", grid);
            HIPCHECK(c, hipMemsetAsync(dTotal.p, 0, 4, c->stream));
            {
                ProfScope ps(c, KK_DSO);
                launch_dso_cells(c->stream, lp, pitch, g.w, g.h, grid, nCols, nRows, (unsigned *) dOcc.p, (int *) dCellCnt.p, (unsigned *) dCellXY.p, (int *) dTotal.p,
                                 c->tab.cfg.ini_th_fast, c->tab.cfg.min_th_fast, 2, w, true);
            }
            cnt = 0;
            if (nInner > 0) {
                HIPCHECK(c, hipMemcpyAsync(&cnt, dTotal.p, 4, hipMemcpyDeviceToHost, c->stream));
                HIPCHECK(c, hipStreamSynchronize(c->stream));
            }
            if (cnt == 0) break;                         // defined: a pass without a single corner ends the level (the reference would spin)
        }
        if (cnt > 0) {
            if ((long long) newTotal + cnt > maxNew) return fail(c, YGZF_ERR_INVALID, "more keypoints than planned");
            launch_dso_compact(c->stream, (const int *) dCellCnt.p, (const unsigned *) dCellXY.p, nInner, n_existing + newTotal, dList.p,
                               (unsigned *) dNewXY.p + newTotal, l);
            lvlCount[l] = cnt;
            newTotal += cnt;
        }
    }
    *grid_size = grid;
    const int total = n_existing + newTotal;
    *n_total = total;
    if (total > cap) return fail(c, YGZF_ERR_INVALID, "capacity %d < %d keypoints", cap, total);
    if (total == 0) return YGZF_OK;
    if (!desc || !keys) return fail(c, YGZF_ERR_INVALID, "null output");
    std::vector<unsigned> nxy(std::max(newTotal, 1));
    if (newTotal > 0) HIPCHECK(c, hipMemcpyAsync(nxy.data(), dNewXY.p, 4 * (size_t) newTotal, hipMemcpyDeviceToHost, c->stream));
    std::vector<float> ang;
    if ((rc = grid_extract_describe(c, fs, dList.p, total, &ang, desc))) return rc;
    for (int i = 0; i < n_existing; i++) keys[i].angle = ang[i];
    int at = 0;
    for (int l = 0; l < L; l++)
        for (int j = 0; j < lvlCount[l]; j++, at++) {    // :1473-1484, then pt *= scale (:1116-1121)
            ygzf_kp &k = keys[n_existing + at];
            k.x = (float) (nxy[at] & 0xFFFFu);
            k.y = (float) (nxy[at] >> 16);
            if (l != 0) { k.x *= c->tab.scale[l]; k.y *= c->tab.scale[l]; }
            k.size = 7.f;
            k.angle = ang[n_existing + at];
            k.response = 0.f;
            k.octave = l;
            k.class_id = -1;
        }
    return YGZF_OK;
}

// MapPoint::PredictScale (src/MapPoint.cc:359-373) is a non-decreasing step function of ratio = mfMaxDistance / dist; its steps are
// tabulated here with the host's own libm so that the device reproduces it by comparisons: step[k] = smallest float ratio whose level
// is >= k (k = 1 .. nlevels-1).
static int predict_scale_host(float ratio, float logScaleFactor, int nScaleLevels) {
    int nScale = (int) std::ceil(std::log(ratio) / logScaleFactor);
    if (nScale < 0) nScale = 0;
    else if (nScale >= nScaleLevels) nScale = nScaleLevels - 1;
    return nScale;
}
static void predict_scale_steps(float logScaleFactor, int nScaleLevels, float *step) {
    for (int k = 0; k < kMaxLevels; k++) step[k] = std::numeric_limits<float>::infinity();
    for (int k = 1; k < nScaleLevels && k < kMaxLevels; k++) {
        uint32_t lo = 0x00800000u, hi = 0x7F7FFFFFu;   // positive normal floats, ordered like their bit patterns
        auto lvl = [&](uint32_t b) { float r; memcpy(&r, &b, 4); return predict_scale_host(r, logScaleFactor, nScaleLevels); };
        if (lvl(hi) < k) continue;
        while (lo < hi) {
            const uint32_t mid = lo + (hi - lo) / 2;
            if (lvl(mid) >= k) hi = mid; else lo = mid + 1;
        }
        memcpy(&step[k], &lo, 4);
    }
}

struct FrustumHost {   // host-side inputs of the fused isInFrustum stage (ygzf_frustum_in without the ABI wrapper)
    const ygzf_frustum_in *in;
    uint8_t *in_view;
    float *proj_x, *proj_y, *proj_xr, *view_cos;
    int *level;
};

// The frustum inputs join the caller's packed upload (frustum_add_inputs before PackedTransfer::upload, frustum_fill_args after it).
struct FrustumOffsets { size_t world, normal, maxInv, minInv, mfMax, cand; };

static int frustum_add_inputs(ygzf_ctx *c, PackedTransfer &P, FrustumOffsets &O, const ygzf_frustum_in *in, int n, int nlevels) {
    if (!in->world || !in->normal || !in->max_dist_inv || !in->min_dist_inv || !in->mf_max_distance) return fail(c, YGZF_ERR_INVALID, "null frustum array");
    if (nlevels < 1 || nlevels > kMaxLevels) return fail(c, YGZF_ERR_INVALID, "nlevels out of range");
    O.world = P.add_in(in->world, 12 * (size_t) n);
    O.normal = P.add_in(in->normal, 12 * (size_t) n);
    O.maxInv = P.add_in(in->max_dist_inv, 4 * (size_t) n);
    O.minInv = P.add_in(in->min_dist_inv, 4 * (size_t) n);
    O.mfMax = P.add_in(in->mf_max_distance, 4 * (size_t) n);
    O.cand = P.add_in(in->candidate, in->candidate ? (size_t) n : 0);
    return YGZF_OK;
}

static void frustum_fill_args(FrustumArgs &A, const uint8_t *dIn, const FrustumOffsets &O, const ygzf_frustum_in *in, const ygzf_camera *cam, int n, int nlevels) {
    memset(&A, 0, sizeof A);
    A.n = n;
    A.candidate = in->candidate ? dIn + O.cand : nullptr;
    A.world = (const float *) (dIn + O.world);
    A.normal = (const float *) (dIn + O.normal);
    A.maxDistInv = (const float *) (dIn + O.maxInv);
    A.minDistInv = (const float *) (dIn + O.minInv);
    A.mfMaxDistance = (const float *) (dIn + O.mfMax);
    memcpy(A.Rcw, in->Rcw, 36);
    memcpy(A.tcw, in->tcw, 12);
    memcpy(A.Ow, in->Ow, 12);
    A.fx = cam->fx; A.fy = cam->fy; A.cx = cam->cx; A.cy = cam->cy; A.mbf = cam->mbf;
    A.minX = cam->min_x; A.minY = cam->min_y; A.maxX = cam->max_x; A.maxY = cam->max_y;
    A.viewingCosLimit = in->viewing_cos_limit;
    predict_scale_steps(in->log_scale_factor, nlevels, A.levelStep);
    A.nLevels = nlevels;
}

// shared body of the two searches whose queries arrive already projected (mode 1: F x local MapPoints, mode 2: Cur x KeyFrame points)
static int projected_match(ygzf_ctx *c, int mode, const ygzf_frame_view *F, const ygzf_camera *cam, int n_mp, const uint8_t *track_in_view,
                           const uint8_t *is_bad, const uint8_t *mp_has_obs, const float *proj_x, const float *proj_y, const float *proj_xr,
                           const float *view_cos, const int *scale_level, const float *mp_angle, const uint8_t *mp_desc, float th,
                           int check_level, float nnratio, int max_dist, int check_ori, uint8_t *owner, int *match, int *nmatches,
                           const ygzf_kp *last_keys = nullptr, int *match12 = nullptr, const FrustumHost *fr = nullptr) {
    if (!c || !F || !cam || !nmatches) return fail(c, YGZF_ERR_INVALID, "null argument");
    *nmatches = 0;
    if (F->n < 0 || n_mp < 0) return fail(c, YGZF_ERR_INVALID, "negative count");
    if (F->n == 0 || n_mp == 0) {
        for (int i = 0; i < F->n; i++) if (match) match[i] = -1;
        return YGZF_OK;
    }
    if (!F->keys || !F->desc || ((!proj_x || !proj_y) && !fr) || !mp_desc || !owner || !match) return fail(c, YGZF_ERR_INVALID, "null array");
    if (fr) {
        if (mode != 1) return fail(c, YGZF_ERR_INVALID, "fused frustum stage only feeds SearchByProjection(F, MapPoints)");
    } else if (mode != 3) {
        if (!track_in_view || (mode == 1 && !view_cos) || (mode == 2 && !mp_angle) || !scale_level) return fail(c, YGZF_ERR_INVALID, "null array");
        for (int i = 0; i < n_mp; i++)
            if (track_in_view[i] && (scale_level[i] < 0 || scale_level[i] >= kMaxLevels)) return fail(c, YGZF_ERR_INVALID, "scale level out of range");
    } else if (!last_keys || !match12) return fail(c, YGZF_ERR_INVALID, "null array");
    HIPCHECK(c, hipSetDevice(c->device));
    const size_t nt = F->n, nq = n_mp;
    int counts[2] = {F->n, n_mp};
    float pose[24] = {0};
    const int frLevels = F->nlevels > 0 ? F->nlevels : c->tab.cfg.nlevels;
    // one packed copy in, one out (PackedTransfer).  The per-MapPoint arrays the fused isInFrustum stage WRITES (projections, viewing cosine,
    // predicted level, in-view flag) live in the output half so that the caller's optional copies of them ride the same copy back.
    PackedTransfer P(c);
    const size_t oCurK = P.add_in(F->keys, nt * sizeof(ygzf_kp)), oCurD = P.add_in(F->desc, nt * 32), oUR = P.add_in(F->u_right, F->u_right ? nt * 4 : 0),
                 oOwn = P.add_in(owner, nt), oLastK = P.add_in(last_keys, last_keys ? nq * sizeof(ygzf_kp) : 0), oMpD = P.add_in(mp_desc, nq * 32),
                 oBad = P.add_in(is_bad, is_bad ? nq : 0), oObs = P.add_in(mp_has_obs, mp_has_obs ? nq : 0), oCnt = P.add_in(counts, sizeof counts),
                 oPose = P.add_in(pose, sizeof pose);
    size_t oPX = 0, oPY = 0, oPXR = 0, oVC = 0, oLv = 0, oTV = 0;
    size_t rPX = 0, rPY = 0, rPXR = 0, rVC = 0, rLv = 0, rTV = 0, rM12 = 0, rOwner = 0, rMatch = 0;
    FrustumOffsets FO;
    int rc;
    if (fr) {
        if ((rc = frustum_add_inputs(c, P, FO, fr->in, n_mp, frLevels))) return rc;
        rLv = P.add_out(fr->level, nq * 4);
        rTV = P.add_out(fr->in_view, nq);
        rPX = P.add_out(fr->proj_x, nq * 4);
        rPY = P.add_out(fr->proj_y, nq * 4);
        rPXR = P.add_out(fr->proj_xr, nq * 4);
        rVC = P.add_out(fr->view_cos, nq * 4);
    } else {
        oPX = P.add_in(proj_x, nq * 4);
        oPY = P.add_in(proj_y, nq * 4);
        oPXR = P.add_in(proj_xr, proj_xr ? nq * 4 : 0);
        if (mode != 3) {
            oTV = P.add_in(track_in_view, nq);
            oVC = P.add_in(mode == 2 ? mp_angle : view_cos, nq * 4);
            oLv = P.add_in(scale_level, nq * 4);
        }
    }
    if (mode == 3) rM12 = P.add_out(match12, nq * sizeof(int));
    rOwner = P.add_out(mode == 3 ? nullptr : owner, nt);            // mode 3 keeps them as kernel scratch (the caller derives them from match12)
    rMatch = P.add_out(mode == 3 ? nullptr : (void *) match, nt * sizeof(int));
    const size_t rN = P.add_out(nmatches, sizeof(int));
    uint8_t *dIn;
    if ((rc = P.upload(&dIn))) return rc;
    const float *dPX, *dY, *dXR, *dVC;
    const int *dLv;
    const uint8_t *dTV;
    if (fr) {   // Frame::isInFrustum on the device: its outputs land where the matcher reads them
        FrustumArgs FA;
        frustum_fill_args(FA, dIn, FO, fr->in, cam, n_mp, frLevels);
        FA.inView = P.d_out(rTV);
        FA.projX = (float *) P.d_out(rPX);
        FA.projY = (float *) P.d_out(rPY);
        FA.projXR = (float *) P.d_out(rPXR);
        FA.viewCos = (float *) P.d_out(rVC);
        FA.level = (int *) P.d_out(rLv);
        {
            ProfScope ps(c, KK_FRUSTUM);
            launch_frustum(c->stream, FA);
        }
        dPX = FA.projX; dY = FA.projY; dXR = FA.projXR; dVC = FA.viewCos; dLv = FA.level; dTV = FA.inView;
    } else {
        dPX = (const float *) (dIn + oPX);
        dY = (const float *) (dIn + oPY);
        dXR = proj_xr ? (const float *) (dIn + oPXR) : nullptr;
        dVC = (const float *) (dIn + oVC);     // unused in mode 3
        dLv = (const int *) (dIn + oLv);
        dTV = mode != 3 ? dIn + oTV : nullptr;
    }
    MatchArgs A;
    memset(&A, 0, sizeof A);
    A.maxDist = 100;   // TH_HIGH
    A.mode = mode;
    A.specDeep = mode == 1 ? 1 : 0;   // best AND runner-up among the free candidates: lists of eight (match_kernels.hip)
    A.maxDist = max_dist;
    A.curKeys = (const ygzf_kp *) (dIn + oCurK);
    A.curDesc = dIn + oCurD;
    A.curURight = F->u_right ? (const float *) (dIn + oUR) : nullptr;
    A.ownerIn = dIn + oOwn;
    A.curCnt = (const int *) (dIn + oCnt);
    A.kpStrideCur = (long long) nt;
    A.lastKeys = (const ygzf_kp *) (dIn + (last_keys ? oLastK : 0));   // not read in modes 1, 2: any valid address
    A.mpDesc = dIn + oMpD;
    A.world = dPX;     // unused in these modes
    A.mpValid = dTV;
    A.match12 = mode == 3 ? (int *) P.d_out(rM12) : nullptr;
    A.outlier = is_bad ? dIn + oBad : nullptr;
    A.hasObs = mp_has_obs ? dIn + oObs : nullptr;
    A.lastCnt = (const int *) (dIn + oCnt);
    A.kpStrideLast = (long long) nq;
    A.cntOffLast = 1;
    A.poses = (const float *) (dIn + oPose);
    A.mpProjX = dPX;
    A.mpProjY = dY;
    A.mpProjXR = dXR;
    A.mpViewCos = dVC;
    A.mpAngle = dVC;
    A.mpLevel = dLv;
    A.nnratio = nnratio;
    fill_camera(A, cam, c);
    if (F->scale_factors) for (int l = 0; l < kMaxLevels && l < F->nlevels; l++) A.scaleFactors[l] = F->scale_factors[l];
    A.th = th;
    A.bMono = 1;
    A.checkLevel = check_level != 0;
    A.checkOri = check_ori != 0;
    A.owner = P.d_out(rOwner);
    A.match = (int *) P.d_out(rMatch);
    A.nmatches = (int *) P.d_out(rN);
    A.capCur = (int) nt;
    A.capLast = (int) nq;
    size_t lds;
    if ((rc = plan_match_lds(c, A, 1, &lds))) return rc;
    if (c->matchDebug) {
        if ((rc = ensure(c, c->dTmpC, 8 * sizeof(long long)))) return rc;
        A.dbg = (long long *) c->dTmpC.p;
    }
    {
        ProfScope ps(c, KK_MATCH);
        launch_match_last(c->stream, A, 1, lds);
    }
    HIPCHECK(c, hipGetLastError());
    if (A.dbg) {
        long long st[8];
        HIPCHECK(c, hipMemcpy(st, A.dbg, sizeof st, hipMemcpyDeviceToHost));
        fprintf(stderr, "[ygzf match mode %d, 100MHz ticks] grid %lld  proj %lld  spec %lld  seq %lld  tail %lld  rescans %lld of %lld queries\n", mode,
                st[1] - st[0], st[2] - st[1], st[3] - st[2], st[4] - st[3], st[5] - st[4], st[6], st[7]);
    }
    if ((rc = P.download())) return rc;
    c->lastMatchPairs = 0;
    return YGZF_OK;
}

int ygzf_search_by_bow(ygzf_ctx *c, int n_nodes, const int *kf_off, const int *kf_idx, const int *f_off, const int *f_idx, int n_kf,
                       const uint8_t *kf_valid, const ygzf_kp *kf_keys, const uint8_t *kf_desc, int n_f, const ygzf_kp *f_keys, const uint8_t *f_desc,
                       float nnratio, int check_orientation, int *match, int *nmatches) {
    if (!c || !nmatches || (n_f > 0 && !match)) return fail(c, YGZF_ERR_INVALID, "null argument");
    *nmatches = 0;
    for (int i = 0; i < n_f; i++) match[i] = -1;   // vpMapPointMatches = vector<MapPoint*>(F.N, NULL)  (:158)
    if (n_nodes <= 0 || n_kf <= 0 || n_f <= 0) return YGZF_OK;
    if (!kf_off || !kf_idx || !f_off || !f_idx || !kf_valid || !kf_keys || !kf_desc || !f_keys || !f_desc) return fail(c, YGZF_ERR_INVALID, "null array");
    const int nk = kf_off[n_nodes], nfi = f_off[n_nodes];
    for (int k = 0; k < n_nodes; k++) {
        if (kf_off[k] > kf_off[k + 1] || f_off[k] > f_off[k + 1] || kf_off[k] < 0 || f_off[k] < 0) return fail(c, YGZF_ERR_INVALID, "node offsets not ascending");
        if (f_off[k + 1] - f_off[k] > 4096) return fail(c, YGZF_ERR_UNSUPPORTED, "more than 4096 frame features in one vocabulary node");
    }
    for (int i = 0; i < nk; i++) if (kf_idx[i] < 0 || kf_idx[i] >= n_kf) return fail(c, YGZF_ERR_INVALID, "KeyFrame feature index out of range");
    for (int i = 0; i < nfi; i++) if (f_idx[i] < 0 || f_idx[i] >= n_f) return fail(c, YGZF_ERR_INVALID, "Frame feature index out of range");
    HIPCHECK(c, hipSetDevice(c->device));
    // nine small host arrays in, two out: one packed copy each way (PackedTransfer; nine staged copies of their own until round 4)
    int rc;
    PackedTransfer P(c);
    const size_t iKO = P.add_in(kf_off, 4 * (size_t) (n_nodes + 1)), iKI = P.add_in(kf_idx, 4 * (size_t) nk), iFO = P.add_in(f_off, 4 * (size_t) (n_nodes + 1)),
                 iFI = P.add_in(f_idx, 4 * (size_t) nfi), iKV = P.add_in(kf_valid, (size_t) n_kf), iKK = P.add_in(kf_keys, sizeof(ygzf_kp) * (size_t) n_kf),
                 iKD = P.add_in(kf_desc, 32 * (size_t) n_kf), iFK = P.add_in(f_keys, sizeof(ygzf_kp) * (size_t) n_f), iFD = P.add_in(f_desc, 32 * (size_t) n_f);
    int tail[64];   // [0] nmatches, [4 .. 34) rotation histogram
    const size_t oM = P.add_out(match, 4 * (size_t) n_f), oT = P.add_out(tail, sizeof tail);
    uint8_t *d;
    if ((rc = P.upload(&d)) || (rc = ensure(c, c->dOwner, (size_t) n_f))) return rc;
    int *dMatch = (int *) P.d_out(oM), *dTail = (int *) P.d_out(oT);
    HIPCHECK(c, hipMemsetAsync(dMatch, 0xFF, 4 * (size_t) n_f, c->stream));
    HIPCHECK(c, hipMemsetAsync(dTail, 0, sizeof tail, c->stream));
    {
        ProfScope ps(c, KK_BOWNODES);
        launch_bow(c->stream, n_nodes, (const int *) (d + iKO), (const int *) (d + iKI), (const int *) (d + iFO), (const int *) (d + iFI), d + iKV,
                   (const ygzf_kp *) (d + iKK), d + iKD, n_f, (const ygzf_kp *) (d + iFK), d + iFD, nnratio, check_orientation != 0, dMatch,
                   (unsigned char *) c->dOwner.p, dTail + 4, dTail);
    }
    HIPCHECK(c, hipGetLastError());
    if ((rc = P.download())) return rc;
    *nmatches = tail[0];
    c->lastMatchPairs = 0;
    return YGZF_OK;
}

int ygzf_search_for_triangulation(ygzf_ctx *c, int n_nodes, const int *off1, const int *idx1, const int *off2, const int *idx2,
                                  const ygzf_frame_view *kf1, const uint8_t *has_mp1, const ygzf_frame_view *kf2, const uint8_t *has_mp2,
                                  const float *level_sigma2_2, const float *F12, const float *Cw1, const float *R2w, const float *t2w,
                                  const ygzf_camera *cam2, int only_stereo, int check_orientation, int *match12, int *nmatches) {
    if (!c || !kf1 || !kf2 || !nmatches || (kf1->n > 0 && !match12)) return fail(c, YGZF_ERR_INVALID, "null argument");
    *nmatches = 0;
    const int n1 = kf1->n, n2 = kf2->n;
    for (int i = 0; i < n1; i++) match12[i] = -1;   // vMatches12 = vector<int>(pKF1->N, -1)  (:617)
    if (n_nodes <= 0 || n1 <= 0 || n2 <= 0) return YGZF_OK;
    if (!off1 || !idx1 || !off2 || !idx2 || !has_mp1 || !has_mp2 || !kf1->keys || !kf1->desc || !kf2->keys || !kf2->desc || !F12 || !Cw1 || !R2w ||
        !t2w || !cam2)
        return fail(c, YGZF_ERR_INVALID, "null array");
    const int L = kf2->scale_factors ? kf2->nlevels : c->tab.cfg.nlevels;
    if (L <= 0) return fail(c, YGZF_ERR_INVALID, "no scale levels");
    if (off1[0] != 0 || off2[0] != 0) return fail(c, YGZF_ERR_INVALID, "node offsets do not start at 0");
    for (int k = 0; k < n_nodes; k++) {
        if (off1[k] > off1[k + 1] || off2[k] > off2[k + 1]) return fail(c, YGZF_ERR_INVALID, "node offsets not ascending");
        if (off2[k + 1] - off2[k] > 65535) return fail(c, YGZF_ERR_UNSUPPORTED, "more than 65535 features of the second KeyFrame in one vocabulary node");
    }
    const int ne1 = off1[n_nodes], ne2 = off2[n_nodes];
    for (int i = 0; i < ne1; i++) if (idx1[i] < 0 || idx1[i] >= n1) return fail(c, YGZF_ERR_INVALID, "feature index of the first KeyFrame out of range");
    for (int i = 0; i < ne2; i++) if (idx2[i] < 0 || idx2[i] >= n2) return fail(c, YGZF_ERR_INVALID, "feature index of the second KeyFrame out of range");
    for (int i = 0; i < n2; i++)
        if (kf2->keys[i].octave < 0 || kf2->keys[i].octave >= L) return fail(c, YGZF_ERR_INVALID, "keypoint octave outside the scale tables");
    HIPCHECK(c, hipSetDevice(c->device));
    std::vector<float> sf(L), sg(L);
    for (int l = 0; l < L; l++) {
        sf[l] = kf2->scale_factors ? kf2->scale_factors[l] : c->tab.scale[l];
        sg[l] = level_sigma2_2 ? level_sigma2_2[l] : sf[l] * sf[l];   // mvLevelSigma2[i] = mvScaleFactor[i] * mvScaleFactor[i]  (src/ORBextractor.cc:422)
    }
    TriArgs A;
    {   // epipole in the second image (:601-608): C2 = R2w * Cw + t2w, coefficient order of the 3x3 product
        float C2[3];
        for (int r = 0; r < 3; r++) C2[r] = R2w[3 * r] * Cw1[0] + R2w[3 * r + 1] * Cw1[1] + R2w[3 * r + 2] * Cw1[2];
        for (int r = 0; r < 3; r++) C2[r] = C2[r] + t2w[r];
        const float invz = 1.0f / C2[2];
        A.ex = cam2->fx * C2[0] * invz + cam2->cx;
        A.ey = cam2->fy * C2[1] * invz + cam2->cy;
    }
    for (int i = 0; i < 9; i++) A.F[i] = F12[i];
    int rc;
    PackedTransfer P(c);
    const size_t N1 = (size_t) n1, N2 = (size_t) n2;
    const size_t iO1 = P.add_in(off1, 4 * (size_t) (n_nodes + 1)), iI1 = P.add_in(idx1, 4 * (size_t) ne1), iO2 = P.add_in(off2, 4 * (size_t) (n_nodes + 1)),
                 iI2 = P.add_in(idx2, 4 * (size_t) ne2), iK1 = P.add_in(kf1->keys, sizeof(ygzf_kp) * N1), iK2 = P.add_in(kf2->keys, sizeof(ygzf_kp) * N2),
                 iD1 = P.add_in(kf1->desc, 32 * N1), iD2 = P.add_in(kf2->desc, 32 * N2), iM1 = P.add_in(has_mp1, N1), iM2 = P.add_in(has_mp2, N2),
                 iU1 = P.add_in(kf1->u_right, kf1->u_right ? 4 * N1 : 0), iU2 = P.add_in(kf2->u_right, kf2->u_right ? 4 * N2 : 0),
                 iSf = P.add_in(sf.data(), 4 * (size_t) L), iSg = P.add_in(sg.data(), 4 * (size_t) L);
    int tail[64];   // [0] nmatches, [4 .. 34) rotation histogram
    const size_t oM = P.add_out(match12, 4 * N1), oT = P.add_out(tail, sizeof(tail));
    uint8_t *d;
    if ((rc = P.upload(&d)) || (rc = ensure(c, c->dGen[9], N1 + 16))) return rc;
    A.nEntries = ne1; A.nNodes = n_nodes; A.n1 = n1;
    A.off1 = (const int *) (d + iO1); A.idx1 = (const int *) (d + iI1); A.off2 = (const int *) (d + iO2); A.idx2 = (const int *) (d + iI2);
    A.keys1 = (const ygzf_kp *) (d + iK1); A.keys2 = (const ygzf_kp *) (d + iK2);
    A.desc1 = d + iD1; A.desc2 = d + iD2; A.hasMp1 = d + iM1; A.hasMp2 = d + iM2;
    A.uR1 = kf1->u_right ? (const float *) (d + iU1) : nullptr;
    A.uR2 = kf2->u_right ? (const float *) (d + iU2) : nullptr;
    A.sf2 = (const float *) (d + iSf); A.sigma2 = (const float *) (d + iSg);
    A.onlyStereo = only_stereo != 0; A.checkOri = check_orientation != 0;
    A.match12 = (int *) P.d_out(oM);
    A.binOf = (unsigned char *) c->dGen[9].p;
    A.nmatches = (int *) P.d_out(oT);
    A.hist = A.nmatches + 4;
    HIPCHECK(c, hipMemsetAsync(P.d_out(oM), 0xFF, 4 * N1, c->stream));
    HIPCHECK(c, hipMemsetAsync(P.d_out(oT), 0, sizeof(tail), c->stream));
    {
        ProfScope ps(c, KK_TRI);
        launch_triangulation(c->stream, A);
    }
    HIPCHECK(c, hipGetLastError());
    if ((rc = P.download())) return rc;
    *nmatches = tail[0];
    c->lastMatchPairs = 0;
    return YGZF_OK;
}

int ygzf_search_for_initialization(ygzf_ctx *c, const ygzf_frame_view *F1, const ygzf_frame_view *F2, const ygzf_camera *cam, float *prev_matched_xy,
                                   int window_size, float nnratio, int check_orientation, int *matches12, int *nmatches) {
    if (!c || !F1 || !F2 || !cam || !nmatches || !matches12 || !prev_matched_xy) return fail(c, YGZF_ERR_INVALID, "null argument");
    *nmatches = 0;
    for (int i = 0; i < F1->n; i++) matches12[i] = -1;   // vnMatches12 = vector<int>(F1.N, -1)  (:379)
    if (F1->n <= 0 || F2->n <= 0) return YGZF_OK;
    std::vector<float> px(F1->n), py(F1->n);
    for (int i = 0; i < F1->n; i++) { px[i] = prev_matched_xy[2 * i]; py[i] = prev_matched_xy[2 * i + 1]; }
    std::vector<uint8_t> owner(F2->n, 0);
    std::vector<int> match21(F2->n, -1);
    int rc = projected_match(c, 3, F2, cam, F1->n, nullptr, nullptr, nullptr, px.data(), py.data(), nullptr, nullptr, nullptr, nullptr, F1->desc,
                             (float) window_size, 0, nnratio, 50, check_orientation, owner.data(), match21.data(), nmatches, F1->keys, matches12);
    if (rc) return rc;
    for (int i = 0; i < F1->n; i++)      // :470-474 update prev matched
        if (matches12[i] >= 0) {
            prev_matched_xy[2 * i] = F2->keys[matches12[i]].x;
            prev_matched_xy[2 * i + 1] = F2->keys[matches12[i]].y;
        }
    return YGZF_OK;
}

int ygzf_search_by_projection_mappoints(ygzf_ctx *c, const ygzf_frame_view *F, const ygzf_camera *cam, int n_mp, const uint8_t *track_in_view,
                                        const uint8_t *is_bad, const uint8_t *mp_has_obs, const float *proj_x, const float *proj_y,
                                        const float *proj_xr, const float *view_cos, const int *scale_level, const uint8_t *mp_desc, float th,
                                        int check_level, float nnratio, uint8_t *owner, int *match, int *nmatches) {
    return projected_match(c, 1, F, cam, n_mp, track_in_view, is_bad, mp_has_obs, proj_x, proj_y, proj_xr, view_cos, scale_level, nullptr, mp_desc,
                           th, check_level, nnratio, 100, 0, owner, match, nmatches);
}

int ygzf_predict_scale_steps(float log_scale_factor, int nlevels, float *steps) {
    if (!steps || nlevels < 1 || nlevels > kMaxLevels) return YGZF_ERR_INVALID;
    float st[kMaxLevels];
    predict_scale_steps(log_scale_factor, nlevels, st);
    for (int k = 0; k < nlevels; k++) steps[k] = k == 0 ? 0.f : st[k];
    return YGZF_OK;
}

int ygzf_is_in_frustum_batch(ygzf_ctx *c, const ygzf_camera *cam, int nlevels, int n, const ygzf_frustum_in *in, uint8_t *in_view, float *proj_x,
                             float *proj_y, float *proj_xr, int *level, float *view_cos) {
    if (!c || !cam || !in || !in_view || !proj_x || !proj_y || !proj_xr || !level || !view_cos) return fail(c, YGZF_ERR_INVALID, "null argument");
    if (n < 0) return fail(c, YGZF_ERR_INVALID, "negative count");
    if (n == 0) return YGZF_OK;
    HIPCHECK(c, hipSetDevice(c->device));
    int rc;
    PackedTransfer P(c);
    FrustumOffsets FO;
    if ((rc = frustum_add_inputs(c, P, FO, in, n, nlevels))) return rc;
    const size_t N = (size_t) n;
    const size_t rPX = P.add_out(proj_x, 4 * N), rPY = P.add_out(proj_y, 4 * N), rPXR = P.add_out(proj_xr, 4 * N), rVC = P.add_out(view_cos, 4 * N),
                 rLv = P.add_out(level, 4 * N), rIV = P.add_out(in_view, N);
    uint8_t *dIn;
    if ((rc = P.upload(&dIn))) return rc;
    FrustumArgs A;
    frustum_fill_args(A, dIn, FO, in, cam, n, nlevels);
    A.inView = P.d_out(rIV);
    A.projX = (float *) P.d_out(rPX); A.projY = (float *) P.d_out(rPY); A.projXR = (float *) P.d_out(rPXR); A.viewCos = (float *) P.d_out(rVC);
    A.level = (int *) P.d_out(rLv);
    HIPCHECK(c, hipMemsetAsync(P.d_out(0), 0, P.outBytes, c->stream));   // rejected points read as zeros
    {
        ProfScope ps(c, KK_FRUSTUM);
        launch_frustum(c->stream, A);
    }
    HIPCHECK(c, hipGetLastError());
    return P.download();
}

int ygzf_search_local_points(ygzf_ctx *c, const ygzf_frame_view *F, const ygzf_camera *cam, int n_mp, const ygzf_frustum_in *in,
                             const uint8_t *mp_has_obs, const uint8_t *mp_desc, float th, int check_level, float nnratio, uint8_t *owner, int *match,
                             int *nmatches, uint8_t *in_view, float *proj_x, float *proj_y, float *proj_xr, int *level, float *view_cos) {
    if (!in) return fail(c, YGZF_ERR_INVALID, "null argument");
    FrustumHost fr = {in, in_view, proj_x, proj_y, proj_xr, view_cos, level};
    return projected_match(c, 1, F, cam, n_mp, nullptr, nullptr, mp_has_obs, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, mp_desc, th, check_level,
                           nnratio, 100, 0, owner, match, nmatches, nullptr, nullptr, &fr);
}

int ygzf_features_in_area(ygzf_ctx *c, const ygzf_camera *cam, int n_keys, const ygzf_kp *keys, int n_queries, const float *xyr, const int *levels,
                          int cap, int *out_idx, int *out_n) {
    if (!c || !cam || (n_keys > 0 && !keys) || (n_queries > 0 && (!xyr || !out_n)) || (n_queries > 0 && cap > 0 && !out_idx))
        return fail(c, YGZF_ERR_INVALID, "null argument");
    if (n_keys < 0 || n_queries < 0 || cap < 0) return fail(c, YGZF_ERR_INVALID, "negative count");
    if (!(cam->max_x > cam->min_x) || !(cam->max_y > cam->min_y)) return fail(c, YGZF_ERR_INVALID, "empty image bounds");
    if (n_queries == 0) return YGZF_OK;
    if (fia_lds_bytes(n_keys) > (size_t) kMaxDynLds) return fail(c, YGZF_ERR_UNSUPPORTED, "more than %d keypoints in one grid", (int) ((kMaxDynLds - 25000) / 4));
    HIPCHECK(c, hipSetDevice(c->device));
    int rc;
    const size_t qBytes = (size_t) n_queries * 12, lBytes = levels ? (size_t) n_queries * 8 : 0;
    const size_t oBytes = (size_t) n_queries * (size_t) cap * 4, nBytes = (size_t) n_queries * 4;
    const size_t nPad = (nBytes + 15) & ~(size_t) 15;
    if ((size_t) n_keys * sizeof(ygzf_kp) + qBytes + lBytes + nBytes + oBytes <= kPackedMax) {   // one packed copy each way
        PackedTransfer P(c);
        const size_t iK = P.add_in(keys, (size_t) n_keys * sizeof(ygzf_kp)), iQ = P.add_in(xyr, qBytes), iL = P.add_in(levels, lBytes);
        const size_t oN = P.add_out(out_n, nBytes), oI = P.add_out(out_idx, oBytes);
        uint8_t *d;
        if ((rc = P.upload(&d))) return rc;
        FiaArgs A;
        A.keys = (const ygzf_kp *) (d + iK);
        A.n = n_keys;
        A.minX = cam->min_x; A.minY = cam->min_y;
        A.gridInvW = (float) 64 / (cam->max_x - cam->min_x);
        A.gridInvH = (float) 48 / (cam->max_y - cam->min_y);
        A.nq = n_queries;
        A.xyr = (const float *) (d + iQ);
        A.levels = levels ? (const int *) (d + iL) : nullptr;
        A.cap = cap;
        A.outN = (int *) P.d_out(oN);
        A.outIdx = (int *) P.d_out(oI);
        {
            ProfScope ps(c, KK_GRID);
            HIPCHECK(c, launch_features_in_area(c->stream, A));
        }
        HIPCHECK(c, hipGetLastError());
        return P.download();
    }
    if ((rc = ensure(c, c->dTmpA, (size_t) n_keys * sizeof(ygzf_kp) + 64)) || (rc = ensure(c, c->dTmpB, qBytes + lBytes + 64)) ||
        (rc = ensure(c, c->dTmpC, nPad + oBytes + 64)))
        return rc;
    if (n_keys > 0) HIPCHECK(c, hipMemcpyAsync(c->dTmpA.p, keys, (size_t) n_keys * sizeof(ygzf_kp), hipMemcpyHostToDevice, c->stream));
    HIPCHECK(c, hipMemcpyAsync(c->dTmpB.p, xyr, qBytes, hipMemcpyHostToDevice, c->stream));
    if (levels) HIPCHECK(c, hipMemcpyAsync((uint8_t *) c->dTmpB.p + qBytes, levels, lBytes, hipMemcpyHostToDevice, c->stream));
    FiaArgs A;
    A.keys = (const ygzf_kp *) c->dTmpA.p;
    A.n = n_keys;
    A.minX = cam->min_x; A.minY = cam->min_y;
    A.gridInvW = (float) 64 / (cam->max_x - cam->min_x);   // mfGridElementWidthInv / HeightInv, src/Frame.cc:302-303
    A.gridInvH = (float) 48 / (cam->max_y - cam->min_y);
    A.nq = n_queries;
    A.xyr = (const float *) c->dTmpB.p;
    A.levels = levels ? (const int *) ((uint8_t *) c->dTmpB.p + qBytes) : nullptr;
    A.cap = cap;
    A.outN = (int *) c->dTmpC.p;
    A.outIdx = (int *) ((uint8_t *) c->dTmpC.p + nPad);
    {
        ProfScope ps(c, KK_GRID);
        HIPCHECK(c, launch_features_in_area(c->stream, A));
    }
    HIPCHECK(c, hipGetLastError());
    HIPCHECK(c, hipMemcpyAsync(out_n, A.outN, nBytes, hipMemcpyDeviceToHost, c->stream));
    if (oBytes) HIPCHECK(c, hipMemcpyAsync(out_idx, A.outIdx, oBytes, hipMemcpyDeviceToHost, c->stream));
    HIPCHECK(c, hipStreamSynchronize(c->stream));
    return YGZF_OK;
}

int ygzf_distinctive_descriptors_batch(ygzf_ctx *c, int n_points, const int *obs_off, const uint8_t *desc, int *best_idx) {
    if (!c || (n_points > 0 && (!obs_off || !best_idx))) return fail(c, YGZF_ERR_INVALID, "null argument");
    if (n_points <= 0) return YGZF_OK;
    const int total = obs_off[n_points];
    std::vector<int> large;          // points beyond the register-resident form (> 256 observations): the histogram kernel's
    for (int p = 0; p < n_points; p++) {
        const int n = obs_off[p + 1] - obs_off[p];
        if (n < 0 || obs_off[p] < 0) return fail(c, YGZF_ERR_INVALID, "observation offsets not ascending");
        if (n > 256) large.push_back(p);
    }
    if (total > 0 && !desc) return fail(c, YGZF_ERR_INVALID, "null descriptors");
    HIPCHECK(c, hipSetDevice(c->device));
    int rc;
    const size_t offBytes = (4 * (size_t) (n_points + 1) + 15) & ~(size_t) 15;
    if ((rc = ensure(c, c->dTmpA, offBytes + 4 * large.size() + 16)) || (rc = ensure(c, c->dTmpB, 4 * (size_t) n_points)) ||
        (rc = ensure(c, c->dTmpC, 32 * (size_t) (total + 1))))
        return rc;
    HIPCHECK(c, hipMemcpyAsync(c->dTmpA.p, obs_off, 4 * (size_t) (n_points + 1), hipMemcpyHostToDevice, c->stream));
    if (!large.empty()) HIPCHECK(c, hipMemcpyAsync((uint8_t *) c->dTmpA.p + offBytes, large.data(), 4 * large.size(), hipMemcpyHostToDevice, c->stream));
    if (total > 0) HIPCHECK(c, hipMemcpyAsync(c->dTmpC.p, desc, 32 * (size_t) total, hipMemcpyHostToDevice, c->stream));
    {
        ProfScope ps(c, KK_DISTINCTIVE);
        launch_distinctive(c->stream, n_points, (const int *) c->dTmpA.p, (const uint8_t *) c->dTmpC.p, (int *) c->dTmpB.p, (int) large.size(),
                           (const int *) ((uint8_t *) c->dTmpA.p + offBytes));
    }
    HIPCHECK(c, hipGetLastError());
    HIPCHECK(c, hipMemcpyAsync(best_idx, c->dTmpB.p, 4 * (size_t) n_points, hipMemcpyDeviceToHost, c->stream));
    HIPCHECK(c, hipStreamSynchronize(c->stream));
    return YGZF_OK;
}

// ---- Frame::ComputeBoW: vocabulary on the device + tree descent ------------------------------------------------------------------------
int ygzf_vocabulary_set(ygzf_ctx *c, int n_nodes, int depth_levels, const int *parent, const uint8_t *desc) {
    if (!c || !parent || !desc) return fail(c, YGZF_ERR_INVALID, "null argument");
    if (n_nodes < 1 || depth_levels < 0) return fail(c, YGZF_ERR_INVALID, "bad vocabulary size");
    HIPCHECK(c, hipSetDevice(c->device));
    // children lists in ascending node id (= the loaders' push_back order), as CSR
    std::vector<int> off((size_t) n_nodes + 1, 0), idx((size_t) std::max(n_nodes - 1, 1));
    for (int i = 1; i < n_nodes; i++) {
        if (parent[i] < 0 || parent[i] >= n_nodes || parent[i] == i) return fail(c, YGZF_ERR_INVALID, "node %d: parent %d out of range", i, parent[i]);
        off[(size_t) parent[i] + 1]++;
    }
    for (int i = 0; i < n_nodes; i++) off[(size_t) i + 1] += off[i];
    {
        std::vector<int> fill(off.begin(), off.end() - 1);
        for (int i = 1; i < n_nodes; i++) idx[(size_t) fill[parent[i]]++] = i;
    }
    int rc;
    if ((rc = ensure(c, c->dVoc[0], off.size() * sizeof(int))) || (rc = ensure(c, c->dVoc[1], idx.size() * sizeof(int))) ||
        (rc = ensure(c, c->dVoc[2], (size_t) n_nodes * 32)))
        return rc;
    c->vocNodes = 0;
    HIPCHECK(c, hipMemcpyAsync(c->dVoc[0].p, off.data(), off.size() * sizeof(int), hipMemcpyHostToDevice, c->stream));
    HIPCHECK(c, hipMemcpyAsync(c->dVoc[1].p, idx.data(), idx.size() * sizeof(int), hipMemcpyHostToDevice, c->stream));
    HIPCHECK(c, hipMemcpyAsync(c->dVoc[2].p, desc, (size_t) n_nodes * 32, hipMemcpyHostToDevice, c->stream));
    HIPCHECK(c, hipStreamSynchronize(c->stream));
    c->vocNodes = n_nodes;
    c->vocLevels = depth_levels;
    return YGZF_OK;
}

int ygzf_bow_transform(ygzf_ctx *c, int n, const uint8_t *desc, int levelsup, int *leaf_node, int *level_node) {
    if (!c) return YGZF_ERR_INVALID;
    if (c->vocNodes < 1) return fail(c, YGZF_ERR_STATE, "no vocabulary on the device (ygzf_vocabulary_set)");
    if (n < 0) return fail(c, YGZF_ERR_INVALID, "negative count");
    if (n == 0) return YGZF_OK;
    if (!desc || !leaf_node || !level_node) return fail(c, YGZF_ERR_INVALID, "null argument");
    HIPCHECK(c, hipSetDevice(c->device));
    int rc;
    if ((rc = ensure(c, c->dBow[0], (size_t) n * 32)) || (rc = ensure(c, c->dBow[1], (size_t) n * 4)) || (rc = ensure(c, c->dBow[2], (size_t) n * 4))) return rc;
    HIPCHECK(c, hipMemcpyAsync(c->dBow[0].p, desc, (size_t) n * 32, hipMemcpyHostToDevice, c->stream));
    {
        ProfScope ps(c, KK_BOW);
        launch_bow_descend(c->stream, n, (const uint8_t *) c->dBow[0].p, (const int *) c->dVoc[0].p, (const int *) c->dVoc[1].p, (const uint8_t *) c->dVoc[2].p,
                           c->vocLevels - levelsup, (int *) c->dBow[1].p, (int *) c->dBow[2].p);
    }
    HIPCHECK(c, hipGetLastError());
    HIPCHECK(c, hipMemcpyAsync(leaf_node, c->dBow[1].p, (size_t) n * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHECK(c, hipMemcpyAsync(level_node, c->dBow[2].p, (size_t) n * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHECK(c, hipStreamSynchronize(c->stream));
    return YGZF_OK;
}

int ygzf_search_by_projection_kf(ygzf_ctx *c, const ygzf_frame_view *cur, const ygzf_camera *cam, int n_mp, const uint8_t *valid,
                                 const float *proj_x, const float *proj_y, const int *pred_level, const float *kf_angle, const uint8_t *mp_desc,
                                 float th, int orb_dist, int check_orientation, uint8_t *owner, int *match, int *nmatches) {
    if (!c || !cur || !owner) return fail(c, YGZF_ERR_INVALID, "null argument");
    if (orb_dist < 0 || orb_dist > 256) return fail(c, YGZF_ERR_INVALID, "ORBdist %d outside 0..256", orb_dist);
    for (int i = 0; i < cur->n; i++) owner[i] = owner[i] ? 2 : 0;   // `if (CurrentFrame.mvpMapPoints[i2]) continue;` (:1419): any MapPoint blocks
    return projected_match(c, 2, cur, cam, n_mp, valid, nullptr, nullptr, proj_x, proj_y, nullptr, nullptr, pred_level, kf_angle, mp_desc, th, 0,
                           0.f, orb_dist, check_orientation, owner, match, nmatches);
}

// ---- Frame::ComputeStereoMatches ----------------------------------------------------------------------------------------------
static void fill_stereo_common(ygzf_ctx *c, StereoArgs &A, float mb, float mbf, int h) {
    const int L = c->tab.cfg.nlevels;
    for (int l = 0; l < kMaxLevels; l++) {
        A.scale[l] = l < L ? c->tab.scale[l] : 1.f;
        A.invScale[l] = l < L ? c->tab.invScale[l] : 1.f;
    }
    A.mb = mb;
    A.mbf = mbf;
    A.nRows = h;
    A.geom = (const LevelGeom *) c->dGeom.p;
    float smax = 1.f;
    for (int l = 0; l < L; l++) smax = std::max(smax, c->tab.scale[l]);
    A.bandMax = (int) std::ceil(4.0 * (double) smax) + 2;      // ceil(y + r) - floor(y - r) <= 2 r + 2, r = 2 * scale
    A.binShift = 3;
    while ((((std::max(h, 1) - 1) >> A.binShift) + 1) > kStereoBinInts - 1) A.binShift++;
    A.nBins = ((std::max(h, 1) - 1) >> A.binShift) + 1;
}

int ygzf_stereo_batch(ygzf_ctx *c, float mb, float mbf) {
    if (!c) return YGZF_ERR_INVALID;
    if (c->lastFrames < 2 || (c->lastFrames & 1)) return fail(c, YGZF_ERR_STATE, "stereo needs an extracted batch of (left, right) frame pairs");
    if (!(mb > 0)) return fail(c, YGZF_ERR_INVALID, "baseline mb must be positive");
    HIPCHECK(c, hipSetDevice(c->device));
    const Geometry &G = c->geo;
    const int P = c->lastFrames / 2;
    if (G.kpStride > 65535) return fail(c, YGZF_ERR_UNSUPPORTED, "more than 65535 keypoints per frame");
    int rc;
    const size_t per = (size_t) G.kpStride;
    if ((rc = ensure(c, c->dSt[0], sizeof(StereoRec) * per * P + 64)) || (rc = ensure(c, c->dSt[1], 4 * per * P + 64)) ||
        (rc = ensure(c, c->dSt[2], 4 * per * P + 64)) || (rc = ensure(c, c->dSt[3], 4 * per * P + 64)) ||
        (rc = ensure(c, c->dStBins, sizeof(int) * kStereoBinInts * (size_t) P)))
        return rc;
    StereoArgs A;
    memset(&A, 0, sizeof A);
    A.keys = (const ygzf_kp *) c->dOutKp.p;       // slot 0 = carry; frame f lives in slot f + 1
    A.desc = (const uint8_t *) c->dOutDesc.p;
    A.keyStride = 2 * (long long) G.kpStride;
    A.keyOffL = G.kpStride;
    A.keyOffR = 2 * G.kpStride;
    A.cnt = (const int *) c->dOutCnt.p;
    A.cntStride = 2;
    A.cntOffL = 1;
    A.cntOffR = 2;
    A.fs = c->lastFs;
    A.frame0 = 0;
    A.frameStep = 2;
    fill_stereo_common(c, A, mb, mbf, G.h);
    A.rec = (StereoRec *) c->dSt[0].p;
    A.recStride = (long long) per;
    A.binStart = (int *) c->dStBins.p;
    A.uRight = (float *) c->dSt[1].p;
    A.depth = (float *) c->dSt[2].p;
    A.sad = (int *) c->dSt[3].p;
    A.outStride = (long long) per;
    {
        ProfScope ps(c, KK_STEREO);
        launch_stereo(c->stream, A, P, G.kpStride, G.kpStride);
    }
    HIPCHECK(c, hipGetLastError());
    c->lastStereoPairs = P;
    return YGZF_OK;
}

int ygzf_stereo_fetch(ygzf_ctx *c, int pair, float *u_right, float *depth, int cap) {
    if (!c || !u_right || !depth) return fail(c, YGZF_ERR_INVALID, "null argument");
    if (pair < 0 || pair >= c->lastStereoPairs) return fail(c, YGZF_ERR_STATE, "pair %d: no stereo result", pair);
    int n = 0;
    HIPCHECK(c, hipMemcpyAsync(&n, (int *) c->dOutCnt.p + 1 + 2 * pair, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIPCHECK(c, hipStreamSynchronize(c->stream));
    if (n > cap) return fail(c, YGZF_ERR_INVALID, "capacity %d < %d left keypoints", cap, n);
    if (n == 0) return YGZF_OK;
    const size_t off = (size_t) pair * c->geo.kpStride;
    HIPCHECK(c, hipMemcpyAsync(u_right, (float *) c->dSt[1].p + off, 4 * (size_t) n, hipMemcpyDeviceToHost, c->stream));
    HIPCHECK(c, hipMemcpyAsync(depth, (float *) c->dSt[2].p + off, 4 * (size_t) n, hipMemcpyDeviceToHost, c->stream));
    HIPCHECK(c, hipStreamSynchronize(c->stream));
    return YGZF_OK;
}

// every pair of the last ygzf_stereo_batch at once: rows of `stride` floats (>= ygzf_max_keypoints), pair p's first n_kp[2 p] entries valid
int ygzf_stereo_fetch_all(ygzf_ctx *c, float *u_right, float *depth, int stride) {
    if (!c || !u_right || !depth) return fail(c, YGZF_ERR_INVALID, "null argument");
    if (c->lastStereoPairs < 1) return fail(c, YGZF_ERR_STATE, "no stereo result");
    const int ks = c->geo.kpStride, P = c->lastStereoPairs;
    if (stride < ks) return fail(c, YGZF_ERR_INVALID, "stride %d < %d (ygzf_max_keypoints)", stride, ks);
    HIPCHECK(c, hipMemcpy2DAsync(u_right, 4 * (size_t) stride, c->dSt[1].p, 4 * (size_t) ks, 4 * (size_t) ks, P, hipMemcpyDeviceToHost, c->stream));
    HIPCHECK(c, hipMemcpy2DAsync(depth, 4 * (size_t) stride, c->dSt[2].p, 4 * (size_t) ks, 4 * (size_t) ks, P, hipMemcpyDeviceToHost, c->stream));
    HIPCHECK(c, hipStreamSynchronize(c->stream));
    return YGZF_OK;
}

int ygzf_compute_stereo_matches(ygzf_ctx *c, const uint8_t *img_left, const uint8_t *img_right, int w, int h, int stride, int n_left,
                                const ygzf_kp *keys_left, const uint8_t *desc_left, int n_right, const ygzf_kp *keys_right, const uint8_t *desc_right,
                                float mb, float mbf, float *u_right, float *depth) {
    if (!c || !img_left || !img_right) return fail(c, YGZF_ERR_INVALID, "null argument");
    if (n_left < 0 || n_right < 0 || n_right > 65535) return fail(c, YGZF_ERR_INVALID, "bad keypoint counts");
    if (n_left == 0) return YGZF_OK;
    if (!keys_left || !desc_left || !u_right || !depth || (n_right > 0 && (!keys_right || !desc_right))) return fail(c, YGZF_ERR_INVALID, "null array");
    if (!(mb > 0)) return fail(c, YGZF_ERR_INVALID, "baseline mb must be positive");
    const int L = c->tab.cfg.nlevels;
    for (int i = 0; i < n_left; i++) if (keys_left[i].octave < 0 || keys_left[i].octave >= L) return fail(c, YGZF_ERR_INVALID, "left key %d: octave out of range", i);
    for (int i = 0; i < n_right; i++) if (keys_right[i].octave < 0 || keys_right[i].octave >= L) return fail(c, YGZF_ERR_INVALID, "right key %d: octave out of range", i);
    HIPCHECK(c, hipSetDevice(c->device));
    if (c->maxBatch < 2) return fail(c, YGZF_ERR_INVALID, "context created with max_batch < 2");
    int rc = apply_geometry(c, w, h, 2);
    if (rc) return rc;
    // both eyes' pyramids: the two extractor instances computed exactly these levels (ComputePyramid), recomputed here on the device
    std::vector<uint8_t> both((size_t) 2 * w * h);
    for (int y = 0; y < h; y++) {
        memcpy(&both[(size_t) y * w], img_left + (size_t) y * stride, w);
        memcpy(&both[(size_t) (h + y) * w], img_right + (size_t) y * stride, w);
    }
    FrameSet fs;
    if ((rc = upload_frames(c, both.data(), 2, w, h, w, (size_t) w * h, &fs))) return rc;
    const Geometry &G = c->geo;
    for (int l = 1; l < L; l++) {
        ProfScope ps(c, KK_PYR);
        launch_pyr_resize(c->stream, fs, (const LevelGeom *) c->dGeom.p, G.lv[l], l, 2, (const int *) c->dXofs.p, (const short *) c->dXalpha.p,
                          (const int *) c->dYofs.p, (const short *) c->dYbeta.p);
    }
    c->lastFrames = 0;
    c->carryValid = false;
    const size_t nk = (size_t) n_left + n_right;
    int counts[2] = {n_left, n_right};
    if ((rc = ensure(c, c->dSt[0], sizeof(StereoRec) * (size_t) (n_right + 1))) || (rc = ensure(c, c->dSt[1], 4 * (size_t) n_left)) ||
        (rc = ensure(c, c->dSt[2], 4 * (size_t) n_left)) || (rc = ensure(c, c->dSt[3], 4 * (size_t) n_left)) ||
        (rc = ensure(c, c->dSt[4], sizeof(ygzf_kp) * nk + 64)) || (rc = ensure(c, c->dSt[5], 32 * nk + 64)) ||
        (rc = ensure(c, c->dStBins, sizeof(int) * kStereoBinInts)))
        return rc;
    uint8_t *dk = (uint8_t *) c->dSt[4].p, *dd = (uint8_t *) c->dSt[5].p;
    HIPCHECK(c, hipMemcpyAsync(dk, keys_left, sizeof(ygzf_kp) * (size_t) n_left, hipMemcpyHostToDevice, c->stream));
    HIPCHECK(c, hipMemcpyAsync(dd, desc_left, 32 * (size_t) n_left, hipMemcpyHostToDevice, c->stream));
    if (n_right > 0) {
        HIPCHECK(c, hipMemcpyAsync(dk + sizeof(ygzf_kp) * (size_t) n_left, keys_right, sizeof(ygzf_kp) * (size_t) n_right, hipMemcpyHostToDevice, c->stream));
        HIPCHECK(c, hipMemcpyAsync(dd + 32 * (size_t) n_left, desc_right, 32 * (size_t) n_right, hipMemcpyHostToDevice, c->stream));
    }
    if ((rc = ensure(c, c->dNMatch, 8))) return rc;
    HIPCHECK(c, hipMemcpyAsync(c->dNMatch.p, counts, sizeof counts, hipMemcpyHostToDevice, c->stream));
    StereoArgs A;
    memset(&A, 0, sizeof A);
    A.keys = (const ygzf_kp *) dk;
    A.desc = dd;
    A.keyStride = (long long) nk;
    A.keyOffL = 0;
    A.keyOffR = n_left;
    A.cnt = (const int *) c->dNMatch.p;
    A.cntStride = 2;
    A.cntOffL = 0;
    A.cntOffR = 1;
    A.fs = fs;
    A.frame0 = 0;
    A.frameStep = 2;
    fill_stereo_common(c, A, mb, mbf, h);
    A.rec = (StereoRec *) c->dSt[0].p;
    A.recStride = n_right + 1;
    A.binStart = (int *) c->dStBins.p;
    A.uRight = (float *) c->dSt[1].p;
    A.depth = (float *) c->dSt[2].p;
    A.sad = (int *) c->dSt[3].p;
    A.outStride = n_left;
    {
        ProfScope ps(c, KK_STEREO);
        launch_stereo(c->stream, A, 1, n_left, n_right);
    }
    HIPCHECK(c, hipGetLastError());
    HIPCHECK(c, hipMemcpyAsync(u_right, c->dSt[1].p, 4 * (size_t) n_left, hipMemcpyDeviceToHost, c->stream));
    HIPCHECK(c, hipMemcpyAsync(depth, c->dSt[2].p, 4 * (size_t) n_left, hipMemcpyDeviceToHost, c->stream));
    HIPCHECK(c, hipStreamSynchronize(c->stream));
    c->lastStereoPairs = 0;
    return YGZF_OK;
}

// ---- image cache (KeyFrame / current-frame pyramids resident in HBM) + FindDirectProjection batch -------------------------------------
static FrameSet cache_frameset(const ygzf_ctx *c) {
    FrameSet fs;
    fs.img0 = (const uint8_t *) c->dCacheImg.p;
    fs.img0_stride = (long long) c->cachePitch * c->cacheH;
    fs.img0_pitch = c->cachePitch;
    fs.pyr = (uint8_t *) c->dCachePyr.p;
    fs.pyr_stride = c->cachePyrBytes;
    return fs;
}

int ygzf_image_cache_reserve(ygzf_ctx *c, int n_slots, int w, int h) {
    if (!c) return YGZF_ERR_INVALID;
    if (n_slots < 1 || w < 1 || h < 1) return fail(c, YGZF_ERR_INVALID, "bad cache size");
    HIPCHECK(c, hipSetDevice(c->device));
    int rc = apply_geometry(c, w, h, 1);
    if (rc) return rc;
    const int pitch = align_up(w, 64);
    if ((rc = ensure(c, c->dCacheImg, (size_t) n_slots * pitch * h + 256)) || (rc = ensure(c, c->dCachePyr, (size_t) n_slots * c->geo.pyrBytes + 256))) return rc;
    c->cacheSlots = n_slots;
    c->cacheW = w;
    c->cacheH = h;
    c->cachePitch = pitch;
    c->cachePyrBytes = c->geo.pyrBytes;
    c->cacheFilled.assign(n_slots, 0);
    return YGZF_OK;
}

int ygzf_image_cache_put(ygzf_ctx *c, int slot, const uint8_t *img, int w, int h, int stride) {
    if (!c || !img) return fail(c, YGZF_ERR_INVALID, "null argument");
    if (c->cacheSlots <= 0) return fail(c, YGZF_ERR_STATE, "image cache not reserved");
    if (slot < 0 || slot >= c->cacheSlots) return fail(c, YGZF_ERR_INVALID, "slot %d outside 0..%d", slot, c->cacheSlots - 1);
    if (w != c->cacheW || h != c->cacheH || stride < w) return fail(c, YGZF_ERR_INVALID, "image %dx%d does not match the cache (%dx%d)", w, h, c->cacheW, c->cacheH);
    HIPCHECK(c, hipSetDevice(c->device));
    int rc = apply_geometry(c, w, h, 1);
    if (rc) return rc;
    FrameSet fs = cache_frameset(c);
    fs.img0 += (long long) slot * fs.img0_stride;      // the launchers address "frame 0" of the set they are given
    fs.pyr += (long long) slot * fs.pyr_stride;
    if ((rc = upload_rows(c, (void *) fs.img0, (size_t) c->cachePitch, img, (size_t) stride, w, (size_t) h))) return rc;
    if ((rc = pyramid_chain(c, fs, 1))) return rc;
    HIPCHECK(c, hipGetLastError());
    c->cacheFilled[slot] = 1;
    return YGZF_OK;
}

int ygzf_image_cache_put_resident(ygzf_ctx *c, int slot, ygzf_ctx *src) {
    if (!c || !src) return fail(c, YGZF_ERR_INVALID, "null argument");
    if (c->cacheSlots <= 0) return fail(c, YGZF_ERR_STATE, "image cache not reserved");
    if (slot < 0 || slot >= c->cacheSlots) return fail(c, YGZF_ERR_INVALID, "slot %d outside 0..%d", slot, c->cacheSlots - 1);
    if (src == c || src->device != c->device) return fail(c, YGZF_ERR_INVALID, "the source must be another context on the same device");
    if (!src->pyrHeld || src->pyrHeldW != c->cacheW || src->pyrHeldH != c->cacheH || src->geo.w != c->cacheW || src->geo.h != c->cacheH)
        return fail(c, YGZF_ERR_STATE, "the source context holds no %dx%d image with its pyramid", c->cacheW, c->cacheH);
    HIPCHECK(c, hipSetDevice(c->device));
    int rc = apply_geometry(c, c->cacheW, c->cacheH, 1);
    if (rc) return rc;
    const int L = c->tab.cfg.nlevels;
    if (src->tab.cfg.nlevels != L || src->geo.pyrBytes != c->geo.pyrBytes) return fail(c, YGZF_ERR_INVALID, "the two contexts' pyramids differ (levels / scale factor)");
    for (int l = 0; l < L; l++) {
        const LevelGeom &a = src->geo.lv[l], &b = c->geo.lv[l];
        if (a.w != b.w || a.h != b.h || a.pitch != b.pitch || a.off != b.off) return fail(c, YGZF_ERR_INVALID, "the two contexts' pyramids differ (level %d)", l);
    }
    const size_t imgBytes = (size_t) c->cachePitch * c->cacheH;   // both sides: pitch = width rounded up to 64
    if (!c->evShare) HIPCHECK(c, hipEventCreateWithFlags(&c->evShare, hipEventDisableTiming));
    // order: the source's pending work (its pyramid kernels) -> the copies on this context's stream -> the source's later work
    if (src->evPyrDoneValid) HIPCHECK(c, hipStreamWaitEvent(c->stream, src->evPyrDone, 0));   // (not the end of its stream: see mark_pyramid_done)
    else {
        HIPCHECK(c, hipEventRecord(c->evShare, src->stream));
        HIPCHECK(c, hipStreamWaitEvent(c->stream, c->evShare, 0));
    }
    HIPCHECK(c, hipMemcpyAsync((uint8_t *) c->dCacheImg.p + (size_t) slot * imgBytes, src->dImg0.p, imgBytes, hipMemcpyDeviceToDevice, c->stream));
    if (c->geo.pyrBytes > 0)
        HIPCHECK(c, hipMemcpyAsync((uint8_t *) c->dCachePyr.p + (size_t) slot * c->cachePyrBytes, src->dPyr.p, (size_t) c->geo.pyrBytes, hipMemcpyDeviceToDevice, c->stream));
    HIPCHECK(c, hipEventRecord(c->evShare, c->stream));
    HIPCHECK(c, hipStreamWaitEvent(src->stream, c->evShare, 0));
    c->cacheFilled[slot] = 1;
    return YGZF_OK;
}

int ygzf_has_resident_image(const ygzf_ctx *c, int w, int h) {
    return c && c->pyrHeld && c->pyrHeldW == w && c->pyrHeldH == h ? 1 : 0;
}

int ygzf_find_direct_projection_batch(ygzf_ctx *c, const ygzf_camera *cam, int cur_slot, const float *cur_Tcw7, int n, const int *ref_slot,
                                      const float *ref_Tcw7, const ygzf_kp *ref_kp, const float *mp_world, float *px_curr, int *search_level,
                                      uint8_t *success, uint8_t *patches_with_border) {
    if (!c || !cam || !cur_Tcw7) return fail(c, YGZF_ERR_INVALID, "null argument");
    if (n < 0) return fail(c, YGZF_ERR_INVALID, "negative count");
    if (n == 0) return YGZF_OK;
    if (!ref_slot || !ref_Tcw7 || !ref_kp || !mp_world || !px_curr || !search_level || !success) return fail(c, YGZF_ERR_INVALID, "null array");
    if (c->cacheSlots <= 0) return fail(c, YGZF_ERR_STATE, "image cache not reserved");
    if (c->geo.w != c->cacheW || c->geo.h != c->cacheH) {
        int rc0 = apply_geometry(c, c->cacheW, c->cacheH, 1);
        if (rc0) return rc0;
    }
    const int L = c->tab.cfg.nlevels;
    if (cur_slot < 0 || cur_slot >= c->cacheSlots || !c->cacheFilled[cur_slot]) return fail(c, YGZF_ERR_INVALID, "current-frame slot %d is empty", cur_slot);
    for (int i = 0; i < n; i++) {
        if (ref_slot[i] < 0 || ref_slot[i] >= c->cacheSlots || !c->cacheFilled[ref_slot[i]]) return fail(c, YGZF_ERR_INVALID, "candidate %d: slot %d is empty", i, ref_slot[i]);
        if (ref_kp[i].octave < 0 || ref_kp[i].octave >= L) return fail(c, YGZF_ERR_INVALID, "candidate %d: octave out of range", i);
    }
    HIPCHECK(c, hipSetDevice(c->device));
    ygzf_ctx::Buf *D = c->dDir;
    struct Up { ygzf_ctx::Buf *b; const void *src; size_t bytes; };
    Up ups[] = {{&D[0], ref_slot, 4 * (size_t) n}, {&D[1], ref_Tcw7, 28 * (size_t) n}, {&D[2], ref_kp, sizeof(ygzf_kp) * (size_t) n},
                {&D[3], mp_world, 12 * (size_t) n}, {&D[4], px_curr, 8 * (size_t) n}};
    int rc;
    for (auto &u : ups) {
        if ((rc = ensure(c, *u.b, u.bytes))) return rc;
        HIPCHECK(c, hipMemcpyAsync(u.b->p, u.src, u.bytes, hipMemcpyHostToDevice, c->stream));
    }
    if ((rc = ensure(c, D[5], 4 * (size_t) n)) || (rc = ensure(c, D[6], (size_t) n)) || (patches_with_border && (rc = ensure(c, D[7], 100 * (size_t) n)))) return rc;
    DirectArgs A;
    memset(&A, 0, sizeof A);
    A.cache = cache_frameset(c);
    A.geom = (const LevelGeom *) c->dGeom.p;
    A.nlevels = L;
    A.curSlot = cur_slot;
    memcpy(A.curTcw, cur_Tcw7, 28);
    A.fx = cam->fx; A.fy = cam->fy; A.cx = cam->cx; A.cy = cam->cy;
    for (int l = 0; l < kMaxLevels; l++) {
        A.scale[l] = l < L ? c->tab.scale[l] : 1.f;
        A.invScale[l] = l < L ? c->tab.invScale[l] : 1.f;
    }
    A.invLevelSigma2_1 = c->tab.invSigma2[L > 1 ? 1 : 0];
    A.n = n;
    A.refSlot = (const int *) D[0].p;
    A.refTcw7 = (const float *) D[1].p;
    A.refKp = (const ygzf_kp *) D[2].p;
    A.mpWorld = (const float *) D[3].p;
    A.pxCurr = (float *) D[4].p;
    A.searchLevel = (int *) D[5].p;
    A.success = (uint8_t *) D[6].p;
    A.patches = patches_with_border ? (uint8_t *) D[7].p : nullptr;
    {
        ProfScope ps(c, KK_DIRECT);
        launch_direct_projection(c->stream, A);
    }
    HIPCHECK(c, hipGetLastError());
    HIPCHECK(c, hipMemcpyAsync(px_curr, D[4].p, 8 * (size_t) n, hipMemcpyDeviceToHost, c->stream));
    HIPCHECK(c, hipMemcpyAsync(search_level, D[5].p, 4 * (size_t) n, hipMemcpyDeviceToHost, c->stream));
    HIPCHECK(c, hipMemcpyAsync(success, D[6].p, (size_t) n, hipMemcpyDeviceToHost, c->stream));
    if (patches_with_border) HIPCHECK(c, hipMemcpyAsync(patches_with_border, D[7].p, 100 * (size_t) n, hipMemcpyDeviceToHost, c->stream));
    HIPCHECK(c, hipStreamSynchronize(c->stream));
    return YGZF_OK;
}

// ---- SparseImgAlign over a resident batch ---------------------------------------------------------------------------------
int ygzf_align_batch_prev(ygzf_ctx *c, const ygzf_camera *cam, int max_level, int min_level, int n_iter) {
    if (!c || !cam) return fail(c, YGZF_ERR_INVALID, "null argument");
    if (c->lastFrames < 1) return fail(c, YGZF_ERR_STATE, "no extracted batch");
    const int L = c->tab.cfg.nlevels;
    if (min_level < 1 || max_level < min_level || max_level >= L)
        return fail(c, YGZF_ERR_INVALID, "level range [%d,%d] (the resident form aligns on pyramid levels >= 1, as Tracking does)", min_level, max_level);
    HIPCHECK(c, hipSetDevice(c->device));
    const Geometry &G = c->geo;
    const int B = c->lastFrames;
    if (G.kpStride == 0) return fail(c, YGZF_ERR_STATE, "configuration yields no keypoints");
    c->alignCarry = true;
    int rc;
    ygzf_ctx::Buf *S = c->dAl;   // 0 level tables, 1 poses, 2 caches, 3 out, (world = dWorld)
    if ((rc = ensure(c, c->dWorld, (size_t) (B + 1) * G.kpStride * 3 * sizeof(float))) ||
        (rc = ensure(c, S[0], (size_t) B * 2 * kMaxLevels * sizeof(SiaLevel))) || (rc = ensure(c, S[1], (size_t) B * 14 * sizeof(float))) ||
        (rc = ensure(c, S[2], c->siaPerLevel && B <= kSiaPerLevelPairs ? (size_t) (max_level - min_level + 1) * B * G.kpStride * (52 * sizeof(float) + 1) + (size_t) B * G.kpStride + 64
                                              : (size_t) B * G.kpStride * ((16 + 96) * sizeof(float) + 1) + 64)) ||
        (rc = ensure(c, S[3], (size_t) B * 48 * sizeof(float))) || (rc = ensure(c, c->dCarryPyr, (size_t) G.pyrBytes + 256)))
        return rc;
    // level tables + identity poses: uploaded when anything they depend on changed
    std::vector<unsigned char> key;
    {
        const void *parts[] = {c->dPyr.p, c->dCarryPyr.p, S[0].p, S[1].p};
        key.insert(key.end(), (const unsigned char *) parts, (const unsigned char *) parts + sizeof parts);
        const int ints[] = {B, G.w, G.h, c->carryPyrValid ? 1 : 0};
        key.insert(key.end(), (const unsigned char *) ints, (const unsigned char *) ints + sizeof ints);
    }
    if (key != c->alKey) {
        std::vector<SiaLevel> lv((size_t) B * 2 * kMaxLevels);
        memset(lv.data(), 0, lv.size() * sizeof(SiaLevel));
        for (int p = 0; p < B; p++)
            for (int l = 1; l < L; l++) {
                const LevelGeom &g = G.lv[l];
                SiaLevel &r = lv[((size_t) p * 2 + 0) * kMaxLevels + l], &cu = lv[((size_t) p * 2 + 1) * kMaxLevels + l];
                r.w = cu.w = g.w; r.h = cu.h = g.h; r.pitch = cu.pitch = g.pitch;
                cu.img = (const uint8_t *) c->dPyr.p + (size_t) p * G.pyrBytes + g.off;
                r.img = p > 0 ? (const uint8_t *) c->dPyr.p + (size_t) (p - 1) * G.pyrBytes + g.off : (const uint8_t *) c->dCarryPyr.p + g.off;
            }
        std::vector<float> poses((size_t) B * 14, 0.f);
        for (int p = 0; p < B; p++) poses[(size_t) p * 14 + 3] = poses[(size_t) p * 14 + 10] = 1.f;   // identity quaternions
        HIPCHECK(c, hipMemcpyAsync(S[0].p, lv.data(), lv.size() * sizeof(SiaLevel), hipMemcpyHostToDevice, c->stream));
        HIPCHECK(c, hipMemcpyAsync(S[1].p, poses.data(), poses.size() * sizeof(float), hipMemcpyHostToDevice, c->stream));
        HIPCHECK(c, hipStreamSynchronize(c->stream));
        c->alKey = key;
    }
    const ygzf_kp *kp = (const ygzf_kp *) c->dOutKp.p;
    const int *cnt = (const int *) c->dOutCnt.p;
    SiaArgs A;
    memset(&A, 0, sizeof A);
    A.keys = kp;                 // pair p: reference = output slot p (slot 0 = carry), current = slot p + 1
    A.world = (const float *) c->dWorld.p;
    A.unitWorld = 1;             // MapPoints at unit depth along the keypoints' rays, computed where they are used
    A.kpStride = G.kpStride;
    A.nRef = cnt;
    A.poses = (const float *) S[1].p;
    A.refLv = (const SiaLevel *) S[0].p;
    A.curLv = A.refLv + kMaxLevels;
    A.lvStride = 2 * kMaxLevels;
    for (int l = 0; l < kMaxLevels; l++) A.invScale[l] = l < L ? c->tab.invScale[l] : 1.f;
    A.fx = cam->fx; A.fy = cam->fy; A.cx = cam->cx; A.cy = cam->cy;
    A.maxLevel = max_level; A.minLevel = min_level; A.nIter = n_iter;
    A.eps = 0.000001f;
    A.patchCache = (float *) S[2].p;
    A.jacCache = nullptr;
    A.visible = (uint8_t *) (A.patchCache + (size_t) B * G.kpStride * 48);
    A.momCache = A.patchCache + (size_t) B * G.kpStride * 64;   // (the buffer holds 112 floats per keypoint slot)
    if (c->siaPerLevel && B <= kSiaPerLevelPairs) {   // [patch rows: nLv x B x kpStride x 48 floats | moments: nLv x B x kpStride x 4 | flags: nLv x B x kpStride bytes | visible]
        const size_t nLv = (size_t) (max_level - min_level + 1), slots = (size_t) B * G.kpStride;
        A.perLevel = 1;
        A.pcLevelStride = slots * 48;
        A.momLevelStride = slots * 4;
        A.flagLevelStride = slots;
        A.momCache = A.patchCache + nLv * slots * 48;
        A.levelFlags = (uint8_t *) (A.momCache + nLv * slots * 4);
        A.visible = A.levelFlags + nLv * slots;
    }
    A.out = (float *) S[3].p;
    const int first = c->carryPyrValid ? 0 : 1;   // without a carried pyramid frame 0 has no reference image
    if (!c->carryPyrValid) HIPCHECK(c, hipMemsetAsync(S[3].p, 0, 48 * sizeof(float), c->stream));
    if (B - first > 0) {
        SiaArgs A2 = A;
        A2.keys += (size_t) first * G.kpStride;
        A2.world += (size_t) first * G.kpStride * 3;
        A2.nRef += first;
        A2.poses += (size_t) first * 14;
        A2.refLv += (size_t) first * A.lvStride;
        A2.curLv += (size_t) first * A.lvStride;
        A2.patchCache += (size_t) first * G.kpStride * 48;
        A2.visible += (size_t) first * G.kpStride;
        A2.momCache += (size_t) first * G.kpStride * 4;
        if (A2.perLevel) A2.levelFlags += (size_t) first * G.kpStride;
        A2.out += (size_t) first * 48;
        size_t sl = sia_lds_bytes(G.kpStride);
        A2.ldsFeat = G.kpStride;
        A2.jacLds = sia_jac_in_lds(G.kpStride) ? 1 : 0;
        if (sl > 150 * 1024) return fail(c, YGZF_ERR_UNSUPPORTED, "SparseImgAlign supports at most %d features", (int) (150 * 1024 / 24));
        {
            size_t largest = 0;
            for (int l = min_level; l <= max_level; l++) largest = std::max(largest, (size_t) G.lv[l].pitch * G.lv[l].h);
            A2.stageOff = (int) ((sl + 15) & ~(size_t) 15);
            A2.stageBytes = (int) sia_stage_bytes((size_t) A2.stageOff, largest);
            // Many pairs in flight: the workgroup keeps to 74 KB of LDS so that TWO share a CU -- one pair's solve (a single wave) and barriers
            // then overlap the other's accumulate; the coarse levels that no longer fit the staging area are gathered from L2 instead
            // (measured on 256-pair launches from three streams: 141.1 -> 147.2 k frames/s; a lone pair keeps the full staging area).
            static const long capKb = getenv("YGZF_SIA_LDS_CAP") ? atol(getenv("YGZF_SIA_LDS_CAP")) : 74;   // A/B runs (0: no cap)
            if (B - first >= 128 && capKb > 0) {
                const long cap = capKb * 1024 - A2.stageOff;
                A2.stageBytes = cap > 4096 ? (int) std::min<long>(A2.stageBytes, cap & ~15L) : 0;
            }
            sl = (size_t) A2.stageOff + (size_t) A2.stageBytes;
        }
        HIPCHECK(c, sia_prepare(sl));
        ProfScope ps(c, KK_SIA);
        if (A2.perLevel) launch_sia_precompute(c->stream, A2, B - first, G.kpStride);
        launch_sia(c->stream, A2, B - first, sl);
    }
    HIPCHECK(c, hipGetLastError());
    c->lastAlignPairs = B;
    return YGZF_OK;
}

int ygzf_align_fetch(ygzf_ctx *c, int frame, float *TCR_out, size_t *ret, float *info) {
    if (!c || !TCR_out || !ret) return fail(c, YGZF_ERR_INVALID, "null argument");
    if (c->lastAlignPairs < 1) return fail(c, YGZF_ERR_STATE, "no aligned batch");
    if (frame < 0 || frame >= c->lastAlignPairs) return fail(c, YGZF_ERR_INVALID, "frame %d out of range", frame);
    float out[48];
    HIPCHECK(c, hipMemcpyAsync(out, (float *) c->dAl[3].p + (size_t) frame * 48, sizeof out, hipMemcpyDeviceToHost, c->stream));
    HIPCHECK(c, hipStreamSynchronize(c->stream));
    memcpy(TCR_out, out, 28);
    *ret = (size_t) out[7];
    if (info) { info[0] = out[8]; info[1] = out[9]; }
    return YGZF_OK;
}

}  // extern "C"
