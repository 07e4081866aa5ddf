// ygzf_internal.h -- shared host/device definitions of libygzf (product code; never includes oracle/).
#ifndef YGZF_INTERNAL_H
#define YGZF_INTERNAL_H
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

#include "../../include/ygzf.h"

namespace ygzf {

constexpr int kPatchSize = 31;        // reference src/ORBextractor.cc:73
constexpr int kHalfPatch = 15;        // :74
constexpr int kEdgeThreshold = 19;    // :75
constexpr int kBorder = kEdgeThreshold - 3;  // minBorderX of ComputeKeyPointsOctTree (:733)
constexpr int kMaxLevels = 16;
constexpr int kMaxCellWin = 66;       // window side of one FAST cell: wCell(<60)+6
constexpr int kFastBlock = 256;   // 4 waves (cell positions) per workgroup; single-wave workgroups measured slower: 739 vs 625 us per 256 frames
constexpr int kOctBlock = 1024;   // threads per (level, frame) workgroup of k_octree (512 / 256 measured: no schedule effect, profiles/r05_f_octree_block_sweep.txt)
// The dynamic-LDS ceiling of a kernel is a per-process attribute of the function: it is always set to the same value (the CU's 160 KB
// minus room for static LDS), never to a per-call size, so that contexts used from different threads cannot lower it under each other.
constexpr int kMaxDynLds = 160 * 1024 - 2048;

// Geometry of one pyramid level; one table per (w,h) configuration, uploaded to the device.
struct LevelGeom {
    int w, h, pitch;          // level image; pitch in bytes (64-aligned) for levels >= 1
    long long off;            // byte offset of the level inside one frame's pyramid slab (levels >= 1)
    int area2x;               // 1: previous level is exactly 2x in both axes -> 2x2 area average (cv::resize quirk)
    int tiledOk;              // 1: every output tile's source region fits the LDS staging area of k_pyr_resize_tiled
    int xtab, ytab;           // offsets into the resize coefficient tables
    int pyrCol, pyrRow;       // first PyrColRec / PyrRowRec of this level
    int pyrTile, nPyrTiles;   // this level's PyrTileRec range: one workgroup of k_pyr_resize_tiled each
    // FAST cell grid, src/ORBextractor.cc:733-745
    int nCols, nRows, wCell, hCell;
    int maxBorderX, maxBorderY;
    int cellBase;             // first cell of this level inside a frame's cell arrays
    int groupBase;            // first 2x2 cell group (one FAST workgroup) of this level inside a frame
    int slotCap;              // candidate slots per cell = ceil(wCell/2)*ceil(hCell/2)
    long long slotBase;       // first slot of this level inside a frame's slot array
    // octree, :533-563
    int regW, regH;           // maxBorderX-minBorderX, maxBorderY-minBorderY
    int nIni;
    float hX;
    int depth;                // D: subdivisions encoded in the path key
    int keyBits;              // root bits + 2*D
    int nFeat;                // mnFeaturesPerLevel[level]
    int kpCap, kpBase;        // capacity / base of this level inside a frame's level-keypoint arrays
    long long candBase;       // base inside a frame's candidate scratch arrays
    int candCap;              // nCells*slotCap
    float scale;              // mvScaleFactor[level]
    float kpSize;             // (float)(int)(PATCH_SIZE*scale)
};

// k_pyr_resize_tiled's host-built tables (cv::resize INTER_LINEAR, the same integers as xofs / xalpha / yofs / ybeta):
// what a lane needs for the four output columns it owns, what a row needs, and the list of tiles of a level.
struct PyrColRec {          // per group of 4 output columns, 48 bytes = three 16-byte loads
    int sx0;                // source column of the group's first pixel
    unsigned pad[3];
    unsigned sel[4];        // v_perm_b32 selector of column k: (left, right) source pixel as bytes 0 and 2, relative to sx0
    unsigned ap[4];         // alpha0 | alpha1 << 16
};
struct PyrRowRec {          // per output row, 16 bytes
    unsigned r01;           // source row of the upper tap | lower tap << 16, both clamped to the source image
    unsigned b0, b1;        // beta << 12: v_mul_hi_u32(H & ~15, beta << 12) == (beta * (H >> 4)) >> 16
    unsigned pad;
};
struct PyrTileRec {         // 8 bytes
    unsigned short x0, y0;  // first output column / row
    unsigned short sxa;     // first staged source column (multiple of 4)
    unsigned char nc;       // 16-byte chunks staged per source row
    unsigned char foldLog2; // a wave's lanes cover 2^foldLog2 rows x (256 >> foldLog2) columns per pass
};

// One FAST cell as k_fast_tab wants it (host-built per (w, h) configuration, indexed [cell group * 4 + position in the 2x2 group]):
// everything ComputeKeyPointsOctTree's cell loop derives per cell (src/ORBextractor.cc:747-764), so that a wave starts from ONE 32-byte
// scalar load instead of a level search, a geometry load, an integer division and two dozen scalar operations.
struct FastCellRec {
    unsigned xy;        // (iniX - 1) | iniY << 16: first byte / row of the window inside the level image
    unsigned geo;       // pitch (levels >= 1) | dw << 16 | dh << 24      dw x dh = tested pixels of the cell
    unsigned flags;     // wh (window rows) | kFastCell* << 8 | level << 16
    unsigned cell;      // index inside a frame's per-cell arrays (LevelGeom::cellBase + row-major cell)
    unsigned slot;      // first candidate slot of the cell inside a frame's slot array
    unsigned off;       // byte offset of the level inside a frame's pyramid slab (levels >= 1)
    unsigned pad0, pad1;
};
constexpr unsigned kFastCellExists = 1, kFastCellRun = 2;   // no bit: no cell at this position of the group; Exists only: skipped by the border rules (:751, :759)

struct FrameSet {
    const uint8_t *img0;      // level 0 of frame 0
    long long img0_stride;    // bytes between frames
    int img0_pitch;
    uint8_t *pyr;             // pyramid slab (levels >= 1) of frame 0
    long long pyr_stride;
};

__host__ __device__ inline const uint8_t *level_ptr(const FrameSet &fs, const LevelGeom &g, int level, int frame, int *pitch) {
    if (level == 0) {
        *pitch = fs.img0_pitch;
        return fs.img0 + (long long) frame * fs.img0_stride;
    }
    *pitch = g.pitch;
    return fs.pyr + (long long) frame * fs.pyr_stride + g.off;
}

// Host-side tables of one extractor configuration (product restatement of ORBextractor's ctor).
struct Tables {
    ygzf_extractor_cfg cfg;
    std::vector<float> scale, invScale, sigma2, invSigma2;
    std::vector<int> nFeat;
    int umax[16];
    void init(const ygzf_extractor_cfg &c);
    void levelSize(int w, int h, int level, int *lw, int *lh) const;
};

int cv_round_host(double v);

// ---- the library's environment switches, all of them -------------------------------------------------------------------------------
// Two variables, read when a context (or a multi-GPU handle) is created, never on the launch path.  Neither changes a result.
//   YGZF_DEBUG=flag,flag,...   sync     synchronise after every kernel and name it on stderr
//                              nograph  launch the pyramid chain kernel by kernel instead of replaying its hipGraph
//                              oct / match / sia   phase clocks of k_octree / k_match_last / k_sia_run printed to stderr (tools/*_phases.py)
//   YGZF_FORCE=key=value,...   pins a plan that the library otherwise chooses itself from the shape of the launch, so that the tests can run EVERY
//                              plan of a kernel against the oracle on small inputs (INTEGRATION.md lists the keys); a key that is absent leaves the
//                              library's own choice in place.
// (host/TrackingBatched.cc reads a third one, YGZF_FRUSTUM_DEVICE_MIN, documented there.)
inline const char *env_find(const char *var, const char *key, size_t *len) {   // value of `key` inside the comma list $var; flags have an empty value
    const char *e = getenv(var);
    const size_t kl = strlen(key);
    while (e && *e) {
        const char *end = strchr(e, ',');
        const size_t n = end ? (size_t) (end - e) : strlen(e);
        if (n >= kl && !strncmp(e, key, kl) && (n == kl || e[kl] == '=')) {
            *len = n == kl ? 0 : n - kl - 1;
            return n == kl ? e + kl : e + kl + 1;
        }
        e = end ? end + 1 : nullptr;
    }
    return nullptr;
}
inline bool debug_flag(const char *flag) {
    size_t n;
    return env_find("YGZF_DEBUG", flag, &n) != nullptr;
}
inline long forced(const char *key, long dflt) {
    size_t n;
    const char *v = env_find("YGZF_FORCE", key, &n);
    return v && n ? atol(v) : dflt;
}
inline bool forced_is(const char *key, const char *value) {
    size_t n;
    const char *v = env_find("YGZF_FORCE", key, &n);
    return v && n == strlen(value) && !strncmp(v, value, n);
}

}  // namespace ygzf
#endif
