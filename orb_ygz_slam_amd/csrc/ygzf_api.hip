// ygzf_api.hip -- the C ABI of libygzf (include/ygzf.h): context, geometry tables, buffer management and the launch
// sequence of the extractor hot path.  Product code: no CPU fallback, nothing from oracle/ is included or linked.
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <mutex>

#include "ygzf_ctx.h"

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

}  // namespace ygzf

using namespace ygzf;

thread_local std::string g_create_err;   // last failed ygzf_create of THIS thread

// Per-(w,h) geometry: level sizes, resize coefficient tables (cv::resize INTER_LINEAR fixed point, restated from the
// OpenCV 2.4/3.2 algorithm: fx = (float)((dx+0.5)*scale - 0.5), 11-bit coefficients), FAST cell grid (:733-745),
// octree root layout (:537-557) and buffer offsets.
// Row ranges of k_pyr_strips (extract_kernels.hip): the last level is cut into S strips; walking up, a strip needs of level l the source
// rows of the rows it produces of level l + 1 (yofs is monotone: first row's sy .. last row's sy + 1, clamped as the kernel clamps) and
// owns the S-th part of every level.  S grows until the two LDS regions (even / odd levels) fit.
static bool plan_pyr_strips_from(Geometry &G, int L, int base, bool only32) {
    if (L - base < 3) return false;
    if (G.lv[base].h >= 65536) return false;   // the strip plan packs row ranges into 16-bit fields: taller images keep one launch per level
    for (int l = base + 1; l < L; l++)
        if (G.lv[l].area2x || !G.lv[l].tiledOk || (G.lv[l].w + 3) / 4 > kPyrStripMaxThreads) return false;
    // 752x480, one frame: 16 strips 148 us per resident extraction, 24 146, 32 144, 48 143 (the halo rows grow with the strip count: 1.23 x
    // the pixels of the pyramid at 16); eight frames: 16 strips 218 us, 32 215
    // (more strips when the LDS regions do not fit, fewer for images whose last level has fewer than 64 rows).
    const int candidates[] = {32, 48, 64, 16, 8};
    const int minS = (int) forced("pyr_strips", 0);   // at least this many strips (tests: the strip plans of large images on small ones)
    for (int S : candidates) {
        if (S < minS) continue;
        if (only32 && S != 32) continue;
        if (G.lv[L - 1].h < 2 * S) continue;
        std::vector<PyrStripPlan> plan(S);
        size_t bytes[2] = {0, 0};
        int maxRows = 0;
        for (int s = 0; s < S; s++) {
            int rows = 0, pca = 0, pcb = 0;   // the range produced of the level below the one in hand
            PyrStripPlan &p = plan[s];
            std::memset(&p, 0, sizeof p);
            for (int l = L - 1; l >= base; l--) {
                const int hl = G.lv[l].h;
                int wa = (int) ((long long) hl * s / S), wb = (int) ((long long) hl * (s + 1) / S), ca = wa, cb = wb;
                if (l < L - 1) {
                    const LevelGeom &n = G.lv[l + 1];
                    const int na = std::min(std::max(G.yofs[n.ytab + pca], 0), hl - 1);
                    const int nb = std::min(std::max(G.yofs[n.ytab + pcb - 1] + 1, 0), hl - 1) + 1;
                    if (l == base) { ca = na; cb = nb; wa = wb = 0; }   // the base level is staged, not produced
                    else { ca = std::min(wa, na); cb = std::max(wb, nb); }
                }
                p.lv[l] = make_uint2((unsigned) ca | ((unsigned) cb << 16), (unsigned) wa | ((unsigned) wb << 16));
                pca = ca; pcb = cb;
                bytes[(l - base) & 1] = std::max(bytes[(l - base) & 1], (size_t) (cb - ca) * pyr_strip_lds_pitch(G.lv[l].w));
                if (l > base) rows += cb - ca;
            }
            maxRows = std::max(maxRows, rows);
        }
        if (maxRows > kPyrStripMaxThreads) continue;   // one thread per row fills the row table
        const size_t rowBytes = ((size_t) maxRows * 8 + 15) & ~(size_t) 15, head = rowBytes;   // (the column coefficients stay in global memory: k_pyr_strips)
        const size_t a = (bytes[0] + 16 + 15) & ~(size_t) 15, b = (bytes[1] + 16 + 15) & ~(size_t) 15;   // hrow reads up to 11 bytes past a row's pixels
        if (head + a + b > 160 * 1024) continue;
        G.pyrPlan = plan;
        G.pyrBase = base;
        G.pyrStripOffA = (int) head;
        G.pyrStripOffB = (int) (head + a);
        G.pyrStripLds = head + a + b;
        std::memset(G.pyrLevels, 0, sizeof G.pyrLevels);
        for (int l = 0; l < L; l++) {
            const LevelGeom &g = G.lv[l];
            G.pyrLevels[l] = PyrStripLevel{g.w, g.h, g.pitch, g.xtab, g.ytab, 0, (long long) g.off};
        }
        return true;
    }
    return false;
}

// The whole chain from the image when that fits LDS; otherwise (3840x2160: a strip of the last level would need 270 KB of level 0) the first levels
// come from k_pyr_resize_tiled, one launch each, and the strips start from the first level whose chain fits -- a one-frame call then pays 3-4
// launches instead of 11.  YGZF_FORCE=pyr_strip_base=s: not below level s (tests).
static void plan_pyr_strips(Geometry &G, int L) {
    G.pyrPlan.clear();
    G.pyrBase = 0;
    // (a 3840x2160 pair, same box: from level 3 with 48 strips 0.774 ms -- and two clusters of calls, 0.72 and 0.78 --, from level 4 with 32 strips 0.710,
    // from level 5 / 6 / 9 0.76 / 0.75 / 0.75: above the image the plan of 32 strips is preferred to an earlier level with more)
    const int first = (int) forced("pyr_strip_base", 0);
    if (first == 0 && plan_pyr_strips_from(G, L, 0, false)) return;
    for (int base = std::max(first, 1); base + 3 <= L; base++)
        if (plan_pyr_strips_from(G, L, base, true)) return;
    for (int base = std::max(first, 1); base + 3 <= L; base++)
        if (plan_pyr_strips_from(G, L, base, false)) return;
}

int build_geometry(ygzf_ctx *c, int w, int h, Geometry &G) {
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
            // k_pyr_resize_tiled's tables.  A level is cut into tiles of 8192 output pixels: 256-column tiles of 32 rows while >= 225 columns
            // remain, then the rest in binary pieces of 128 / 64 / 32 columns whose waves fold 2 / 4 / 8 rows into one pass.  True for the
            // usual scale factors (<= ~1.27): steeper pyramids (a tile's source region would not fit the staging area) take the per-pixel kernel.
            g.tiledOk = 1;
            g.pyrCol = (int) G.pyrCols.size();
            g.pyrRow = (int) G.pyrRows.size();
            g.pyrTile = (int) G.pyrTiles.size();
            for (int xb = 0; xb < g.w; xb += 4) {
                PyrColRec r;
                memset(&r, 0, sizeof r);
                r.sx0 = G.xofs[g.xtab + xb];
                for (int k = 0; k < 4; k++) {
                    const int x = std::min(xb + k, g.w - 1);
                    const int sx = G.xofs[g.xtab + x], sx1 = std::min(sx + 1, s.w - 1);
                    if (sx1 - r.sx0 > 7 || sx < r.sx0) g.tiledOk = 0;   // the pair of every column comes out of the eight bytes from sx0 on
                    r.sel[k] = (unsigned) ((sx - r.sx0) & 7) | 0x0C00u | ((unsigned) ((sx1 - r.sx0) & 7) << 16) | 0x0C000000u;
                    r.ap[k] = (unsigned) (unsigned short) G.xalpha[2 * (g.xtab + x)] | ((unsigned) (unsigned short) G.xalpha[2 * (g.xtab + x) + 1] << 16);
                    if (G.xalpha[2 * (g.xtab + x)] < 0 || G.xalpha[2 * (g.xtab + x) + 1] < 0) g.tiledOk = 0;
                }
                G.pyrCols.push_back(r);
            }
            for (int y = 0; y < g.h; y++) {
                const int sy = G.yofs[g.ytab + y];
                const int r0 = std::min(std::max(sy, 0), s.h - 1), r1 = std::min(std::max(sy + 1, 0), s.h - 1);
                const int b0 = G.ybeta[2 * (g.ytab + y)], b1 = G.ybeta[2 * (g.ytab + y) + 1];
                if (b0 < 0 || b1 < 0 || b0 > 2048 || b1 > 2048 || s.h > 65535) g.tiledOk = 0;
                G.pyrRows.push_back(PyrRowRec{(unsigned) r0 | ((unsigned) r1 << 16), (unsigned) b0 << 12, (unsigned) b1 << 12, 0u});
            }
            for (int p = 1; p < 8; p++) G.pyrRows.push_back(G.pyrRows.back());   // a wave reads the records of its 8 rows in one go
            for (int x0 = 0; x0 < g.w && g.tiledOk;) {
                const int rest = g.w - x0;
                int fl = 0;   // fold: the widest piece of the binary decomposition of the rest, in units of 32 columns
                if (rest <= 224) {
                    const int units = (rest + 31) / 32;
                    fl = units >= 4 ? 1 : units >= 2 ? 2 : 3;
                }
                const int tw = 256 >> fl, th = kPyrTileRows << fl;
                const int xl = std::min(x0 + tw, g.w) - 1;
                const int sxa = G.xofs[g.xtab + x0] & ~3, sxb = std::min(G.xofs[g.xtab + xl] + 1, s.w - 1);
                const int nc = (sxb - sxa) / 16 + 1;
                if (nc > pyr_fold_chunks(fl) || sxa > 65535) g.tiledOk = 0;
                for (int y0 = 0; y0 < g.h && g.tiledOk; y0 += th) {
                    const int yl = std::min(y0 + th, g.h) - 1;
                    const int sya = (int) (G.pyrRows[g.pyrRow + y0].r01 & 0xffffu), syb = (int) (G.pyrRows[g.pyrRow + yl].r01 >> 16);
                    if (syb - sya + 1 > pyr_fold_rows(fl) || syb < sya) g.tiledOk = 0;
                    G.pyrTiles.push_back(PyrTileRec{(unsigned short) x0, (unsigned short) y0, (unsigned short) sxa, (unsigned char) nc, (unsigned char) fl});
                }
                x0 += tw;
            }
            g.nPyrTiles = (int) G.pyrTiles.size() - g.pyrTile;
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
        G.kpCapMax = std::max(G.kpCapMax, (g.kpCap + 3) & ~3);   // (a multiple of 4: k_octree's per-node arrays stay 16-byte aligned)
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

int apply_geometry(ygzf_ctx *c, int w, int h, int nFrames) {
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
        {
            struct { ygzf_ctx::Buf *b; const void *src; size_t bytes; } up2[3] = {
                {&c->dPyrCols, G.pyrCols.data(), G.pyrCols.size() * sizeof(PyrColRec)},
                {&c->dPyrRows, G.pyrRows.data(), G.pyrRows.size() * sizeof(PyrRowRec)},
                {&c->dPyrTiles, G.pyrTiles.data(), G.pyrTiles.size() * sizeof(PyrTileRec)}};
            for (auto &u : up2) {
                rc = ensure(c, *u.b, std::max<size_t>(u.bytes, 64));
                if (rc) return rc;
                if (u.bytes) HIPCHECK(c, hipMemcpy(u.b->p, u.src, u.bytes, hipMemcpyHostToDevice));
            }
        }
        if (!G.pyrPlan.empty()) {
            rc = ensure(c, c->dPyrPlan, sizeof G.pyrLevels + G.pyrPlan.size() * sizeof(PyrStripPlan));   // level table, then one plan per strip
            if (rc) return rc;
            {   // the kernel counts levels from the strips' base level: both tables go up shifted
                PyrStripLevel lv[kMaxLevels];
                std::memset(lv, 0, sizeof lv);
                for (int l = G.pyrBase; l < L; l++) lv[l - G.pyrBase] = G.pyrLevels[l];
                std::vector<PyrStripPlan> rel(G.pyrPlan.size());
                for (size_t q = 0; q < rel.size(); q++) {
                    std::memset(&rel[q], 0, sizeof rel[q]);
                    for (int l = G.pyrBase; l < L; l++) rel[q].lv[l - G.pyrBase] = G.pyrPlan[q].lv[l];
                }
                HIPCHECK(c, hipMemcpy(c->dPyrPlan.p, lv, sizeof lv, hipMemcpyHostToDevice));
                HIPCHECK(c, hipMemcpy((char *) c->dPyrPlan.p + sizeof G.pyrLevels, rel.data(), rel.size() * sizeof(PyrStripPlan), hipMemcpyHostToDevice));
            }
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
            const size_t budget = (size_t) 71 * 1024;   // two workgroups per CU beside 9 KB of static LDS each (radix histogram); 48 / 36 / 24 KB measured slower (profiles/r05_f_octree_lds_and_streams.txt)
            c->octLdsCand = fixed + 16 * 256 < budget ? (int) ((budget - fixed) / 16) : 0;
            if (c->octLdsCand > 8192) c->octLdsCand = 8192;
        }
        c->octLds = octree_lds_bytes(G.maxCellsPerLevel, G.kpCapMax, c->octLdsCand, c->octGlobalNodes);
        // Levels too large for LDS-resident candidate buffers (1920x1080 / 4000 and up) take the histogram plan: no sort, tree passes in LDS
        // (extract_kernels.hip, k_octree<.., kHist>).  Consecutive levels are launched together while their list caps stay within a factor of two of
        // the group's first level, so that the small levels do not reserve the LDS of the large ones.
        c->octGroups.clear();
        // (YGZF_FORCE=oct_plan=hist / sort pins one plan, oct_hist_bins bounds the histogram: the tests run every geometry through both plans
        // and through the fall-back from one to the other)
        const bool wantHist = forced_is("oct_plan", "hist") ? true : forced_is("oct_plan", "sort") ? false : (c->octGlobalNodes || c->octLdsCand == 0);
        const int binsEnv = (int) forced("oct_hist_bins", 0);
        if (wantHist) {
            bool ok = true;
            for (int l = 0; l < L && ok;) {
                ygzf_ctx::OctGroup grp;
                grp.l0 = l;
                int cells = 0, tabs = 0;
                for (; l < L && (grp.n == 0 || 2 * G.lv[l].kpCap > G.lv[grp.l0].kpCap); l++, grp.n++) {
                    const LevelGeom &g = G.lv[l];
                    const int nc = g.nCols * g.nRows;
                    grp.cap = std::max(grp.cap, (g.kpCap + 3) & ~3);
                    cells = std::max(cells, nc + 1);
                    if (nc > 0)
                        tabs = std::max(tabs, nc + 1 + std::max(g.regW, g.nCols * g.wCell) + 9 + std::max(g.regH, g.nRows * g.hCell) + 9 + nc);
                }
                if (grp.cap < 4) grp.cap = 4;
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
        if (!forced_is("oct_plan", "sort")) {
            ygzf_ctx::OctGroup grp;
            grp.l0 = 0;
            grp.n = L;
            int cells = 0, tabs = 0;
            for (int l = 0; l < L; l++) {
                const LevelGeom &g = G.lv[l];
                const int nc = g.nCols * g.nRows;
                grp.cap = std::max(grp.cap, (g.kpCap + 3) & ~3);
                cells = std::max(cells, nc + 1);
                if (nc > 0) tabs = std::max(tabs, nc + 1 + std::max(g.regW, g.nCols * g.wCell) + 9 + std::max(g.regH, g.nRows * g.hCell) + 9 + nc);
            }
            if (grp.cap < 4) grp.cap = 4;
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

// The launch sequence of ORBextractor::operator()(image...) for a batch resident on the device.
// Pyramid levels 1 .. L-1 of the frames in fs.  For ONE frame the chain is seven (nlevels - 1) small dependent kernels whose launches take the
// host longer than the kernels run; it is captured once per (buffers, geometry) into a graph and replayed with one call.
int pyramid_chain(ygzf_ctx *c, const FrameSet &fs, int nFrames) {
    const Geometry &G = c->geo;
    const int L = c->tab.cfg.nlevels;
    const LevelGeom *dGeom = (const LevelGeom *) c->dGeom.p;
    auto launch_all = [&](bool prof) {
        for (int l = 1; l < L; l++) {
            if (prof) {
                ProfScope ps(c, KK_PYR);
                launch_pyr_resize(c->stream, fs, dGeom, G.lv[l], l, nFrames, pyr_tabs(c));
            } else
                launch_pyr_resize(c->stream, fs, dGeom, G.lv[l], l, nFrames, pyr_tabs(c));
        }
    };
    if (!G.pyrPlan.empty() && nFrames <= c->pyrStripFrames) {   // a few frames: the whole chain in one launch (large images: from the plan's base level on)
        FrameSet fb = fs;
        for (int l = 1; l <= G.pyrBase; l++) {
            ProfScope ps(c, KK_PYR);
            launch_pyr_resize(c->stream, fs, dGeom, G.lv[l], l, nFrames, pyr_tabs(c));
        }
        if (G.pyrBase > 0) {
            fb.img0 = fs.pyr + G.lv[G.pyrBase].off;
            fb.img0_stride = fs.pyr_stride;
            fb.img0_pitch = G.lv[G.pyrBase].pitch;
        }
        ProfScope ps(c, KK_PYR);
        launch_pyr_strips(c->stream, fb, L - G.pyrBase, (const PyrStripPlan *) ((const char *) c->dPyrPlan.p + sizeof G.pyrLevels), (const PyrStripLevel *) c->dPyrPlan.p,
                          (int) G.pyrPlan.size(), G.pyrStripOffA, G.pyrStripOffB, G.pyrStripLds, nFrames,
                          (const int *) c->dXofs.p, (const short *) c->dXalpha.p, (const int *) c->dYofs.p, (const short *) c->dYbeta.p);
        return YGZF_OK;
    }
    if (nFrames != 1 || L < 3 || !c->useGraphs || c->profile || c->debugSync) {
        launch_all(true);
        return YGZF_OK;
    }
    const void *key[6] = {c->dGeom.p, c->dXofs.p, c->dPyrCols.p, c->dPyrRows.p, c->dPyrTiles.p,
                          (const void *) (((uintptr_t) G.w << 32) | (uintptr_t) (unsigned) G.h)};
    if (!c->pyrGraph.exec || memcmp(key, c->pyrGraphKey, sizeof key) != 0) {
        pyr_chain_graph_destroy(&c->pyrGraph);
        const hipError_t e = pyr_chain_graph_build(&c->pyrGraph, fs, dGeom, G.lv.data(), L, pyr_tabs(c));
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
int mark_pyramid_done(ygzf_ctx *c) {
    if (!c->evPyrDone) HIPCHECK(c, hipEventCreateWithFlags(&c->evPyrDone, hipEventDisableTiming));
    HIPCHECK(c, hipEventRecord(c->evPyrDone, c->stream));
    c->evPyrDoneValid = true;
    return YGZF_OK;
}

// pyramidReady: dPyr already holds the pyramid of the frame to extract (ygzf_compute_pyramid / ygzf_extract_resident) -- the previous
// extraction's pyramid is gone from it, so the pyramid carry of ygzf_align_batch_prev is valid only when the caller copied it out BEFORE
// overwriting dPyr (pyramidCarried; ygzf_compute_pyramid in extract-ahead mode does).
int run_extract(ygzf_ctx *c, const FrameSet &fs, int nFrames, bool pyramidReady, bool pyramidCarried) {
    const Geometry &G = c->geo;
    c->pyrResident = false;
    c->aheadPending = false;
    const int L = c->tab.cfg.nlevels;
    const LevelGeom *dGeom = (const LevelGeom *) c->dGeom.p;
    // slot 0 of the output arrays carries the last frame of the previous batch (Last frame of pair 0 in ygzf_match_batch_prev)
    ygzf_kp *outKp = (ygzf_kp *) c->dOutKp.p;
    uint8_t *outDesc = (uint8_t *) c->dOutDesc.p;
    int *outCnt = (int *) c->dOutCnt.p;
    if (!c->carryLaunched) {
        if (c->carryOff) c->slot0Stale = true;
        else {
            launch_carry_slot(c->stream, outKp, outDesc, outCnt, (c->carryValid && c->lastFrames > 0 && G.kpStride > 0) ? (long long) c->lastFrames : 0, G.kpStride);
            c->slot0Stale = false;
        }
    }
    c->carryLaunched = false;
    outKp += G.kpStride;
    outDesc += (size_t) G.kpStride * 32;
    outCnt += 1;
    if (c->alignCarry && !c->carryOff && G.pyrBytes > 0) {   // the previous batch's last pyramid is the reference of pair 0 in ygzf_align_batch_prev
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
            unsigned cells = 0, usedMin = 0, extra = 0, planBits = 0, cornerQuads = 0, runs = 0;
            const volatile unsigned *hs = c->hFastStats;   // written by the copy engine, possibly right now
            for (int k = 0; k < 64; k++) {
                cells += hs[8 * k]; usedMin += hs[8 * k + 1]; extra += hs[8 * k + 2]; planBits |= hs[8 * k + 3];
                cornerQuads += hs[8 * k + 4]; runs += hs[8 * k + 7];
            }
            const unsigned plan = planBits & 3u;
            if (cells >= 32 && (plan == 1 || plan == 2)) {
                if (plan == 1) c->fastExtraRounds = (double) extra / cells;   // only the one-pass plan sees every FAST(minTh) corner
                // a score round costs ~180 vector instructions per wave, a second pass 1 ~700: iniTh first pays when the rounds it saves
                // outweigh the second passes of the cells that are empty at iniTh
                c->fastIniFirst = c->fastExtraRounds * 180.0 > ((double) usedMin / cells) * 700.0;
                if (runs > 0) {
                    c->fastCornerQuadsPerRun = (double) cornerQuads / runs;
                    c->fastRunsPerCell = (double) runs / cells;
                }
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
            // Large frames in large launches: persistent waves that draw their cells from per-XCD counters (extract_kernels.hip, k_fast_tab_persist) -- as many
            // workgroups as the device holds at once.  Measured on MI355X (profiles/r06_fast_persist_ab.txt), isolated and inside the three-context
            // pipeline: 1920x1080 (1620 cell groups per frame) 1484 -> 1248 us per 128 frames, 48.9 -> 51.3 k frames/s; 3840x2160 stereo 3.5 -> 2.7 ms
            // per 64 frames, 11.6 -> 12.7 k frames/s; 752x480 (259 groups) 473 -> 443 us per 256 frames alone but 237 -> 228 k frames/s in the pipeline:
            // waves that never leave keep the other contexts' octree / matcher workgroups (71 KB of LDS each) waiting for a CU, and at that size
            // the cell loop gains less than they lose.  Hence the rule: frames of >= 1000 cell groups.
            const size_t fastLds = tab ? fast_tab_lds_bytes(G.fastWinRows, G.fastSmapRows, G.fastQuadCap) : 0;
            const int perCu = tab ? (int) std::min<size_t>((size_t) c->fastPersistWgs, (size_t) (160 * 1024) / std::max<size_t>(fastLds, 1)) : 0;
            const int resident = perCu * c->cuCount;
            const int persistMode = c->fastPersistMode;   // tests: 1 always, 0 never
            // (forced or not, never for a launch of fewer than 64 cell groups: the items are dealt to the eight XCDs' workgroups, and a launch of two workgroups has none on six of them)
            const bool persist = tab && resident > 0 && (long long) nFrames * G.totalGroups >= 64 &&
                                 (persistMode == 1 || (persistMode != 0 && G.totalGroups >= 1000 && (long long) nFrames * G.totalGroups >= 4LL * resident));
            if (persist) {
                int rcC = ensure(c, c->dFastCtr, kFastPersistCounterBytes);
                if (rcC) return rcC;
                HIPCHECK(c, hipMemsetAsync(c->dFastCtr.p, 0, kFastPersistCounterBytes, c->stream));
            }
            ProfScope ps(c, persist ? KK_FASTP : tab ? KK_FAST : KK_FASTQ);
            if (persist)
                launch_fast_tab_persist(c->stream, fs, (const FastCellRec *) c->dFastCells.p, c->tab.cfg.ini_th_fast, c->tab.cfg.min_th_fast,
                                        (unsigned short *) c->dCellCnt.p, (unsigned *) c->dSlots.p, G.totalCells, G.totalSlots, G.totalGroups, G.fastSmapRows,
                                        nFrames, G.fastWinRows, G.fastQuadCap, iniFirst, collect ? (unsigned *) c->dFastStats.p : nullptr,
                                        (unsigned *) c->dFastCtr.p, (int) std::min<long long>(resident, (long long) nFrames * G.totalGroups));
            else if (tab)
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
            hipStream_t so = c->stream;
            ProfScope ps(c, KK_OCTREE, so);
            const bool small = c->haveOctSmall && nFrames * L <= c->octSmallWgs;
            if (small) {
                const auto &grp = c->octSmall;
                // helper workgroups for the keys of a level (k_octree): as many as keep the WHOLE launch resident at one workgroup per compute unit
                // (workgroup 0 of a level waits for its helpers), at most 8, and only where a level is worth the hand-over (a few microseconds)
                int helpers = c->octHelpersForced;
                if (helpers < 0) helpers = G.totalCells >= 2000 ? std::min(8, std::max(1, c->cuCount) / std::max(1, nFrames * L)) : 1;
                helpers = std::max(1, std::min(helpers, std::min(8, std::max(1, c->cuCount / std::max(1, nFrames * L)))));
                int *gHist = nullptr, *gDone = nullptr;
                if (helpers > 1) {
                    const size_t words = (size_t) nFrames * L * (helpers - 1) * grp.histBins, bytes = (words + 64) * sizeof(int) + (size_t) nFrames * L * sizeof(int);
                    void *old = c->dOctHist.p;
                    int rcH = ensure(c, c->dOctHist, bytes);
                    if (rcH) return rcH;
                    // the counters only grow -- by helpers - 1 per launch -- and are zeroed once per layout
                    if (old != c->dOctHist.p || c->octHistWords != words || c->octDoneTarget > (1 << 30)) {
                        HIPCHECK(c, hipMemsetAsync((int *) c->dOctHist.p + words, 0, (size_t) nFrames * L * sizeof(int), so));
                        c->octHistWords = words;
                        c->octDoneTarget = 0;
                    }
                    c->octDoneTarget += helpers - 1;
                    gHist = (int *) c->dOctHist.p;
                    gDone = gHist + words;
                }
                launch_octree(so, dGeom, L, grp.l0, grp.n, (const unsigned short *) c->dCellCnt.p, (const unsigned *) c->dSlots.p,
                              G.totalCells, G.totalSlots, (unsigned *) c->dK0.p, (unsigned *) c->dV0.p, (unsigned *) c->dK1.p,
                              (unsigned *) c->dV1.p, (unsigned *) c->dXY.p, G.candStride, (unsigned *) c->dLvlXY.p,
                              (unsigned char *) c->dLvlScore.p, (int *) c->dLvlCnt.p, (int *) c->dLvlCand.p,
                              (uint2 *) c->dProcOrder.p, G.kpStride, grp.cap, 0, grp.lds, nFrames, odbg, nullptr, grp.regionInts, grp.histBins,
                              helpers, gHist, gDone, c->octDoneTarget, c->octHelperSpin);
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
        if (odbg) {
            long long st[kOctDbgWords];
            HIPCHECK(c, hipStreamSynchronize(c->stream));
            HIPCHECK(c, hipMemcpy(st, odbg, sizeof st, hipMemcpyDeviceToHost));
            for (int l = 0; l < L; l++)
                fprintf(stderr, "[ygzf octree lvl %d, 10ns ticks] prefix %lld keys %lld sort %lld bfs %lld final %lld  M=%lld n=%lld\n", l, st[l * 8 + 1] - st[l * 8],
                        st[l * 8 + 2] - st[l * 8 + 1], st[l * 8 + 3] - st[l * 8 + 2], st[l * 8 + 4] - st[l * 8 + 3], st[l * 8 + 5] - st[l * 8 + 4], st[l * 8 + 6], st[l * 8 + 7]);
            for (int l = 0; l < L; l++) {   // the tree passes (-DYGZF_PHASE_CLOCK builds only): list size after the one-wave head, after every full pass (+) and every expand round (-), ticks since the passes began
                const long long *ev = st + 16 * 8 + 16 + l * 16;
                if (!ev[1]) continue;
                fprintf(stderr, "[ygzf octree lvl %d passes]", l);
                for (int k = 0; k < 8 && ev[2 * k + 1]; k++) fprintf(stderr, " %+lld@%lld", ev[2 * k], ev[2 * k + 1] - st[l * 8 + 3]);
                fprintf(stderr, "\n");
            }
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
int upload_rows(ygzf_ctx *c, void *dst, size_t dstPitch, const uint8_t *src, size_t srcPitch, int w, size_t rows) {
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

int upload_frames(ygzf_ctx *c, const uint8_t *imgs, int nFrames, int w, int h, int row_pitch, size_t frame_stride,
                         FrameSet *fs) {
    c->pyrResident = false;
    c->aheadPending = false;
    c->pyrHeld = false;
    const int pitch = align_up(w, 64);
    int rc = ensure(c, c->dImg0, (size_t) nFrames * pitch * h);
    if (rc) return rc;
    if (nFrames == 1) {
        hipPointerAttribute_t at;
        memset(&at, 0, sizeof at);
        bool pageable = true;
        if (hipPointerGetAttributes(&at, imgs) == hipSuccess) pageable = at.type != hipMemoryTypeHost && at.type != hipMemoryTypeDevice && at.type != hipMemoryTypeManaged;
        else (void) hipGetLastError();
        if (!pageable && at.type == hipMemoryTypeHost && at.devicePointer && (size_t) w * h <= c->uploadKernelBytes) {
            // a page-locked frame of a Tracking-sized call: a kernel reads it over the link where it lies (ygzf_extract_batch_host_frames says why)
            HostFrameList L;
            L.addr[0] = (unsigned long long) (uintptr_t) at.devicePointer;
            launch_gather_host_frames(c->stream, L, 1, (size_t) row_pitch, (uint8_t *) c->dImg0.p, (size_t) pitch, (size_t) pitch * h, w, h);
            HIPCHECK(c, hipGetLastError());
            fs->img0 = (const uint8_t *) c->dImg0.p;
            fs->img0_stride = (long long) pitch * h;
            fs->img0_pitch = pitch;
            fs->pyr = (uint8_t *) c->dPyr.p;
            fs->pyr_stride = c->geo.pyrBytes;
            return YGZF_OK;
        }
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
    // Host frames that already carry the device's row pitch (ygzf_host_row_pitch): the copy engine pays per ROW of a pitched copy whose source rows
    // are not multiples of 64 bytes (752-byte rows: 48 GB/s where 1920-byte rows reach 55), so a whole frame -- (h - 1) x pitch + w bytes, padding
    // included -- is handed to it as ONE row (runs of 8 / 60 image rows measured the same: profiles/r05_a_h2d_shapes.txt)
    if (row_pitch == pitch && (nFrames == 1 || frame_stride >= (size_t) pitch * h) && ((uintptr_t) imgs & 3) == 0 && (w & 3) == 0 &&
        (frame_stride & 3) == 0) {
        const size_t fsd = (size_t) pitch * h, fss = nFrames == 1 ? fsd : frame_stride;
        HIPCHECK(c, hipMemcpy2DAsync(c->dImg0.p, fsd, imgs, fss, (size_t) (h - 1) * pitch + w, (size_t) nFrames, hipMemcpyHostToDevice, c->stream));
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
    {
        int cus = 0;
        CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device));
        c->cuCount = cus > 0 ? cus : 256;
    }
    CK(hipEventCreate(&c->tStart));
    CK(hipEventCreate(&c->tStop));
    CK(upload_constants(c->tab.umax));
    CK(hipHostMalloc((void **) &c->hFastStats, kFastStatWords * sizeof(unsigned)));
    memset(c->hFastStats, 0, kFastStatWords * sizeof(unsigned));
#undef CK
    // validate the largest configuration up front (and size the buffers once)
    int rc = apply_geometry(c, max_width, max_height, max_batch);
    if (rc) return bail(rc);
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
                             &c->dNMatch, &c->dPoses, &c->dQp, &c->dProcOrder, &c->dSpill, &c->dCarryPyr, &c->dOctNodes, &c->dResPack, &c->dFastCtr,
                             &c->dPyrPlan, &c->dPyrCols, &c->dPyrRows, &c->dPyrTiles, &c->dOctHist};
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

int ygzf_get_fast_stats(const ygzf_ctx *c, float *corner_quads_per_pass, float *passes_per_cell) {
    if (!c) return YGZF_ERR_INVALID;
    if (corner_quads_per_pass) *corner_quads_per_pass = (float) c->fastCornerQuadsPerRun;
    if (passes_per_cell) *passes_per_cell = (float) c->fastRunsPerCell;
    return YGZF_OK;
}

int ygzf_phase_clocks(ygzf_ctx *c, int kernel, unsigned long long *out16, int reset) {
    if (!c || !out16 || kernel < 0 || kernel > 1) return YGZF_ERR_INVALID;
    HIPCHECK(c, hipSetDevice(c->device));
    HIPCHECK(c, hipStreamSynchronize(c->stream));
    const hipError_t e = phase_clocks_read(kernel, out16, reset != 0);
    if (e == hipErrorNotSupported) return fail(c, YGZF_ERR_INVALID, "this library was built without -DYGZF_PHASE_CLOCK (python -m orb_ygz_slam_amd.build --phase-clock)");
    HIPCHECK(c, e);
    return YGZF_OK;
}

int ygzf_set_fast_kernel(ygzf_ctx *c, int kernel) {
    if (!c) return YGZF_ERR_INVALID;
    if (kernel < YGZF_FAST_KERNEL_AUTO || kernel > YGZF_FAST_KERNEL_CELL_TABLE) return fail(c, YGZF_ERR_INVALID, "FAST kernel %d (0 auto, 1 register staging, 2 cell table + LDS-DMA)", kernel);
    c->fastKernel = kernel;
    return YGZF_OK;
}

int ygzf_set_carry_previous(ygzf_ctx *c, int on) {
    if (!c) return YGZF_ERR_INVALID;
    c->carryOff = !on;
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

int ygzf_pyramid_plan_base_host(const ygzf_extractor_cfg *cfg, int w, int h) {
    if (!cfg || cfg->nlevels < 1 || cfg->nlevels > kMaxLevels || cfg->nfeatures < 0 || !(cfg->scale_factor > 1.0f) || w < 1 || h < 1) return YGZF_ERR_INVALID;
    ygzf_ctx tmp;
    tmp.tab.init(*cfg);
    Geometry G;
    const int rc = build_geometry(&tmp, w, h, G);
    return rc ? rc : G.pyrBase;
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
    if (c->extractAhead && c->streamCopy && c->alignCarry && !c->carryOff && c->geo.pyrBytes > 0 && c->carryValid && c->lastFrames > 0) {
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
    // (the page-locked staging area as the device addresses it: the packing kernel -- 16-byte stores -- writes the levels over the link itself, on the
    // stream the copy would have taken; a byte-per-thread packing kernel doing so was slower than the copy engine: ComputePyramid 94 -> 97 us)
    const bool linkPack = c->hStageDev != nullptr && c->linkKernels && c->pyrLink;
    if (!linkPack) {
        launch_pack_levels(c->stream, fs, (const LevelGeom *) c->dGeom.p, 1, L, offs, (uint8_t *) c->dTmpC.p);
        HIPCHECK(c, hipGetLastError());
    }
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
    if (total && linkPack) {
        launch_pack_levels(rd, fs, (const LevelGeom *) c->dGeom.p, 1, L, offs, (uint8_t *) c->hStageDev);
        HIPCHECK(c, hipGetLastError());
    } else if (total) HIPCHECK(c, hipMemcpyAsync(c->hStage, c->dTmpC.p, total, hipMemcpyDeviceToHost, rd));
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

// The carry (last frame of the previous batch -> slot 0 of the outputs) depends on nothing an upload brings: queued BEFORE the host frames, it runs
// while they cross the link instead of between their arrival and the pyramid (6 us of a one-frame call).  run_extract then skips its own.
static void carry_early(ygzf_ctx *c) {
    const Geometry &G = c->geo;
    c->carryLaunched = true;
    if (c->carryOff) { c->slot0Stale = true; return; }
    c->slot0Stale = false;
    launch_carry_slot(c->stream, (ygzf_kp *) c->dOutKp.p, (uint8_t *) c->dOutDesc.p, (int *) c->dOutCnt.p,
                      (c->carryValid && c->lastFrames > 0 && G.kpStride > 0) ? (long long) c->lastFrames : 0, G.kpStride);
    c->carryLaunched = true;
}

int ygzf_extract_batch_host(ygzf_ctx *c, const uint8_t *imgs, int n_frames, int w, int h, int row_pitch, size_t frame_stride) {
    if (!c || !imgs) return fail(c, YGZF_ERR_INVALID, "null argument");
    if (row_pitch < w) return fail(c, YGZF_ERR_INVALID, "row_pitch %d < width %d", row_pitch, w);
    HIPCHECK(c, hipSetDevice(c->device));
    int rc = apply_geometry(c, w, h, n_frames);
    if (rc) return rc;
    carry_early(c);
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
    carry_early(c);
    c->pyrResident = false;
    c->aheadPending = false;
    c->pyrHeld = false;
    const int pitch = align_up(w, 64);
    if ((rc = ensure(c, c->dImg0, (size_t) n_frames * pitch * h))) return rc;
    {
        // Into a tight staging buffer with the copy engine's fast paths, then ONE launch per 128 frames that lays the rows out at the context's
        // pitch.  Frames that come as runs of u back-to-back frames at one distance S (a device slot's round-robin share of a tight clip: units of u
        // frames, S = slots x u frames) go up as ONE two-dimensional copy whose "rows" are the runs; anything else as one linear copy per frame.
        // ONE page-locked frame (a Tracking-sized call): a kernel reads it over the link where it lies.  The copy engine starts ~10 us after the copy
        // is queued and hands over to the first kernel ~8 us after it ends; a kernel's launch and hand-over are a third of that.
        if (n_frames <= c->uploadKernelFrames && (size_t) n_frames * w * h <= c->uploadKernelBytes) {
            HostFrameList L;
            bool mapped = true;
            for (int f = 0; f < n_frames && mapped; f++) {
                hipPointerAttribute_t at;
                memset(&at, 0, sizeof at);
                mapped = hipPointerGetAttributes(&at, frames[f]) == hipSuccess && at.type == hipMemoryTypeHost && at.devicePointer != nullptr;
                if (!mapped) (void) hipGetLastError();
                L.addr[f] = (unsigned long long) (uintptr_t) at.devicePointer;
            }
            if (mapped) {
                launch_gather_host_frames(c->stream, L, n_frames, (size_t) row_pitch, (uint8_t *) c->dImg0.p, (size_t) pitch, (size_t) pitch * h, w, h);
                HIPCHECK(c, hipGetLastError());
                goto uploaded;
            }
        }
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
        if (regular && row_pitch == pitch && (w & 3) == 0 && ((uintptr_t) frames[0] & 3) == 0 && (S & 3) == 0) {
            // frames that carry the device's pitch (ygzf_host_row_pitch; tight frames whose width is a multiple of 64 -- 640, 1920, 3840 -- do so by
            // themselves): the runs go straight to their place, no staging and no re-pitch launch
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
    }
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

// queues the gather kernel and the ONE copy of ygzf_batch_fetch_packed on the context's stream; the caller synchronises
int queue_packed_fetch(ygzf_ctx *c, void *host, size_t host_bytes, size_t *off_kps, size_t *off_desc, int *row_entries, size_t *bytes_out) {
    if (c->lastFrames < 1) return fail(c, YGZF_ERR_STATE, "no extracted batch");
    const size_t B = (size_t) c->lastFrames, ks = (size_t) c->geo.kpStride;
    const size_t oK = (B * sizeof(int) + 255) & ~(size_t) 255, oD = oK + ((B * ks * sizeof(ygzf_kp) + 255) & ~(size_t) 255), total = oD + B * ks * 32;
    if (total > host_bytes) return fail(c, YGZF_ERR_INVALID, "packed results need %zu bytes, %zu given", total, host_bytes);
    if (total >= (1ull << 32)) return fail(c, YGZF_ERR_UNSUPPORTED, "packed results of %zu bytes: use ygzf_batch_fetch_all", total);
    // page-locked memory the device can address: the gather kernel writes the block over the link itself (no copy-engine start-up / hand-over)
    void *direct = nullptr;
    if (c->linkKernels) {
        hipPointerAttribute_t at;
        memset(&at, 0, sizeof at);
        if (hipPointerGetAttributes(&at, host) == hipSuccess && at.type == hipMemoryTypeHost && at.devicePointer && ((uintptr_t) at.devicePointer & 3) == 0) direct = at.devicePointer;
        else (void) hipGetLastError();
    }
    int rc = direct ? YGZF_OK : ensure(c, c->dResPack, total + 256);
    if (rc) return rc;
    launch_pack_results(c->stream, (const int *) c->dOutCnt.p + 1, (const ygzf_kp *) c->dOutKp.p + ks, (const uint8_t *) c->dOutDesc.p + ks * 32, (int) B, (int) ks,
                        direct ? direct : c->dResPack.p, oK, oD);
    HIPCHECK(c, hipGetLastError());
    if (!direct) HIPCHECK(c, hipMemcpyAsync(host, c->dResPack.p, total, hipMemcpyDeviceToHost, c->stream));
    *off_kps = oK;
    *off_desc = oD;
    *row_entries = (int) ks;
    if (bytes_out) *bytes_out = total;
    return YGZF_OK;
}

int ygzf_batch_fetch_packed(ygzf_ctx *c, void *host, size_t host_bytes, size_t *off_kps, size_t *off_desc, int *row_entries, size_t *bytes_out) {
    if (!c || !host || !off_kps || !off_desc || !row_entries) return fail(c, YGZF_ERR_INVALID, "null argument");
    int rc = queue_packed_fetch(c, host, host_bytes, off_kps, off_desc, row_entries, bytes_out);
    if (rc) return rc;
    HIPCHECK(c, hipStreamSynchronize(c->stream));
    return YGZF_OK;
}

// One stereo pair from host memory with its two eyes on TWO contexts / streams of the same device (ygzf_mgpu_extract_stereo for a slot that is handed
// exactly one pair: BASELINE.json's 3840x2160 configuration, a pair per GPU).  In one context the pair is one upload of both frames followed by one
// kernel chain: 16.6 MB on the link (300 us) during which no kernel of the pair can run.  Here the right eye's upload waits for the LEFT eye's upload
// only, so it crosses the link while the left eye's pyramid / FAST / octree / descriptors run; the right eye's chain then overlaps the left one's
// tail; ComputeStereoMatches runs on the left context's stream once both are done, reading the right eye's keypoints, descriptors and pyramid where
// the right context left them.  Same kernels on the same inputs: the bytes of ygzf_extract_batch_host(both frames) + ygzf_stereo_batch.
int ygzf_stereo_pair_host(ygzf_ctx *l, ygzf_ctx *r, const uint8_t *left, const uint8_t *right, int w, int h, int row_pitch, float mb, float mbf,
                          void *host_left, void *host_right, size_t host_bytes, size_t *off_kps, size_t *off_desc, int *row_entries,
                          float *u_right, float *depth) {
    if (!l || !r || l == r || !left || !right || !host_left || !host_right || !off_kps || !off_desc || !row_entries || !u_right || !depth)
        return fail(l, YGZF_ERR_INVALID, "null argument");
    if (l->device != r->device) return fail(l, YGZF_ERR_INVALID, "the two contexts of a pair must be on one device");
    if (row_pitch < w) return fail(l, YGZF_ERR_INVALID, "row_pitch %d < width %d", row_pitch, w);
    HIPCHECK(l, hipSetDevice(l->device));
    int rc;
    if ((rc = apply_geometry(l, w, h, 1))) return rc;
    if ((rc = apply_geometry(r, w, h, 1))) { l->err = r->err; return rc; }
    if (!l->evShare) HIPCHECK(l, hipEventCreateWithFlags(&l->evShare, hipEventDisableTiming));
    if (!r->evShare) HIPCHECK(l, hipEventCreateWithFlags(&r->evShare, hipEventDisableTiming));
    carry_early(l);
    carry_early(r);
    FrameSet fl, fr;
    if ((rc = upload_frames(l, left, 1, w, h, row_pitch, 0, &fl))) return rc;
    HIPCHECK(l, hipEventRecord(l->evShare, l->stream));                 // the left eye is on the device ...
    HIPCHECK(l, hipStreamWaitEvent(r->stream, l->evShare, 0));          // ... and only then does the right eye take the link
    if ((rc = upload_frames(r, right, 1, w, h, row_pitch, 0, &fr))) { l->err = r->err; return rc; }
    if ((rc = run_extract(l, fl, 1))) return rc;
    if ((rc = run_extract(r, fr, 1))) { l->err = r->err; return rc; }
    HIPCHECK(l, hipEventRecord(r->evShare, r->stream));
    size_t ok2 = 0, od2 = 0;
    int re2 = 0;
    if ((rc = queue_packed_fetch(r, host_right, host_bytes, &ok2, &od2, &re2, nullptr))) { l->err = r->err; return rc; }
    if ((rc = queue_packed_fetch(l, host_left, host_bytes, off_kps, off_desc, row_entries, nullptr))) return rc;
    HIPCHECK(l, hipStreamWaitEvent(l->stream, r->evShare, 0));
    if ((rc = stereo_across(l, r, mb, mbf))) return rc;
    const size_t ks = (size_t) l->geo.kpStride;
    HIPCHECK(l, hipMemcpyAsync(u_right, l->dSt[1].p, 4 * ks, hipMemcpyDeviceToHost, l->stream));
    HIPCHECK(l, hipMemcpyAsync(depth, l->dSt[2].p, 4 * ks, hipMemcpyDeviceToHost, l->stream));
    HIPCHECK(l, hipStreamSynchronize(r->stream));
    HIPCHECK(l, hipStreamSynchronize(l->stream));
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
    if (c->hStageDev && c->linkKernels) {
        // count, keypoint row and descriptor row written into the page-locked staging area by a kernel, over the link: three copies by the copy engine
        // start ~10 us after they are queued and the last one hands back ~8 us after it ends -- a launch does neither
        launch_pack_results(c->stream, (const int *) c->dOutCnt.p + 1, (const ygzf_kp *) c->dOutKp.p + ks, (const uint8_t *) c->dOutDesc.p + ks * 32, 1, (int) ks, c->hStageDev, oK, oD);
        HIPCHECK(c, hipGetLastError());
    } else {
        HIPCHECK(c, hipMemcpyAsync(c->hStage, (int *) c->dOutCnt.p + 1, sizeof(int), hipMemcpyDeviceToHost, c->stream));
        HIPCHECK(c, hipMemcpyAsync(c->hStage + oK, (ygzf_kp *) c->dOutKp.p + ks, ks * sizeof(ygzf_kp), hipMemcpyDeviceToHost, c->stream));   // slot 1 = frame 0
        HIPCHECK(c, hipMemcpyAsync(c->hStage + oD, (uint8_t *) c->dOutDesc.p + ks * 32, ks * 32, hipMemcpyDeviceToHost, c->stream));
    }
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

}  // extern "C"
