// ygzf_api_detect.hip -- the Thirdparty/fast replacement, the DSO / FAST_KEYPOINT detectors of ORBextractor::operator()(Frame*, ...) and descriptors of keys a frame already holds (C ABI of libygzf, include/ygzf.h; product code: no CPU fallback, nothing from oracle/ is included or linked).
#include "ygzf_ctx.h"

extern "C" {

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
        launch_pyr_resize(c->stream, fs, dGeom, G.lv[l], l, 1, pyr_tabs(c));
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
        launch_pyr_resize(c->stream, *fs, dGeom, G.lv[l], l, 1, pyr_tabs(c));
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

}  // extern "C"
