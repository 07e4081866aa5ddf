// ygzf_api_align.hip -- SparseImgAlign::run (one pair from host pyramids, from the image cache, the resident batch), the image cache and ORBmatcher::FindDirectProjection (C ABI of libygzf, include/ygzf.h; product code: no CPU fallback, nothing from oracle/ is included or linked).
#include "ygzf_ctx.h"

extern "C" {

static FrameSet cache_frameset_impl(const ygzf_ctx *c);
// The reference patches of all levels in a kernel of their own (k_sia_precompute) pay for launches of a few pairs -- one pair: 365 -> 305 us, the
// 71 us the pair's one workgroup spent on them become 10 us chip-wide -- and cost large ones: at 256 pairs the in-kernel form overlaps one pair's
// patches with another pair's solve on the same CU and its single cache stays L2-resident (971 against 1075 us per launch, 142 k against 134 k frames/s).
constexpr int kSiaPerLevelPairs = 32;


// ---- SparseImgAlign::run ------------------------------------------------------------------------------------------------
// ygzf_sia_run (pyramids from host memory) and ygzf_sia_run_cached (pyramids of two image-cache slots, already on the device).
// cached: ref_slot / cur_slot >= 0, cur carries the pose only (its level arrays are not read, ref's neither).
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
            FrameSet fs = cache_frameset_impl(c);
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

// ---- image cache (KeyFrame / current-frame pyramids resident in HBM) + FindDirectProjection batch -------------------------------------
static FrameSet cache_frameset_impl(const ygzf_ctx *c) {
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
    FrameSet fs = cache_frameset_impl(c);
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
    A.cache = cache_frameset_impl(c);
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
    if (c->slot0Stale) return fail(c, YGZF_ERR_STATE, "the previous frame was not carried (ygzf_set_carry_previous is off)");
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
            constexpr long capKb = 74;
            if (B - first >= 128) {
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
