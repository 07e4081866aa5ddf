// ygzf_api_stereo.hip -- Frame::ComputeStereoMatches (C ABI of libygzf, include/ygzf.h; product code: no CPU fallback, nothing from oracle/ is included or linked).
#include "ygzf_ctx.h"

extern "C" {

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

}  // extern "C"

// ComputeStereoMatches of ONE pair whose left eye is frame 0 of context l's last extraction and whose right eye is frame 0 of context r's
// (ygzf_stereo_pair_host); queued on l's stream, which the caller has made wait for r's extraction.  Results where ygzf_stereo_batch leaves pair 0's.
int stereo_across(ygzf_ctx *l, ygzf_ctx *r, float mb, float mbf) {
    if (l->lastFrames != 1 || r->lastFrames != 1) return fail(l, YGZF_ERR_STATE, "stereo across contexts needs one extracted frame in each");
    if (!(mb > 0)) return fail(l, YGZF_ERR_INVALID, "baseline mb must be positive");
    const Geometry &G = l->geo;
    if (G.kpStride != r->geo.kpStride || G.w != r->geo.w || G.h != r->geo.h) return fail(l, YGZF_ERR_STATE, "the two eyes differ in geometry");
    if (G.kpStride > 65535) return fail(l, YGZF_ERR_UNSUPPORTED, "more than 65535 keypoints per frame");
    int rc;
    const size_t per = (size_t) G.kpStride;
    if ((rc = ensure(l, l->dSt[0], sizeof(StereoRec) * per + 64)) || (rc = ensure(l, l->dSt[1], 4 * per + 64)) || (rc = ensure(l, l->dSt[2], 4 * per + 64)) ||
        (rc = ensure(l, l->dSt[3], 4 * per + 64)) || (rc = ensure(l, l->dStBins, sizeof(int) * kStereoBinInts)))
        return rc;
    StereoArgs A;
    memset(&A, 0, sizeof A);
    A.keys = (const ygzf_kp *) l->dOutKp.p;       // slot 0 = carry; frame 0 lives in slot 1 -- of either context
    A.desc = (const uint8_t *) l->dOutDesc.p;
    A.keysR = (const ygzf_kp *) r->dOutKp.p;
    A.descR = (const uint8_t *) r->dOutDesc.p;
    A.cntR = (const int *) r->dOutCnt.p;
    A.keyStride = (long long) G.kpStride;
    A.keyOffL = G.kpStride;
    A.keyOffR = G.kpStride;
    A.cnt = (const int *) l->dOutCnt.p;
    A.cntStride = 1;
    A.cntOffL = 1;
    A.cntOffR = 1;
    A.fs = l->lastFs;
    A.fsR = r->lastFs;                            // the kernels address the right image as frame frame0 + 1: one frame back, so that this is r's frame 0
    A.fsR.img0 -= A.fsR.img0_stride;
    A.fsR.pyr -= A.fsR.pyr_stride;
    A.frame0 = 0;
    A.frameStep = 0;
    fill_stereo_common(l, A, mb, mbf, G.h);
    A.rec = (StereoRec *) l->dSt[0].p;
    A.recStride = (long long) per;
    A.binStart = (int *) l->dStBins.p;
    A.uRight = (float *) l->dSt[1].p;
    A.depth = (float *) l->dSt[2].p;
    A.sad = (int *) l->dSt[3].p;
    A.outStride = (long long) per;
    {
        ProfScope ps(l, KK_STEREO);
        launch_stereo(l->stream, A, 1, G.kpStride, G.kpStride);
    }
    HIPCHECK(l, hipGetLastError());
    l->lastStereoPairs = 1;
    return YGZF_OK;
}

extern "C" {

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
        launch_pyr_resize(c->stream, fs, (const LevelGeom *) c->dGeom.p, G.lv[l], l, 2, pyr_tabs(c));
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

}  // extern "C"
