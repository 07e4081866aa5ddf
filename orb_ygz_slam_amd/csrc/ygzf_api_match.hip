// ygzf_api_match.hip -- ORBmatcher's entry points: SearchByProjection (all overloads), SearchByBoW, SearchForInitialization, SearchForTriangulation, the Frame grid, Frame::isInFrustum, MapPoint::ComputeDistinctiveDescriptors, ORBVocabulary::transform (C ABI of libygzf, include/ygzf.h; product code: no CPU fallback, nothing from oracle/ is included or linked).
#include "ygzf_ctx.h"

extern "C" {

// ---- matcher ----------------------------------------------------------------------------------------------------------
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
    if (c->slot0Stale) return fail(c, YGZF_ERR_STATE, "the previous frame was not carried (ygzf_set_carry_previous is off)");
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
        ProfScope ps(c, KK_MATCH);
        launch_match_last(c->stream, A, B, lds);
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
    A.mode = mode;
    A.specDeep = mode == 1 ? 1 : 0;   // best AND runner-up among the free candidates: lists of eight (match_kernels.hip)
    // accept threshold (TH_HIGH, ORBdist of mode 2).  A candidate becomes "best" only with dist < 256 (`int bestDist = 256 ... if (dist < bestDist)`,
    // src/ORBmatcher.cc:1413-1429), so a threshold of 256 or more accepts exactly what 255 accepts -- except that the reference then also "accepts" a
    // MapPoint without any free candidate and writes mvpMapPoints[-1] (:1431-1432, undefined; its callers pass 100 and 64).  Defined here, as in the
    // oracle, as: no candidate, no match.  (Found by the fuzzer once its KeyFrame leg compared with the oracle: 256 reached the kernel, whose keys
    // reserve dist > 255 for "nothing" -- wrong counts and an out-of-bounds store.)
    A.maxDist = max_dist > 255 ? 255 : max_dist < 0 ? 0 : max_dist;
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

}  // extern "C"
