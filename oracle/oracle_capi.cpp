// oracle_capi.cpp -- CPU ORACLE (test infrastructure): flat C entry points so that tests/, bench.py's cpu_baseline
// leg and __graft_entry__.smoke() can drive the oracle through ctypes.  Nothing in the product links this.
#include <chrono>
#include <cmath>
#include <cstring>
#include <thread>

#include "ygz_oracle.h"

using namespace ygzo;

extern "C" {

void *yo_extractor_create(int nfeatures, float scaleFactor, int nlevels, int iniTh, int minTh) {
    return new Extractor(nfeatures, scaleFactor, nlevels, iniTh, minTh);
}
void yo_extractor_destroy(void *e) { delete (Extractor *) e; }

void yo_extractor_tables(void *e_, float *scale, float *inv_scale, float *sigma2, float *inv_sigma2, int *nfeat, int *umax16) {
    Extractor *e = (Extractor *) e_;
    for (int i = 0; i < e->nlevels; i++) {
        scale[i] = e->mvScaleFactor[i];
        inv_scale[i] = e->mvInvScaleFactor[i];
        sigma2[i] = e->mvLevelSigma2[i];
        inv_sigma2[i] = e->mvInvLevelSigma2[i];
        nfeat[i] = e->mnFeaturesPerLevel[i];
    }
    for (int i = 0; i < 16; i++) umax16[i] = e->umax[i];
}

void yo_level_size(void *e, int w, int h, int level, int *lw, int *lh) { ((Extractor *) e)->LevelSize(w, h, level, lw, lh); }

void yo_pyramid(void *e, const uint8_t *img, int w, int h, int stride) { ((Extractor *) e)->ComputePyramid(img, w, h, stride); }

void yo_get_level(void *e_, int level, uint8_t *out) {
    Extractor *e = (Extractor *) e_;
    const Image &im = e->mvImagePyramid[level];
    std::memcpy(out, im.d.data(), im.d.size());
}

// FAST candidates of one pyramid level (after yo_pyramid): region coordinates, cell-major order.
int yo_cell_candidates(void *e_, int level, int *xs, int *ys, int *scores, int cap) {
    Extractor *e = (Extractor *) e_;
    std::vector<KeyPoint> v;
    e->CellCandidates(level, v);
    int n = (int) v.size();
    for (int i = 0; i < n && i < cap; i++) {
        xs[i] = (int) v[i].x;
        ys[i] = (int) v[i].y;
        scores[i] = (int) v[i].response;
    }
    return n;
}

// DistributeOctTree on an explicit candidate list; returns number kept, out_idx = index into the input list.
int yo_octree(void *e_, const int *xs, const int *ys, const int *resp, int n, int minX, int maxX, int minY, int maxY,
              int N, int *out_idx, int cap) {
    Extractor *e = (Extractor *) e_;
    std::vector<KeyPoint> v(n);
    for (int i = 0; i < n; i++) {
        v[i].x = (float) xs[i];
        v[i].y = (float) ys[i];
        v[i].response = (float) resp[i];
        v[i].class_id = i;
        v[i].size = 7;
        v[i].angle = -1;
        v[i].octave = 0;
    }
    std::vector<KeyPoint> r = e->DistributeOctTree(v, minX, maxX, minY, maxY, N);
    for (size_t i = 0; i < r.size() && (int) i < cap; i++) out_idx[i] = r[i].class_id;
    return (int) r.size();
}

int yo_extract(void *e_, const uint8_t *img, int w, int h, int stride, KeyPoint *kps, int cap, uint8_t *desc) {
    Extractor *e = (Extractor *) e_;
    std::vector<KeyPoint> k;
    std::vector<uint8_t> d;
    e->Extract(img, w, h, stride, k, d);
    int n = (int) k.size();
    if (n > cap) return -n;
    if (n > 0) {   // empty vectors have a null data(): memcpy(…, nullptr, 0) is undefined
        std::memcpy(kps, k.data(), sizeof(KeyPoint) * n);
        std::memcpy(desc, d.data(), 32 * (size_t) n);
    }
    return n;
}

// Keypoints of one level before scaling (level coordinates), with angle -- used to test the device stages one by one.
int yo_level_keypoints(void *e_, int level, KeyPoint *kps, int cap) {
    Extractor *e = (Extractor *) e_;
    std::vector<std::vector<KeyPoint>> all;
    e->ComputeKeyPointsOctTree(all);
    int n = (int) all[level].size();
    for (int i = 0; i < n && i < cap; i++) kps[i] = all[level][i];
    return n;
}

float yo_ic_angle(void *e_, const uint8_t *img, int w, int h, float x, float y) {
    Extractor *e = (Extractor *) e_;
    Image im(w, h);
    std::memcpy(im.d.data(), img, (size_t) w * h);
    return e->ICAngle(im, x, y);
}

void yo_descriptor(void *e_, const uint8_t *blurred, int w, int h, float x, float y, float angle, uint8_t *desc) {
    Extractor *e = (Extractor *) e_;
    Image im(w, h);
    std::memcpy(im.d.data(), blurred, (size_t) w * h);
    KeyPoint kp{x, y, 31, angle, 0, 0, -1};
    e->ComputeDescriptor(kp, im, desc);
}

void yo_blur_kernel(int mode, int *k7) { gaussian_kernel7_s2(mode, k7); }

void yo_blur(const uint8_t *src, int w, int h, uint8_t *dst) {
    Image s(w, h), d;
    std::memcpy(s.d.data(), src, (size_t) w * h);
    gaussian_blur7_s2_u8(s, d);
    std::memcpy(dst, d.d.data(), (size_t) w * h);
}

void yo_resize(const uint8_t *src, int sw, int sh, uint8_t *dst, int dw, int dh) {
    Image s(sw, sh), d(dw, dh);
    std::memcpy(s.d.data(), src, (size_t) sw * sh);
    resize_linear_u8(s, d);
    std::memcpy(dst, d.d.data(), (size_t) dw * dh);
}

// operator()(Frame*, ..., DSO_KEYPOINT): keys holds n_existing keys on entry; returns the total count (existing + new)
int yo_extract_dso(void *e_, const uint8_t *img, int w, int h, int stride, KeyPoint *keys, int n_existing, int cap, uint8_t *desc, int *grid_size) {
    Extractor *e = (Extractor *) e_;
    std::vector<KeyPoint> k(keys, keys + n_existing);
    std::vector<uint8_t> d;
    e->mnGridSize = *grid_size;
    e->ExtractDSO(img, w, h, stride, k, d);
    *grid_size = e->mnGridSize;
    const int n = (int) k.size();
    if (n > cap) return -n;
    if (n > 0) {
        std::memcpy(keys, k.data(), sizeof(KeyPoint) * n);
        std::memcpy(desc, d.data(), 32 * (size_t) n);
    }
    return n;
}

// operator()(Frame*, ..., FAST_KEYPOINT) (mode 1) / the Frame overload over the multi-level ComputeKeyPointsDSO (mode 3): keys holds
// n_existing keys on entry; returns the total count (existing + new), -count when cap is too small
int yo_extract_grid(void *e_, int mode, const uint8_t *img, int w, int h, int stride, KeyPoint *keys, int n_existing, int cap, uint8_t *desc, int *grid_size) {
    Extractor *e = (Extractor *) e_;
    std::vector<KeyPoint> k(keys, keys + n_existing);
    std::vector<uint8_t> d;
    if (mode == 1) e->ExtractFast(img, w, h, stride, k, d);
    else {
        e->mnGridSize = *grid_size;
        e->ExtractDSOMultiLevel(img, w, h, stride, k, d);
        *grid_size = e->mnGridSize;
    }
    const int n = (int) k.size();
    if (n > cap) return -n;
    if (n > 0) {
        std::memcpy(keys, k.data(), sizeof(KeyPoint) * n);
        std::memcpy(desc, d.data(), 32 * (size_t) n);
    }
    return n;
}

// the "existing ones" loop of the Frame overload (src/ORBextractor.cc:1093-1106) on the pyramid of `img`;
// recompute != 0: IC_Angle first, as ComputeKeyPointsDSOSingleLevel :1380-1383 (angles are written back into keys)
void yo_describe_keys(void *e_, const uint8_t *img, int w, int h, int stride, KeyPoint *keys, int n, int recompute, uint8_t *desc) {
    Extractor *e = (Extractor *) e_;
    e->ComputePyramid(img, w, h, stride);
    std::vector<Image> blurred(e->nlevels);
    for (int i = 0; i < e->nlevels; i++) gaussian_blur7_s2_u8(e->mvImagePyramid[i], blurred[i]);
    for (int i = 0; i < n; i++) {
        KeyPoint tmp = keys[i];
        tmp.x *= e->mvInvScaleFactor[tmp.octave];
        tmp.y *= e->mvInvScaleFactor[tmp.octave];
        if (recompute) keys[i].angle = tmp.angle = e->ICAngle(e->mvImagePyramid[tmp.octave], tmp.x, tmp.y);
        e->ComputeDescriptor(tmp, blurred[tmp.octave], desc + (size_t) i * 32);
    }
}

float yo_shi_tomasi(void *e_, const uint8_t *img, int w, int h, int u, int v) {
    Extractor *e = (Extractor *) e_;
    Image im(w, h);
    std::memcpy(im.d.data(), img, (size_t) w * h);
    return e->ShiTomasiScore(im, u, v);
}

float yo_fast_atan2(float y, float x) { return fast_atan2_deg(y, x); }
void yo_sincos_deg(float a, float *c, float *s) { sincos_deg(a, c, s); }
int yo_cv_round(double v) { return cv_round(v); }

int yo_fast9(const uint8_t *img, int stride, int w, int h, int threshold, int nonmax, int *xs, int *ys, int *scores, int cap) {
    std::vector<FastPt> v;
    fast9(img, stride, w, h, threshold, nonmax != 0, v);
    for (size_t i = 0; i < v.size() && (int) i < cap; i++) {
        xs[i] = v[i].x;
        ys[i] = v[i].y;
        scores[i] = v[i].score;
    }
    return (int) v.size();
}

int yo_hamming(const uint8_t *a, const uint8_t *b) { return descriptor_distance(a, b); }

// ---- matcher ------------------------------------------------------------------------------------------------
struct yo_frame {  // POD mirror of FrameView for ctypes
    int N;
    const KeyPoint *keys;
    const uint8_t *desc;
    const float *uRight;
    float minX, minY, maxX, maxY;
    float fx, fy, cx, cy, mb, mbf;
    const float *scaleFactors;
    int nlevels;
};
static FrameView to_view(const yo_frame *f) {
    FrameView v;
    v.N = f->N;
    v.keys = f->keys;
    v.desc = f->desc;
    v.uRight = f->uRight;
    v.minX = f->minX; v.minY = f->minY; v.maxX = f->maxX; v.maxY = f->maxY;
    v.gridInvW = (float) Grid::COLS / (f->maxX - f->minX);  // src/Frame.cc:302-303
    v.gridInvH = (float) Grid::ROWS / (f->maxY - f->minY);
    v.fx = f->fx; v.fy = f->fy; v.cx = f->cx; v.cy = f->cy; v.mb = f->mb; v.mbf = f->mbf;
    v.scaleFactors = f->scaleFactors;
    v.nlevels = f->nlevels;
    return v;
}

int yo_features_in_area(const yo_frame *f, float x, float y, float r, int minLevel, int maxLevel, int *out, int cap) {
    FrameView v = to_view(f);
    Grid g;
    g.Assign(v);
    std::vector<int> idx;
    g.FeaturesInArea(v, x, y, r, minLevel, maxLevel, idx);
    for (size_t i = 0; i < idx.size() && (int) i < cap; i++) out[i] = idx[i];
    return (int) idx.size();
}

int yo_search_by_projection_last(const yo_frame *cur, int lastN, const KeyPoint *last_keys, const uint8_t *mp_valid,
                                 const uint8_t *outlier, const uint8_t *mp_has_obs, const float *mp_world,
                                 const uint8_t *mp_desc, const float *Rcw, const float *tcw, const float *Rlw,
                                 const float *tlw, float th, int bMono, int checkLevel, int checkOri, uint8_t *cur_owner,
                                 int *cur_match) {
    FrameView v = to_view(cur);
    Grid g;
    g.Assign(v);
    ProjLastInput in;
    in.N = lastN;
    in.keys = last_keys;
    in.mp_valid = mp_valid;
    in.outlier = outlier;
    in.mp_has_obs = mp_has_obs;
    in.mp_world = mp_world;
    in.mp_desc = mp_desc;
    std::memcpy(in.Rcw, Rcw, 36);
    std::memcpy(in.tcw, tcw, 12);
    std::memcpy(in.Rlw, Rlw, 36);
    std::memcpy(in.tlw, tlw, 12);
    return search_by_projection_last(v, g, in, th, bMono != 0, checkLevel != 0, checkOri != 0, cur_owner, cur_match);
}

int yo_search_by_projection_mappoints(const yo_frame *F, int M, const uint8_t *track_in_view, const uint8_t *bad,
                                      const uint8_t *mp_has_obs, const float *projX, const float *projY,
                                      const float *projXR, const float *viewCos, const int *scaleLevel,
                                      const uint8_t *mp_desc, float th, int checkLevel, float nnratio, uint8_t *owner,
                                      int *match) {
    FrameView v = to_view(F);
    Grid g;
    g.Assign(v);
    ProjMapPointsInput in;
    in.M = M;
    in.track_in_view = track_in_view;
    in.bad = bad;
    in.mp_has_obs = mp_has_obs;
    in.projX = projX;
    in.projY = projY;
    in.projXR = projXR;
    in.viewCos = viewCos;
    in.scaleLevel = scaleLevel;
    in.mp_desc = mp_desc;
    return search_by_projection_mappoints(v, g, in, th, checkLevel != 0, nnratio, owner, match);
}

int yo_search_by_projection_kf(const yo_frame *cur, int M, const uint8_t *usable, const float *world, const float *maxDistInv,
                               const float *minDistInv, const float *mfMaxDistance, const float *kf_angle, const uint8_t *mp_desc,
                               const float *Rcw, const float *tcw, float logScaleFactor, int nScaleLevels, float th, int ORBdist,
                               int checkOri, uint8_t *cur_owner, int *cur_match, uint8_t *out_valid, float *out_u, float *out_v,
                               int *out_level) {
    FrameView v = to_view(cur);
    Grid g;
    g.Assign(v);
    ProjKFInput in;
    in.M = M;
    in.usable = usable;
    in.world = world;
    in.maxDistInv = maxDistInv;
    in.minDistInv = minDistInv;
    in.mfMaxDistance = mfMaxDistance;
    in.kf_angle = kf_angle;
    in.mp_desc = mp_desc;
    std::memcpy(in.Rcw, Rcw, 36);
    std::memcpy(in.tcw, tcw, 12);
    in.logScaleFactor = logScaleFactor;
    in.nScaleLevels = nScaleLevels;
    return search_by_projection_kf(v, g, in, th, ORBdist, checkOri != 0, cur_owner, cur_match, out_valid, out_u, out_v, out_level);
}

int yo_search_by_bow(int nNodes, const int *kf_off, const int *kf_idx, const int *f_off, const int *f_idx, const uint8_t *kf_valid,
                     const KeyPoint *kf_keys, const uint8_t *kf_desc, int nF, const KeyPoint *f_keys, const uint8_t *f_desc, float nnratio,
                     int checkOri, int *match) {
    return search_by_bow(nNodes, kf_off, kf_idx, f_off, f_idx, kf_valid, kf_keys, kf_desc, nF, f_keys, f_desc, nnratio, checkOri != 0, match);
}

// flat form of ygzo::TriangulationInput; cam2 = {fx, fy, cx, cy} of KF2
int yo_search_for_triangulation(int nNodes, const int *off1, const int *idx1, const int *off2, const int *idx2, int n1, const KeyPoint *keys1,
                                const uint8_t *desc1, const uint8_t *has_mp1, const float *uRight1, int n2, const KeyPoint *keys2,
                                const uint8_t *desc2, const uint8_t *has_mp2, const float *uRight2, int nlevels2, const float *scaleFactors2,
                                const float *levelSigma2_2, const float *F12, const float *Cw1, const float *R2w, const float *t2w,
                                const float *cam2, int onlyStereo, int checkOri, int *match12) {
    (void) nlevels2;
    TriangulationInput in{nNodes, off1, idx1, off2, idx2, n1, keys1, desc1, has_mp1, uRight1, n2, keys2, desc2, has_mp2, uRight2,
                          scaleFactors2, levelSigma2_2, F12, Cw1, R2w, t2w, cam2[0], cam2[1], cam2[2], cam2[3]};
    return search_for_triangulation(in, onlyStereo != 0, checkOri != 0, match12);
}

void yo_is_in_frustum(const yo_frame *F, int M, const float *world, const float *normal, const float *maxDistInv, const float *minDistInv,
                      const float *mfMaxDistance, const float *Rcw, const float *tcw, const float *Ow, float logScaleFactor, int nScaleLevels,
                      float viewingCosLimit, uint8_t *in_view, float *projX, float *projY, float *projXR, int *level, float *viewCos) {
    FrameView v = to_view(F);
    FrustumInput in;
    in.M = M;
    in.world = world;
    in.normal = normal;
    in.maxDistInv = maxDistInv;
    in.minDistInv = minDistInv;
    in.mfMaxDistance = mfMaxDistance;
    std::memcpy(in.Rcw, Rcw, 36);
    std::memcpy(in.tcw, tcw, 12);
    std::memcpy(in.Ow, Ow, 12);
    in.logScaleFactor = logScaleFactor;
    in.nScaleLevels = nScaleLevels;
    is_in_frustum(v, in, viewingCosLimit, in_view, projX, projY, projXR, level, viewCos);
}

// MapPoint::PredictScale (src/MapPoint.cc:359-373) for an array of ratios = mfMaxDistance / dist
void yo_predict_scale(const float *ratio, int n, float logScaleFactor, int nScaleLevels, int *level) {
    for (int i = 0; i < n; i++) {
        int nScale = (int) std::ceil(std::log(ratio[i]) / logScaleFactor);
        if (nScale < 0) nScale = 0;
        else if (nScale >= nScaleLevels) nScale = nScaleLevels - 1;
        level[i] = nScale;
    }
}

void yo_distinctive_descriptors(int nPoints, const int *obs_off, const uint8_t *desc, int *best) {
    for (int p = 0; p < nPoints; p++) {
        const int n = obs_off[p + 1] - obs_off[p];
        best[p] = n > 0 ? distinctive_descriptor(desc + 32 * (size_t) obs_off[p], n) : -1;
    }
}

int yo_search_for_initialization(const yo_frame *F1, const yo_frame *F2, float *prevMatchedXY, int windowSize,
                                 float nnratio, int checkOri, int *matches12) {
    FrameView v1 = to_view(F1), v2 = to_view(F2);
    Grid g;
    g.Assign(v2);
    return search_for_initialization(v1, v2, g, prevMatchedXY, windowSize, nnratio, checkOri != 0, matches12);
}

// ---- SE3 / aligner ------------------------------------------------------------------------------------------
void yo_se3_exp(const float a[6], float out7[7]) {
    SE3f r = SE3f::Exp(a);
    std::memcpy(out7, r.q, 16);
    std::memcpy(out7 + 4, r.t, 12);
}
void yo_se3_mul(const float a7[7], const float b7[7], float out7[7]) {
    SE3f a, b;
    std::memcpy(a.q, a7, 16); std::memcpy(a.t, a7 + 4, 12);
    std::memcpy(b.q, b7, 16); std::memcpy(b.t, b7 + 4, 12);
    SE3f r = a.Mul(b);
    std::memcpy(out7, r.q, 16);
    std::memcpy(out7 + 4, r.t, 12);
}
void yo_se3_inverse(const float a7[7], float out7[7]) {
    SE3f a;
    std::memcpy(a.q, a7, 16); std::memcpy(a.t, a7 + 4, 12);
    SE3f r = a.Inverse();
    std::memcpy(out7, r.q, 16);
    std::memcpy(out7 + 4, r.t, 12);
}
void yo_se3_act(const float a7[7], const float p3[3], float out3[3]) {
    SE3f a;
    std::memcpy(a.q, a7, 16);
    std::memcpy(a.t, a7 + 4, 12);
    a.Act(p3, out3);
}

struct yo_align_frame {
    int N;
    const KeyPoint *keys;
    const uint8_t *mp_valid, *outlier;
    const float *mp_world;
    float Tcw[7];               // qx qy qz qw tx ty tz
    int nlevels;
    const uint8_t *const *levels;  // tight u8 images
    const int *level_w, *level_h;
    const float *invScaleFactors;
    float fx, fy, cx, cy;
};

// returns n_meas/16; out7 = TCR; info[0] = total linearisations, info[1] = chi2
size_t yo_sparse_img_align_mode(const yo_align_frame *ref, const yo_align_frame *cur, int max_level, int min_level, int n_iter,
                                float out7[7], float info[2], float H36[36], int device_order) {
    std::vector<Image> rimgs(ref->nlevels), cimgs(cur->nlevels);
    AlignFrame R, C;
    auto fill = [](const yo_align_frame *f, std::vector<Image> &imgs, AlignFrame &A) {
        A.N = f->N;
        A.keys = f->keys;
        A.mp_valid = f->mp_valid;
        A.outlier = f->outlier;
        A.mp_world = f->mp_world;
        std::memcpy(A.Tcw.q, f->Tcw, 16);
        std::memcpy(A.Tcw.t, f->Tcw + 4, 12);
        for (int l = 0; l < f->nlevels; l++) {
            imgs[l] = Image(f->level_w[l], f->level_h[l]);
            std::memcpy(imgs[l].d.data(), f->levels[l], imgs[l].d.size());
            A.pyramid.push_back(&imgs[l]);
        }
        A.invScaleFactors = f->invScaleFactors;
        A.fx = f->fx; A.fy = f->fy; A.cx = f->cx; A.cy = f->cy;
    };
    fill(ref, rimgs, R);
    fill(cur, cimgs, C);
    AlignResult r = sparse_img_align(R, C, max_level, min_level, n_iter, device_order != 0);
    std::memcpy(out7, r.TCR.q, 16);
    std::memcpy(out7 + 4, r.TCR.t, 12);
    if (info) { info[0] = (float) r.iters_total; info[1] = r.chi2; }
    if (H36) std::memcpy(H36, r.H, sizeof(r.H));
    return r.ret;
}
// fp64 evaluation of the same algorithm (sparse_img_align_f64): out7 / info in double
size_t yo_sparse_img_align_f64(const yo_align_frame *ref, const yo_align_frame *cur, int max_level, int min_level, int n_iter, double out7[7],
                               double info[2]) {
    std::vector<Image> rimgs(ref->nlevels), cimgs(cur->nlevels);
    AlignFrame R, C;
    auto fill = [](const yo_align_frame *f, std::vector<Image> &imgs, AlignFrame &A) {
        A.N = f->N; A.keys = f->keys; A.mp_valid = f->mp_valid; A.outlier = f->outlier; A.mp_world = f->mp_world;
        std::memcpy(A.Tcw.q, f->Tcw, 16);
        std::memcpy(A.Tcw.t, f->Tcw + 4, 12);
        for (int l = 0; l < f->nlevels; l++) {
            imgs[l] = Image(f->level_w[l], f->level_h[l]);
            std::memcpy(imgs[l].d.data(), f->levels[l], imgs[l].d.size());
            A.pyramid.push_back(&imgs[l]);
        }
        A.invScaleFactors = f->invScaleFactors;
        A.fx = f->fx; A.fy = f->fy; A.cx = f->cx; A.cy = f->cy;
    };
    fill(ref, rimgs, R);
    fill(cur, cimgs, C);
    const AlignResultF64 r = sparse_img_align_f64(R, C, max_level, min_level, n_iter);
    std::memcpy(out7, r.T, sizeof r.T);
    if (info) { info[0] = (double) r.iters_total; info[1] = r.chi2; }
    return r.ret;
}
size_t yo_sparse_img_align(const yo_align_frame *ref, const yo_align_frame *cur, int max_level, int min_level, int n_iter,
                           float out7[7], float info[2], float H36[36]) {
    return yo_sparse_img_align_mode(ref, cur, max_level, min_level, n_iter, out7, info, H36, 0);
}

// Frame::ComputeStereoMatches on two images: both pyramids are computed with extractor e (as the two ORBextractor instances do)
void yo_compute_stereo_matches(void *e_, const uint8_t *imgL, const uint8_t *imgR, int w, int h, int N, const KeyPoint *keysL,
                               const uint8_t *descL, int Nr, const KeyPoint *keysR, const uint8_t *descR, float mb, float mbf, float *uRight,
                               float *depth) {
    Extractor *e = (Extractor *) e_;
    e->ComputePyramid(imgL, w, h, w);
    std::vector<Image> L = e->mvImagePyramid;
    e->ComputePyramid(imgR, w, h, w);
    std::vector<Image> R = e->mvImagePyramid;
    std::vector<const Image *> pl, pr;
    for (int l = 0; l < e->nlevels; l++) { pl.push_back(&L[l]); pr.push_back(&R[l]); }
    compute_stereo_matches(N, keysL, descL, Nr, keysR, descR, pl, pr, e->mvScaleFactor.data(), e->mvInvScaleFactor.data(), mb, mbf, uRight, depth);
}

// FindDirectProjection for n candidates: reference KeyFrame images ref_imgs[ref_slot[i]] (all w x h), one current image; every pyramid is
// computed with extractor e.  px_curr (n x 2) in/out, search_level / success out, patches (n x 100, may be null) = _patch_with_border.
void yo_find_direct_projection_batch(void *e_, int n_refs, const uint8_t *const *ref_imgs, const uint8_t *cur_img, int w, int h,
                                     const float *cur_Tcw7, float fx, float fy, float cx, float cy, int n, const int *ref_slot,
                                     const float *ref_Tcw7, const KeyPoint *ref_kp, const float *mp_world, float *px_curr, int *search_level,
                                     uint8_t *success, uint8_t *patches) {
    Extractor *e = (Extractor *) e_;
    std::vector<std::vector<Image>> refs(n_refs);
    for (int r = 0; r < n_refs; r++) {
        e->ComputePyramid(ref_imgs[r], w, h, w);
        refs[r] = e->mvImagePyramid;
    }
    e->ComputePyramid(cur_img, w, h, w);
    std::vector<Image> cur = e->mvImagePyramid;
    DirectCur C;
    std::memcpy(C.Tcw.q, cur_Tcw7, 16);
    std::memcpy(C.Tcw.t, cur_Tcw7 + 4, 12);
    for (int l = 0; l < e->nlevels; l++) C.pyramid.push_back(&cur[l]);
    C.scaleFactors = e->mvScaleFactor.data();
    C.invScaleFactors = e->mvInvScaleFactor.data();
    C.fx = fx; C.fy = fy; C.cx = cx; C.cy = cy;
    for (int i = 0; i < n; i++) {
        DirectRef R;
        R.kp = ref_kp[i];
        std::memcpy(R.Tcw.q, ref_Tcw7 + 7 * i, 16);
        std::memcpy(R.Tcw.t, ref_Tcw7 + 7 * i + 4, 12);
        R.level_img = &refs[ref_slot[i]][R.kp.octave];
        R.scaleFactors = e->mvScaleFactor.data();
        R.invLevelSigma2_1 = e->mvInvLevelSigma2[e->nlevels > 1 ? 1 : 0];
        R.nlevels = e->nlevels;
        R.fx = fx; R.fy = fy; R.cx = cx; R.cy = cy;
        success[i] = find_direct_projection(R, C, mp_world + 3 * i, px_curr + 2 * i, &search_level[i], patches ? patches + 100 * (size_t) i : nullptr);
    }
}

// ---- cpu_baseline helper: extract + frame-to-frame projection match over a list of frames, `threads` workers ----
// Frames are u8 images of identical size laid out back to back.  Frame f (f >= 1) is matched against frame f-1
// with an identity relative pose and unit-depth back-projected points (SURVEY §8d metric definition).
// Returns elapsed seconds; n_kp_total / n_match_total are checksums so the work cannot be optimised away.
// mode: bit 0 SearchByProjection(cur, last) of every frame against its predecessor; bit 1 SparseImgAlign(nlevels - 1, 1, 10 iterations) of the same
// pair (features = the predecessor's keypoints with MapPoints at their unit-depth back-projection, both poses identity: what ygzf_align_batch_prev
// computes); bit 2 the clip holds (left, right) pairs: Frame::ComputeStereoMatches on frames (2p, 2p + 1) after their extraction, matching only
// between left frames.  The CPU side of bench.py's per-workload cpu_baseline (test infrastructure).
double yo_bench_extract_match(int nfeatures, float scaleFactor, int nlevels, int iniTh, int minTh, const uint8_t *frames,
                              int nframes, int w, int h, int threads, int frames_per_thread, double max_seconds, float fx, float fy,
                              float cx, float cy, int mode, long *n_kp_total, long *n_match_total, long *n_frames_done) {
    std::vector<long> kp_acc(threads, 0), m_acc(threads, 0), f_acc(threads, 0);
    const bool doMatch = mode & 1, doAlign = (mode & 2) != 0, stereo = (mode & 4) != 0;
    auto t0 = std::chrono::steady_clock::now();
    auto worker = [&](int tid) {
        Extractor ex(nfeatures, scaleFactor, nlevels, iniTh, minTh), exR(nfeatures, scaleFactor, nlevels, iniTh, minTh);
        std::vector<KeyPoint> kprev, kcur, kright;
        std::vector<uint8_t> dprev, dcur, dright;
        std::vector<Image> pyrPrev;
        // each worker walks `frames_per_thread` consecutive frames of the clip (wrapping around), starting at its own offset,
        // so that the t-1 -> t pairs stay on one worker and every worker does the same amount of work
        const int step = stereo ? 2 : 1;
        const int start = (int) (((long) tid * 8) % nframes) & ~(step - 1);
        for (int j = 0; j < frames_per_thread; j += step) {
            // time-bounded sample: stop at the deadline (the host may give this process far fewer cores than it shows)
            if (max_seconds > 0 && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > max_seconds) break;
            const int f = (start + j) % nframes;
            ex.Extract(frames + (size_t) f * w * h, w, h, w, kcur, dcur);
            kp_acc[tid] += (long) kcur.size();
            if (stereo) {
                const int fr = (f + 1) % nframes;
                exR.Extract(frames + (size_t) fr * w * h, w, h, w, kright, dright);
                kp_acc[tid] += (long) kright.size();
                std::vector<const Image *> pl, pr;
                for (int l = 0; l < nlevels; l++) { pl.push_back(&ex.mvImagePyramid[l]); pr.push_back(&exR.mvImagePyramid[l]); }
                std::vector<float> ur(kcur.size() + 1), dp(kcur.size() + 1);
                compute_stereo_matches((int) kcur.size(), kcur.data(), dcur.data(), (int) kright.size(), kright.data(), dright.data(), pl, pr,
                                       ex.mvScaleFactor.data(), ex.mvInvScaleFactor.data(), 0.11f, 47.9f, ur.data(), dp.data());
            }
            if (j > 0 && !kprev.empty() && !kcur.empty() && (doMatch || doAlign)) {
                int n = (int) kprev.size();
                std::vector<uint8_t> ones(n, 1), zeros(n, 0);
                std::vector<float> world((size_t) 3 * n);
                for (int i = 0; i < n; i++) {
                    world[3 * i] = (kprev[i].x - cx) / fx;
                    world[3 * i + 1] = (kprev[i].y - cy) / fy;
                    world[3 * i + 2] = 1.f;
                }
                if (doMatch) {
                    FrameView cur;
                    cur.N = (int) kcur.size();
                    cur.keys = kcur.data();
                    cur.desc = dcur.data();
                    cur.uRight = nullptr;
                    cur.minX = 0; cur.minY = 0; cur.maxX = (float) w; cur.maxY = (float) h;
                    cur.gridInvW = (float) Grid::COLS / (cur.maxX - cur.minX);
                    cur.gridInvH = (float) Grid::ROWS / (cur.maxY - cur.minY);
                    cur.fx = fx; cur.fy = fy; cur.cx = cx; cur.cy = cy; cur.mb = 0; cur.mbf = 0;
                    cur.scaleFactors = ex.mvScaleFactor.data();
                    cur.nlevels = nlevels;
                    Grid g;
                    g.Assign(cur);
                    ProjLastInput in;
                    in.N = n;
                    in.keys = kprev.data();
                    in.mp_valid = ones.data();
                    in.outlier = zeros.data();
                    in.mp_has_obs = ones.data();
                    in.mp_world = world.data();
                    in.mp_desc = dprev.data();
                    const float I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, z[3] = {0, 0, 0};
                    std::memcpy(in.Rcw, I, 36); std::memcpy(in.tcw, z, 12);
                    std::memcpy(in.Rlw, I, 36); std::memcpy(in.tlw, z, 12);
                    std::vector<uint8_t> owner(cur.N, 0);
                    std::vector<int> match(cur.N, -1);
                    m_acc[tid] += search_by_projection_last(cur, g, in, 15.f, true, true, true, owner.data(), match.data());
                }
                if (doAlign && nlevels >= 2 && (int) pyrPrev.size() == nlevels) {
                    AlignFrame R, C;
                    R.N = n; R.keys = kprev.data(); R.mp_valid = ones.data(); R.outlier = zeros.data(); R.mp_world = world.data();
                    for (int l = 0; l < nlevels; l++) { R.pyramid.push_back(&pyrPrev[l]); C.pyramid.push_back(&ex.mvImagePyramid[l]); }
                    R.invScaleFactors = C.invScaleFactors = ex.mvInvScaleFactor.data();
                    R.fx = C.fx = fx; R.fy = C.fy = fy; R.cx = C.cx = cx; R.cy = C.cy = cy;
                    const AlignResult ar = sparse_img_align(R, C, nlevels - 1, 1, 10);
                    m_acc[tid] += doMatch ? 0 : (long) ar.ret;
                }
            }
            if (doAlign) pyrPrev = ex.mvImagePyramid;
            kprev.swap(kcur);
            dprev.swap(dcur);
            f_acc[tid] += step;
        }
    };
    std::vector<std::thread> th;
    for (int t = 0; t < threads; t++) th.emplace_back(worker, t);
    for (auto &t : th) t.join();
    double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    long a = 0, b = 0, fd = 0;
    for (int t = 0; t < threads; t++) { a += kp_acc[t]; b += m_acc[t]; fd += f_acc[t]; }
    if (n_frames_done) *n_frames_done = fd;
    if (n_kp_total) *n_kp_total = a;
    if (n_match_total) *n_match_total = b;
    return sec;
}

}  // extern "C"
