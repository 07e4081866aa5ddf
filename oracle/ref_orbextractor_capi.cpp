// oracle/ref_orbextractor_capi.cpp -- TEST INFRASTRUCTURE: C entry points around the REFERENCE's ygz::ORBextractor, compiled from
// /root/reference/src/ORBextractor.cc where it lies (oracle/Makefile target `ref_extractor`, output oracle/_ref/libref_orbextractor.so)
// against the OpenCV stand-in of oracle/ref_shim/.  tests/test_ref_extractor.py compares it with the oracle's restatement.
//
// Monotone allocator.  DistributeOctTree sorts pair<int, ExtractorNode*> (src/ORBextractor.cc:653-657): equal node sizes are ordered by
// HEAP ADDRESS, so the reference's keypoint order -- and, through the early break at N nodes, sometimes its keypoint set -- depends on
// the allocator.  The oracle defines that tie-break as creation order.  Inside this library operator new is a bump
// allocator over one reserved address range (never reuses memory, addresses grow with allocation order), which makes the reference's
// own code take exactly that order without touching a line of it.  The symbols are hidden: nothing outside this .so sees them.
#include <sys/mman.h>

#include <atomic>
#include <cstddef>
#include <cstdlib>
#include <new>

namespace {
constexpr size_t kArenaBytes = 64ull << 30;   // address space only (MAP_NORESERVE); pages are touched as used and dropped on reset
unsigned char *g_arena = nullptr;
std::atomic<size_t> g_top{0};
std::atomic<long> g_live_handles{0};
void *arena_alloc(size_t n) {
    if (!g_arena) {
        void *p = mmap(nullptr, kArenaBytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
        if (p == MAP_FAILED) std::abort();
        g_arena = (unsigned char *) p;
    }
    const size_t at = g_top.fetch_add((n + 63) & ~(size_t) 63);
    if (at + n > kArenaBytes) std::abort();
    return g_arena + at;
}
void arena_reset_if_idle() {   // between independent calls (no extractor handle alive) the whole arena is dropped
    if (g_live_handles.load() == 0 && g_arena && g_top.load() > 0) {
        madvise(g_arena, (g_top.load() + 4095) & ~(size_t) 4095, MADV_DONTNEED);
        g_top.store(0);
    }
}
}  // namespace
#define YR_HIDDEN __attribute__((visibility("hidden")))
YR_HIDDEN void *operator new(size_t n) { return arena_alloc(n); }
YR_HIDDEN void *operator new[](size_t n) { return arena_alloc(n); }
YR_HIDDEN void operator delete(void *) noexcept {}
YR_HIDDEN void operator delete[](void *) noexcept {}
YR_HIDDEN void operator delete(void *, size_t) noexcept {}
YR_HIDDEN void operator delete[](void *, size_t) noexcept {}

#include <opencv2/core/core.hpp>

#define protected public   // test infrastructure only: ComputeKeyPointsDSO (the multi-level grid detector) is a protected member whose only call
#include "ORBextractor.h"  // site in the reference is commented out (src/ORBextractor.cc:1053); the pin calls it directly
#undef protected
#include "Frame.h"

static cv::Mat wrap(const uint8_t *img, int w, int h, int stride) {
    cv::Mat m(h, w, CV_8UC1);
    for (int y = 0; y < h; y++) std::memcpy(m.ptr(y), img + (size_t) y * stride, (size_t) w);
    return m;
}

// kp: n x 7 floats (x, y, size, angle, response, octave, class_id); desc: n x 32 bytes.  Returns the keypoint count (<= cap) or -1.
static int emit(const std::vector<cv::KeyPoint> &kps, const cv::Mat &desc, float *kp, uint8_t *d, int cap) {
    const int n = (int) kps.size();
    if (n > cap) return -1;
    for (int i = 0; i < n; i++) {
        const cv::KeyPoint &k = kps[i];
        float *o = kp + 7 * (size_t) i;
        o[0] = k.pt.x; o[1] = k.pt.y; o[2] = k.size; o[3] = k.angle; o[4] = k.response; o[5] = (float) k.octave; o[6] = (float) k.class_id;
        if (n && !desc.empty()) std::memcpy(d + 32 * (size_t) i, desc.ptr(i), 32);
    }
    return n;
}

extern "C" {

// ORBextractor::operator()(image, mask, keypoints, descriptors)      src/ORBextractor.cc:962-1028
int yr_extract(const uint8_t *img, int w, int h, int stride, int nfeatures, float scale_factor, int nlevels, int ini_th, int min_th, float *kp,
               uint8_t *desc, int cap) {
    arena_reset_if_idle();
    int n;
    {
        ygz::ORBextractor ex(nfeatures, scale_factor, nlevels, ini_th, min_th);
        std::vector<cv::KeyPoint> kps;
        cv::Mat d;
        ex(wrap(img, w, h, stride), cv::Mat(), kps, d);
        n = emit(kps, d, kp, desc, cap);
    }
    return n;
}

// pyramid level `level` after ComputePyramid (tight copy); returns 0 and the level size
int yr_pyramid_level(const uint8_t *img, int w, int h, int stride, float scale_factor, int nlevels, int level, uint8_t *out, int *lw, int *lh) {
    arena_reset_if_idle();
    ygz::ORBextractor ex(1000, scale_factor, nlevels, 20, 7);
    ex.ComputePyramid(wrap(img, w, h, stride));
    const cv::Mat &m = ex.mvImagePyramid[level];
    *lw = m.cols; *lh = m.rows;
    if (out)
        for (int y = 0; y < m.rows; y++) std::memcpy(out + (size_t) y * m.cols, m.ptr(y), (size_t) m.cols);
    return 0;
}

// ORBextractor::operator()(Frame*, keypoints, descriptors, method, leftEye = true) on a frame that already holds n_existing keys
// (x, y, size, angle, response, octave, class_id as 7 floats each).  method: 0 ORBSLAM_KEYPOINT, 2 DSO_KEYPOINT.   :1031-1127
// The extractor object persists across calls of one handle (mnGridSize of the DSO detector is state): create / call / destroy.
void *yr_frame_extractor_create(int nfeatures, float scale_factor, int nlevels, int ini_th, int min_th) {
    arena_reset_if_idle();
    g_live_handles++;
    return new ygz::ORBextractor(nfeatures, scale_factor, nlevels, ini_th, min_th);
}
void yr_frame_extractor_destroy(void *h) {
    delete (ygz::ORBextractor *) h;
    g_live_handles--;
}
int yr_frame_extract(void *h, const uint8_t *img, int w, int h_, int stride, int method, const float *existing, int n_existing, float *kp, uint8_t *desc,
                     int cap) {
    ygz::ORBextractor &ex = *(ygz::ORBextractor *) h;
    ygz::Frame frame;
    ex.ComputePyramid(wrap(img, w, h_, stride));          // Frame::ComputeImagePyramid: the frame owns clones of the levels (src/Frame.cc:797-815)
    for (const cv::Mat &m : ex.mvImagePyramid) frame.mvImagePyramid.push_back(m.clone());
    for (int i = 0; i < n_existing; i++) {
        const float *e = existing + 7 * (size_t) i;
        cv::KeyPoint k(e[0], e[1], e[2], e[3], e[4], (int) e[5], (int) e[6]);
        frame.mvKeys.push_back(k);
    }
    frame.N = n_existing;
    cv::Mat d;
    // Frame::ExtractFeatures passes the frame's own key vector as the output: (*mpORBextractorLeft)(this, mvKeys, mDescriptors, method, true)
    ex(&frame, frame.mvKeys, d, (ygz::ORBextractor::KeyPointMethod) method, true);
    return emit(frame.mvKeys, d, kp, desc, cap);
}

// ORBextractor::ComputeKeyPointsDSO (:1388-1507) on the pyramid of `img`: new keypoints of all levels in LEVEL coordinates (level order,
// then the function's own order), the re-oriented angles of the existing keys, mnGridSize after the call.
int yr_dso_multilevel(void *h, const uint8_t *img, int w, int h_, int stride, float *existing, int n_existing, float *kp, int cap, int *grid_size) {
    ygz::ORBextractor &ex = *(ygz::ORBextractor *) h;
    ex.ComputePyramid(wrap(img, w, h_, stride));
    for (cv::Mat &m : ex.mvImagePyramid) m = m.clone();   // what the Frame overload installs (:1039): the Frame's border-less clones (src/Frame.cc:812);
                                                          // the detector passes `cols` as the row stride (:1437)
    std::vector<cv::KeyPoint> exist;
    for (int i = 0; i < n_existing; i++) {
        const float *e = existing + 7 * (size_t) i;
        exist.push_back(cv::KeyPoint(e[0], e[1], e[2], e[3], e[4], (int) e[5], (int) e[6]));
    }
    std::vector<std::vector<cv::KeyPoint>> all;
    ex.ComputeKeyPointsDSO(all, exist);
    for (int i = 0; i < n_existing; i++) existing[7 * (size_t) i + 3] = exist[i].angle;
    std::vector<cv::KeyPoint> flat;
    for (auto &v : all) flat.insert(flat.end(), v.begin(), v.end());
    *grid_size = ex.mnGridSize;
    return emit(flat, cv::Mat(), kp, nullptr, cap);
}

}  // extern "C"
