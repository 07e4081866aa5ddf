// ygzf_mgpu.hip -- the multi-GPU split of SURVEY 8(e) inside the product: frames (or indivisible units of consecutive frames: frame pairs,
// stereo pairs) are dealt round-robin over the devices, one host thread + one context (own HIP stream, own buffers) + page-locked staging per
// device, results concatenated on the host in input order.  Frames are independent, a unit's frames stay on one device, so nothing crosses
// between devices: no collective, no peer copy, no RCCL.  Host code over the public C ABI of this library (include/ygzf.h).
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <string>
#include <thread>
#include <vector>

#include "../../include/ygzf.h"

struct ygzf_mgpu {
    struct Dev {
        int device = 0;
        ygzf_ctx *ctx = nullptr;
        uint8_t *hIn = nullptr;        // page-locked: this device's frames of one call, tight
        ygzf_kp *hKp = nullptr;        // page-locked: results of one call
        uint8_t *hDesc = nullptr;
        int *hCnt = nullptr;
        int *hMatch = nullptr;         // page-locked: match rows of one call
        int rc = 0;
        std::string err;
    };
    std::vector<Dev> devs;
    int maxW = 0, maxH = 0, maxFrames = 0, stride = 0;
    int copyThreads = 4;           // host threads per device for the gather / scatter copies (YGZF_MGPU_COPY_THREADS)
    std::string err;
};

static int mfail(ygzf_mgpu *m, int rc, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (m) m->err = buf;
    return rc;
}

extern "C" {

int ygzf_mgpu_create(const int *devices, int n_devices, const ygzf_extractor_cfg *cfg, int max_width, int max_height, int max_frames_per_device,
                     ygzf_mgpu **out) {
    if (!devices || n_devices < 1 || !cfg || !out || max_frames_per_device < 1) return YGZF_ERR_INVALID;
    *out = nullptr;
    ygzf_mgpu *m = new ygzf_mgpu();
    m->maxW = max_width; m->maxH = max_height; m->maxFrames = max_frames_per_device;
    if (const char *e = getenv("YGZF_MGPU_COPY_THREADS")) m->copyThreads = std::max(1, std::min(32, atoi(e)));
    m->devs.resize(n_devices);
    for (int i = 0; i < n_devices; i++) {
        ygzf_mgpu::Dev &d = m->devs[i];
        d.device = devices[i];
        int rc = ygzf_create(devices[i], cfg, max_width, max_height, max_frames_per_device, &d.ctx);
        if (rc != YGZF_OK) {                      // e.g. YGZF_ERR_NO_DEVICE for an index past the last device: never a silent fall-back
            fprintf(stderr, "ygzf_mgpu_create: device %d: %s\n", devices[i], ygzf_last_error(nullptr));
            ygzf_mgpu_destroy(m);
            return rc;
        }
        if (i == 0) m->stride = ygzf_max_keypoints(d.ctx, max_width, max_height);
        const size_t F = (size_t) max_frames_per_device;
        if (hipSetDevice(devices[i]) != hipSuccess || hipHostMalloc((void **) &d.hIn, F * max_width * max_height) != hipSuccess ||
            hipHostMalloc((void **) &d.hKp, F * m->stride * sizeof(ygzf_kp)) != hipSuccess ||
            hipHostMalloc((void **) &d.hDesc, F * m->stride * 32) != hipSuccess || hipHostMalloc((void **) &d.hCnt, F * sizeof(int)) != hipSuccess ||
            hipHostMalloc((void **) &d.hMatch, F * m->stride * sizeof(int)) != hipSuccess) {
            ygzf_mgpu_destroy(m);
            return YGZF_ERR_HIP;
        }
    }
    *out = m;
    return YGZF_OK;
}

void ygzf_mgpu_destroy(ygzf_mgpu *m) {
    if (!m) return;
    for (auto &d : m->devs) {
        if (d.ctx) ygzf_destroy(d.ctx);
        if (!d.hIn && !d.hKp && !d.hDesc && !d.hCnt && !d.hMatch) continue;   // a slot whose context was never created (e.g. a device index that does not exist)
        (void) hipSetDevice(d.device);
        if (d.hIn) (void) hipHostFree(d.hIn);
        if (d.hKp) (void) hipHostFree(d.hKp);
        if (d.hDesc) (void) hipHostFree(d.hDesc);
        if (d.hCnt) (void) hipHostFree(d.hCnt);
        if (d.hMatch) (void) hipHostFree(d.hMatch);
    }
    delete m;
}

const char *ygzf_mgpu_last_error(const ygzf_mgpu *m) { return m ? m->err.c_str() : "null handle"; }
int ygzf_mgpu_device_count(const ygzf_mgpu *m) { return m ? (int) m->devs.size() : 0; }
int ygzf_mgpu_keypoint_stride(const ygzf_mgpu *m) { return m ? m->stride : 0; }
int ygzf_mgpu_slot_of_frame(const ygzf_mgpu *m, int frame, int unit) {
    if (!m || frame < 0 || unit < 1) return -1;
    return (frame / unit) % (int) m->devs.size();
}

int ygzf_mgpu_extract_match(ygzf_mgpu *m, const uint8_t *frames, int n_frames, int w, int h, int row_pitch, size_t frame_stride, int unit,
                            const ygzf_camera *cam, float th, int b_mono, int check_level, int check_orientation, ygzf_kp *kps, uint8_t *desc,
                            int *n_kp, int stride, int *match, int *nmatches) {
    if (!m || !frames || !kps || !desc || !n_kp) return mfail(m, YGZF_ERR_INVALID, "null argument");
    if (n_frames < 1 || unit < 1 || w < 1 || h < 1 || w > m->maxW || h > m->maxH || row_pitch < w || stride < m->stride)
        return mfail(m, YGZF_ERR_INVALID, "bad geometry (frames %d, unit %d, %dx%d, stride %d < %d)", n_frames, unit, w, h, stride, m->stride);
    if ((match || nmatches) && !cam) return mfail(m, YGZF_ERR_INVALID, "matching needs a camera");
    const int nd = (int) m->devs.size();
    // unit u -> device slot u % nd; a slot's frames keep their input order
    std::vector<std::vector<int>> mine(nd);
    for (int f = 0; f < n_frames; f++) mine[(f / unit) % nd].push_back(f);
    for (int s = 0; s < nd; s++)
        if ((int) mine[s].size() > m->maxFrames) return mfail(m, YGZF_ERR_INVALID, "%zu frames for device slot %d (maximum %d)", mine[s].size(), s, m->maxFrames);
    const bool doMatch = match != nullptr || nmatches != nullptr;
    // A device's frames go through in chunks: while the device works on chunk k the host gathers chunk k + 1 into page-locked memory and
    // scatters the results of chunk k - 1 (several copy threads: one thread moves ~10 GB/s, a 752x480 frame every 35 us).  The chunks of a
    // slot form one frame sequence for ygzf_match_batch_prev (its "previous frame" is carried from launch to launch), so the pairs are
    // those of one large batch.
    const int kChunk = 128;
    const int nCopy = m->copyThreads;
    auto parallel = [&](int n, const std::function<void(int)> &fn) {
        if (n <= 0) return;
        const int T = std::min(nCopy, n);
        std::vector<std::thread> ts;
        for (int t = 1; t < T; t++) ts.emplace_back([&, t] { for (int i = t; i < n; i += T) fn(i); });
        for (int i = 0; i < n; i += T) fn(i);
        for (auto &t : ts) t.join();
    };
    auto work = [&](int s) {
        ygzf_mgpu::Dev &d = m->devs[s];
        d.rc = YGZF_OK;
        const std::vector<int> &fr = mine[s];
        const int n = (int) fr.size();
        if (n == 0) return;
        const int nChunks = (n + kChunk - 1) / kChunk;
        auto lo = [&](int k) { return k * kChunk; };
        auto cnt = [&](int k) { return std::min(kChunk, n - k * kChunk); };
        auto gather = [&](int k) {   // into the page-locked staging area, tight rows
            parallel(cnt(k), [&](int j) {
                const int i = lo(k) + j;
                const uint8_t *src = frames + (size_t) fr[i] * frame_stride;
                uint8_t *dst = d.hIn + (size_t) i * h * w;
                if (row_pitch == w) memcpy(dst, src, (size_t) w * h);
                else for (int y = 0; y < h; y++) memcpy(dst + (size_t) y * w, src + (size_t) y * row_pitch, (size_t) w);
            });
        };
        auto launch = [&](int k) {
            int rc = ygzf_extract_batch_host(d.ctx, d.hIn + (size_t) lo(k) * h * w, cnt(k), w, h, w, (size_t) w * h);
            if (rc == YGZF_OK && doMatch) rc = ygzf_match_batch_prev(d.ctx, cam, th, b_mono, check_level, check_orientation);
            return rc;
        };
        std::vector<int> nm(n, 0);
        auto fetch = [&](int k) {
            const size_t o = (size_t) lo(k) * m->stride;
            int rc = ygzf_batch_fetch_all(d.ctx, d.hKp + o, d.hDesc + o * 32, d.hCnt + lo(k), m->stride);
            if (rc == YGZF_OK && doMatch) rc = ygzf_match_counts(d.ctx, nm.data() + lo(k));
            if (rc == YGZF_OK && match) rc = ygzf_match_fetch_all(d.ctx, d.hMatch + o, m->stride);
            return rc;
        };
        auto scatter = [&](int k) {   // to the caller's rows: input order
            parallel(cnt(k), [&](int j) {
                const int i = lo(k) + j, f = fr[i];
                n_kp[f] = d.hCnt[i];
                memcpy(kps + (size_t) f * stride, d.hKp + (size_t) i * m->stride, sizeof(ygzf_kp) * (size_t) d.hCnt[i]);
                memcpy(desc + (size_t) f * stride * 32, d.hDesc + (size_t) i * m->stride * 32, 32 * (size_t) d.hCnt[i]);
                if (!doMatch) return;
                const bool first = (f % unit) == 0;                           // no predecessor inside the unit
                if (nmatches) nmatches[f] = first ? -1 : nm[i];
                if (match) {
                    int *row = match + (size_t) f * stride;
                    if (first) for (int q = 0; q < stride; q++) row[q] = -1;
                    else {
                        memcpy(row, d.hMatch + (size_t) i * m->stride, sizeof(int) * (size_t) d.hCnt[i]);
                        for (int q = d.hCnt[i]; q < stride; q++) row[q] = -1;
                    }
                }
            });
        };
        gather(0);
        int rc = launch(0);
        for (int k = 0; k < nChunks && rc == YGZF_OK; k++) {
            if (k + 1 < nChunks) gather(k + 1);                  // the device works on chunk k meanwhile
            rc = fetch(k);
            if (rc == YGZF_OK && k + 1 < nChunks) rc = launch(k + 1);
            if (rc == YGZF_OK) scatter(k);                       // the device works on chunk k + 1 meanwhile
        }
        d.rc = rc;
        if (rc != YGZF_OK) d.err = ygzf_last_error(d.ctx);
    };
    std::vector<std::thread> th_;
    for (int s = 1; s < nd; s++) th_.emplace_back(work, s);               // one host thread per device; slot 0 on the calling thread
    work(0);
    for (auto &t : th_) t.join();
    for (int s = 0; s < nd; s++)
        if (m->devs[s].rc != YGZF_OK) return mfail(m, m->devs[s].rc, "device slot %d (device %d): %s", s, m->devs[s].device, m->devs[s].err.c_str());
    return YGZF_OK;
}

}  // extern "C"
