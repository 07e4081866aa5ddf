// ygzf_mgpu.hip -- the multi-GPU split of SURVEY 8(e) inside the product: frames (or indivisible units of consecutive frames: frame pairs,
// stereo pairs) are dealt round-robin over the devices, one host thread + one context (own HIP stream, own buffers) + page-locked staging per
// device, results concatenated on the host in input order.  Frames are independent, a unit's frames stay on one device, so nothing crosses
// between devices: no collective, no peer copy, no RCCL.  Host code over the public C ABI of this library (include/ygzf.h).
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>
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
        int rc = 0;
        std::string err;
    };
    std::vector<Dev> devs;
    int maxW = 0, maxH = 0, maxFrames = 0, stride = 0;
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
            hipHostMalloc((void **) &d.hDesc, F * m->stride * 32) != hipSuccess || hipHostMalloc((void **) &d.hCnt, F * sizeof(int)) != hipSuccess) {
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
        if (!d.hIn && !d.hKp && !d.hDesc && !d.hCnt) continue;   // a slot whose context was never created (e.g. a device index that does not exist)
        (void) hipSetDevice(d.device);
        if (d.hIn) (void) hipHostFree(d.hIn);
        if (d.hKp) (void) hipHostFree(d.hKp);
        if (d.hDesc) (void) hipHostFree(d.hDesc);
        if (d.hCnt) (void) hipHostFree(d.hCnt);
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
    auto work = [&](int s) {
        ygzf_mgpu::Dev &d = m->devs[s];
        d.rc = YGZF_OK;
        const std::vector<int> &fr = mine[s];
        const int n = (int) fr.size();
        if (n == 0) return;
        for (int i = 0; i < n; i++)                                       // gather into the page-locked staging area, tight rows
            for (int y = 0; y < h; y++) memcpy(d.hIn + ((size_t) i * h + y) * w, frames + (size_t) fr[i] * frame_stride + (size_t) y * row_pitch, (size_t) w);
        int rc = ygzf_extract_batch_host(d.ctx, d.hIn, n, w, h, w, (size_t) w * h);
        if (rc == YGZF_OK && doMatch) rc = ygzf_match_batch_prev(d.ctx, cam, th, b_mono, check_level, check_orientation);
        if (rc == YGZF_OK) rc = ygzf_batch_fetch_all(d.ctx, d.hKp, d.hDesc, d.hCnt, m->stride);
        std::vector<int> nm(n, 0);
        if (rc == YGZF_OK && doMatch) rc = ygzf_match_counts(d.ctx, nm.data());
        for (int i = 0; i < n && rc == YGZF_OK; i++) {                    // scatter to the caller's rows: input order
            const int f = fr[i];
            n_kp[f] = d.hCnt[i];
            memcpy(kps + (size_t) f * stride, d.hKp + (size_t) i * m->stride, sizeof(ygzf_kp) * (size_t) d.hCnt[i]);
            memcpy(desc + (size_t) f * stride * 32, d.hDesc + (size_t) i * m->stride * 32, 32 * (size_t) d.hCnt[i]);
            if (!doMatch) continue;
            const bool first = (f % unit) == 0;                           // no predecessor inside the unit
            if (nmatches) nmatches[f] = first ? -1 : nm[i];
            if (match) {
                int *row = match + (size_t) f * stride;
                if (first) for (int k = 0; k < stride; k++) row[k] = -1;
                else {
                    rc = ygzf_match_fetch(d.ctx, i, row, nullptr, stride);
                    for (int k = d.hCnt[i]; k < stride; k++) row[k] = -1;
                }
            }
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
