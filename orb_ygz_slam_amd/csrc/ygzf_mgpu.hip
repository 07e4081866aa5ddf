// ygzf_mgpu.hip -- the multi-GPU split of SURVEY 8(e) inside the product: frames (or indivisible units of consecutive frames: frame pairs,
// stereo pairs) are dealt round-robin over the devices, one host thread + two alternating contexts (own HIP streams, own buffers) + page-locked
// staging per device slot, results concatenated on the host in input order.  Frames are independent, a unit's frames stay on one device, so
// nothing crosses between devices: no collective, no peer copy, no RCCL.  Host code over the public C ABI of this library (include/ygzf.h).
#include <hip/hip_runtime.h>

#include <pthread.h>
#include <sched.h>

#include <cstdarg>
#include <cstdio>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <string>
#include <thread>
#include <vector>

#include "../../include/ygzf.h"
#include "ygzf_internal.h"   // the environment switches (YGZF_FORCE)

struct ygzf_mgpu {
    struct Dev {
        int device = 0;
        ygzf_ctx *ctx[2] = {nullptr, nullptr};   // chunks alternate between them: the upload of one overlaps the kernels of the other
        uint8_t *hIn[3] = {nullptr, nullptr, nullptr};   // page-locked: the frames of one chunk at the device's row pitch (used when the caller's frames are pageable);
                                                         // three, so that chunk k + 2 is gathered while chunk k's upload may still be reading its own
        uint8_t *hRes[2] = {nullptr, nullptr};   // page-locked: results of one chunk, ONE block [counts | keypoint rows | descriptor rows] ...
        size_t hResBytes = 0;
        ygzf_kp *hKp[2] = {nullptr, nullptr};    // ... into which these point: where the last fetch of the chunk's context left each part
        uint8_t *hDesc[2] = {nullptr, nullptr};
        int *hCnt[2] = {nullptr, nullptr};
        int resRow[2] = {0, 0};                  // entries per keypoint / descriptor row of that fetch
        int *hAux[2] = {nullptr, nullptr};       // page-locked: match rows (int) or uRight rows followed by depth rows (float) of one chunk
        cpu_set_t cpus;                          // CPUs of the device's NUMA node (empty: unknown, threads stay where they are)
        bool haveCpus = false;
        int rc = 0;
        std::string err;
    };
    std::vector<Dev> devs;
    int maxW = 0, maxH = 0, maxFrames = 0, stride = 0;
    int chunk = 0;                 // frames per chunk (the contexts' max_batch)
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

// CPUs of the NUMA node the device hangs on: /sys/bus/pci/devices/<bus id>/numa_node -> /sys/devices/system/node/node<N>/cpulist ("0-31,64-95")
static bool numa_cpus_of_device(int device, cpu_set_t *set) {
    char bus[64] = {0};
    if (hipDeviceGetPCIBusId(bus, sizeof bus, device) != hipSuccess) { (void) hipGetLastError(); return false; }
    for (char *p = bus; *p; p++) *p = (char) tolower(*p);
    char path[256];
    snprintf(path, sizeof path, "/sys/bus/pci/devices/%s/numa_node", bus);
    FILE *f = fopen(path, "r");
    if (!f) return false;
    int node = -1;
    const int got = fscanf(f, "%d", &node);
    fclose(f);
    if (got != 1 || node < 0) return false;
    snprintf(path, sizeof path, "/sys/devices/system/node/node%d/cpulist", node);
    f = fopen(path, "r");
    if (!f) return false;
    char list[4096] = {0};
    const bool ok = fgets(list, sizeof list, f) != nullptr;
    fclose(f);
    if (!ok) return false;
    CPU_ZERO(set);
    int n = 0;
    for (char *tok = strtok(list, ",\n"); tok; tok = strtok(nullptr, ",\n")) {
        int a = 0, b = 0;
        const int k = sscanf(tok, "%d-%d", &a, &b);
        if (k == 1) b = a;
        if (k < 1) continue;
        for (int c = a; c <= b && c < CPU_SETSIZE; c++) { CPU_SET(c, set); n++; }
    }
    return n > 0;
}

extern "C" {

int ygzf_mgpu_create(const int *devices, int n_devices, const ygzf_extractor_cfg *cfg, int max_width, int max_height, int max_frames_per_device,
                     ygzf_mgpu **out) {
    if (!devices || n_devices < 1 || !cfg || !out || max_frames_per_device < 1 || max_width < 1 || max_height < 1) return YGZF_ERR_INVALID;
    *out = nullptr;
    ygzf_mgpu *m = new ygzf_mgpu();
    m->maxW = max_width; m->maxH = max_height; m->maxFrames = max_frames_per_device;
    // (YGZF_FORCE keys of this entry point: mgpu_copy_threads, mgpu_chunk, mgpu_numa=0, mgpu_pinned=0 -- the tests make chunks small and treat
    // page-locked frames as pageable with them; ygzf_internal.h)
    m->copyThreads = (int) std::max(1l, std::min(32l, ygzf::forced("mgpu_copy_threads", m->copyThreads)));
    // frames per chunk: what a context and its staging are sized for -- 128 frames of 752x480, about 96 MB of level-0 pixels for larger images (46
    // frames of 1920x1080, 10 of 3840x2160), never more than the slot will be given
    const size_t px = (size_t) max_width * max_height;
    int chunk = (int) std::min<size_t>(128, std::max<size_t>(2, (96u << 20) / px));
    chunk = (int) std::max(1l, ygzf::forced("mgpu_chunk", chunk));
    m->chunk = std::max(1, std::min(chunk, max_frames_per_device));
    const bool numa = ygzf::forced("mgpu_numa", 1) != 0;
    m->devs.resize(n_devices);
    for (int i = 0; i < n_devices; i++) {
        ygzf_mgpu::Dev &d = m->devs[i];
        d.device = devices[i];
        CPU_ZERO(&d.cpus);
        for (int b = 0; b < 2; b++) {
            int rc = ygzf_create(devices[i], cfg, max_width, max_height, m->chunk, &d.ctx[b]);
            if (rc != YGZF_OK) {                  // e.g. YGZF_ERR_NO_DEVICE for an index past the last device: never a silent fall-back
                fprintf(stderr, "ygzf_mgpu_create: device %d: %s\n", devices[i], ygzf_last_error(nullptr));
                ygzf_mgpu_destroy(m);
                return rc;
            }
        }
        if (i == 0) m->stride = ygzf_max_keypoints(d.ctx[0], max_width, max_height);
        if (numa) d.haveCpus = numa_cpus_of_device(devices[i], &d.cpus);
        const size_t F = (size_t) m->chunk;
        if (hipSetDevice(devices[i]) != hipSuccess) { ygzf_mgpu_destroy(m); return YGZF_ERR_HIP; }
        // (the three input staging areas -- chunk x pitch x height bytes each, GBs for 4K -- are page-locked by the first call that brings PAGEABLE
        // frames, run_job: a caller whose frames are page-locked never pays for them)
        const size_t oK = (F * sizeof(int) + 255) & ~(size_t) 255, oD = oK + ((F * m->stride * sizeof(ygzf_kp) + 255) & ~(size_t) 255);
        d.hResBytes = oD + F * m->stride * 32;
        for (int b = 0; b < 2; b++) {
            if (hipHostMalloc((void **) &d.hRes[b], d.hResBytes) == hipSuccess) {
                d.hCnt[b] = (int *) d.hRes[b]; d.hKp[b] = (ygzf_kp *) (d.hRes[b] + oK); d.hDesc[b] = d.hRes[b] + oD;
            }
            if (!d.hRes[b] || hipHostMalloc((void **) &d.hAux[b], F * m->stride * sizeof(int)) != hipSuccess) {
                (void) hipGetLastError();
                ygzf_mgpu_destroy(m);
                return YGZF_ERR_HIP;
            }
        }
    }
    *out = m;
    return YGZF_OK;
}

void ygzf_mgpu_destroy(ygzf_mgpu *m) {
    if (!m) return;
    for (auto &d : m->devs) {
        for (int b = 0; b < 2; b++)
            if (d.ctx[b]) ygzf_destroy(d.ctx[b]);
        bool any = false;
        for (int b = 0; b < 2; b++) any = any || d.hRes[b] || d.hAux[b];
        for (int b = 0; b < 3; b++) any = any || d.hIn[b];
        if (!any) continue;   // a slot whose contexts were never created (e.g. a device index that does not exist)
        (void) hipSetDevice(d.device);
        for (int b = 0; b < 3; b++)
            if (d.hIn[b]) (void) hipHostFree(d.hIn[b]);
        for (int b = 0; b < 2; b++) {
            if (d.hRes[b]) (void) hipHostFree(d.hRes[b]);
            if (d.hAux[b]) (void) hipHostFree(d.hAux[b]);
        }
    }
    delete m;
}

void *ygzf_alloc_host(int device, size_t bytes) {
    void *p = nullptr;
    if (hipSetDevice(device) != hipSuccess || hipHostMalloc(&p, bytes ? bytes : 1) != hipSuccess) { (void) hipGetLastError(); return nullptr; }
    return p;
}
void ygzf_free_host(void *p) {
    if (p) (void) hipHostFree(p);
}

// What the node's DRAM can feed: n_threads host threads (bound to `device`'s NUMA node when device >= 0, like a slot's copy threads) each
// stream-read their own page-locked buffer of frames for `seconds`; no GPU work, no copy.  The transfers-included rate of one GPU reads its frames
// at the link's 55 GB/s -- eight of them want 440 GB/s from the host's memory at once, and this says whether the box has it.
int ygzf_host_stream_probe(int device, int n_threads, size_t bytes_per_thread, double seconds, double *gb_per_s) {
    if (n_threads < 1 || n_threads > 256 || bytes_per_thread < (1u << 20) || !(seconds > 0) || !gb_per_s) return YGZF_ERR_INVALID;
    cpu_set_t cpus;
    const bool bind = device >= 0 && numa_cpus_of_device(device, &cpus);
    std::vector<double> bytesDone((size_t) n_threads, 0.0);
    std::vector<unsigned long long> sink((size_t) n_threads, 0ull);
    std::vector<void *> bufs((size_t) n_threads, nullptr);
    std::atomic<int> ready{0};
    std::atomic<bool> go{false}, failed{false};
    const size_t words = bytes_per_thread / 8;
    auto t0 = std::chrono::steady_clock::now();
    auto worker = [&](int t) {
        if (bind) (void) pthread_setaffinity_np(pthread_self(), sizeof cpus, &cpus);
        void *p = nullptr;
        if (hipSetDevice(device >= 0 ? device : 0) != hipSuccess || hipHostMalloc(&p, words * 8) != hipSuccess) { (void) hipGetLastError(); failed = true; }
        bufs[(size_t) t] = p;
        if (p) memset(p, t + 1, words * 8);                      // first touch on this thread's node
        ready++;
        while (!go.load()) std::this_thread::yield();
        if (!p) return;
        const unsigned long long *q = (const unsigned long long *) p;
        unsigned long long a0 = 0, a1 = 0, a2 = 0, a3 = 0;
        double done = 0;
        while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds) {
            for (size_t i = 0; i + 8 <= words; i += 8) { a0 += q[i] + q[i + 4]; a1 += q[i + 1] + q[i + 5]; a2 += q[i + 2] + q[i + 6]; a3 += q[i + 3] + q[i + 7]; }
            done += (double) (words * 8);
        }
        bytesDone[(size_t) t] = done;
        sink[(size_t) t] = a0 + a1 + a2 + a3;
    };
    std::vector<std::thread> ts;
    for (int t = 0; t < n_threads; t++) ts.emplace_back(worker, t);
    while (ready.load() < n_threads) std::this_thread::yield();
    t0 = std::chrono::steady_clock::now();
    go = true;
    for (auto &t : ts) t.join();
    const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    double total = 0;
    unsigned long long s = 0;
    for (int t = 0; t < n_threads; t++) { total += bytesDone[(size_t) t]; s += sink[(size_t) t]; if (bufs[(size_t) t]) (void) hipHostFree(bufs[(size_t) t]); }
    *gb_per_s = (s == 0x5bd1e995ull ? 0.0 : 1.0) * total / sec / 1e9;   // (the sums are used, so the reads are not optimised away)
    return failed.load() ? YGZF_ERR_HIP : YGZF_OK;
}

int ygzf_bind_host_thread_to_device(int device) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) { (void) hipGetLastError(); return YGZF_ERR_NO_DEVICE; }
    if (device < 0 || device >= ndev) return YGZF_ERR_NO_DEVICE;
    cpu_set_t want, have, both;
    if (!numa_cpus_of_device(device, &want)) return 0;            // no NUMA information (one-socket box, container without sysfs): nothing to do
    if (sched_getaffinity(0, sizeof have, &have) != 0) return 0;
    CPU_AND(&both, &want, &have);                                 // never widen what the launcher (taskset, cgroup cpuset) allowed
    const int n = CPU_COUNT(&both);
    if (n == 0) return 0;
    if (sched_setaffinity(0, sizeof both, &both) != 0) return 0;
    return n;
}

const char *ygzf_mgpu_last_error(const ygzf_mgpu *m) { return m ? m->err.c_str() : "null handle"; }
int ygzf_mgpu_device_count(const ygzf_mgpu *m) { return m ? (int) m->devs.size() : 0; }
int ygzf_mgpu_keypoint_stride(const ygzf_mgpu *m) { return m ? m->stride : 0; }
int ygzf_mgpu_chunk_frames(const ygzf_mgpu *m) { return m ? m->chunk : 0; }
int ygzf_mgpu_slot_of_frame(const ygzf_mgpu *m, int frame, int unit) {
    if (!m || frame < 0 || unit < 1) return -1;
    return (frame / unit) % (int) m->devs.size();
}

}  // extern "C"

namespace {
enum Mode { kExtractOnly, kMatch, kStereo };
struct Job {
    Mode mode;
    const uint8_t *frames; int n_frames, w, h, row_pitch; size_t frame_stride; int unit;
    const ygzf_camera *cam; float th; int b_mono, check_level, check_orientation;
    float mb, mbf;
    ygzf_kp *kps; uint8_t *desc; int *n_kp; int stride;
    int *match, *nmatches;
    float *u_right, *depth;
};

int run_job(ygzf_mgpu *m, const Job &J) {
    const int nd = (int) m->devs.size();
    const int w = J.w, h = J.h, unit = J.unit;
    // unit u -> device slot u % nd; a slot's frames keep their input order
    std::vector<std::vector<int>> mine(nd);
    for (int f = 0; f < J.n_frames; f++) mine[(f / unit) % nd].push_back(f);
    for (int s = 0; s < nd; s++)
        if ((int) mine[s].size() > m->maxFrames) return mfail(m, YGZF_ERR_INVALID, "%zu frames for device slot %d (maximum %d)", mine[s].size(), s, m->maxFrames);
    // Units that fit a chunk: whole units per chunk, chunks alternate between the slot's two contexts (nothing is carried from chunk to chunk, so the
    // upload of one overlaps the kernels of the other).  Longer units (a clip whose every frame is matched against its predecessor): all chunks on
    // ONE context, whose carried "previous frame" links frame k * chunk to frame k * chunk - 1, one chunk after the other.
    const bool alternate = unit <= m->chunk;
    const int chunk = alternate ? (m->chunk / unit) * unit : m->chunk;
    // a (left, right) pair never straddles two chunks: ygzf_stereo_batch pairs frames 2p / 2p + 1 of ONE launch, and the aux staging rows are per pair
    if (J.mode == kStereo && (!alternate || (chunk & 1)))
        return mfail(m, YGZF_ERR_INVALID, "stereo pairs need chunks of at least 2 frames (chunk %d: max_frames_per_device / YGZF_FORCE=mgpu_chunk)", m->chunk);
    // frames in page-locked host memory go to the device from where they lie
    bool pinned = false;
    {
        hipPointerAttribute_t at;
        memset(&at, 0, sizeof at);
        if (hipPointerGetAttributes(&at, J.frames) == hipSuccess) pinned = at.type == hipMemoryTypeHost;
        else (void) hipGetLastError();                // an ordinary malloc pointer is "invalid value" to the runtime: pageable
        pinned = pinned && ygzf::forced("mgpu_pinned", 1) != 0;
    }
    const int nCopy = m->copyThreads;
    auto work = [&](int s, bool ownThread) {
        ygzf_mgpu::Dev &d = m->devs[s];
        d.rc = YGZF_OK;
        const std::vector<int> &fr = mine[s];
        const int n = (int) fr.size();
        if (n == 0) return;
        // (never the caller's own thread: its affinity is the caller's business)
        if (ownThread && d.haveCpus) (void) pthread_setaffinity_np(pthread_self(), sizeof d.cpus, &d.cpus);   // copy threads inherit it
        auto parallel = [&](int cnt, const std::function<void(int)> &fn) {
            if (cnt <= 0) return;
            const int T = std::min(nCopy, cnt);
            std::vector<std::thread> ts;
            for (int t = 1; t < T; t++) ts.emplace_back([&, t] { for (int i = t; i < cnt; i += T) fn(i); });
            for (int i = 0; i < cnt; i += T) fn(i);
            for (auto &t : ts) t.join();
        };
        const int nChunks = (n + chunk - 1) / chunk;
        auto lo = [&](int k) { return k * chunk; };
        auto cnt = [&](int k) { return std::min(chunk, n - k * chunk); };
        std::vector<const uint8_t *> ptrs((size_t) chunk);
        std::vector<int> nm(n, 0);
        auto cx = [&](int k) { return d.ctx[alternate ? (k & 1) : 0]; };
        // only the matcher reads the carried "previous frame": extraction-only and stereo jobs leave it out (one launch less in front of every upload)
        for (ygzf_ctx *q : d.ctx) (void) ygzf_set_carry_previous(q, J.mode == kMatch ? 1 : 0);
        const int sp = ygzf_host_row_pitch(w);   // the staging area carries the device's row pitch: a chunk goes up as whole frames, not row by row
        if (!pinned && !d.hIn[0]) {              // first pageable job of this slot: page-lock its three input staging areas (sized for the handle's maxima)
            const size_t bytes = (size_t) m->chunk * (size_t) ygzf_host_row_pitch(m->maxW) * (size_t) m->maxH;
            bool ok = hipSetDevice(d.device) == hipSuccess;
            for (int b = 0; b < 3 && ok; b++) ok = hipHostMalloc((void **) &d.hIn[b], bytes) == hipSuccess;
            if (!ok) {
                (void) hipGetLastError();
                for (int b = 0; b < 3; b++) { if (d.hIn[b]) (void) hipHostFree(d.hIn[b]); d.hIn[b] = nullptr; }
                d.err = "page-locking the input staging areas failed";
                d.rc = YGZF_ERR_HIP;
                return;
            }
        }
        auto prepare = [&](int k) {      // pageable frames: gather chunk k into its page-locked staging area
            if (pinned) return;
            const int b = k % 3;
            parallel(cnt(k), [&](int j) {
                const uint8_t *src = J.frames + (size_t) fr[lo(k) + j] * J.frame_stride;
                uint8_t *dst = d.hIn[b] + (size_t) j * h * sp;
                if (J.row_pitch == sp) memcpy(dst, src, (size_t) sp * (h - 1) + w);
                else for (int y = 0; y < h; y++) memcpy(dst + (size_t) y * sp, src + (size_t) y * J.row_pitch, (size_t) w);
            });
        };
        auto launch = [&](int k) {       // queue chunk k on its context: upload, extraction, matching / stereo (returns without waiting)
            const int b = k & 1;
            int rc;
            if (pinned) {
                for (int j = 0; j < cnt(k); j++) ptrs[j] = J.frames + (size_t) fr[lo(k) + j] * J.frame_stride;
                rc = ygzf_extract_batch_host_frames(cx(k), ptrs.data(), cnt(k), w, h, J.row_pitch);
            } else
                rc = ygzf_extract_batch_host(cx(k), d.hIn[k % 3], cnt(k), w, h, sp, (size_t) sp * h);
            if (rc == YGZF_OK && J.mode == kMatch) rc = ygzf_match_batch_prev(cx(k), J.cam, J.th, J.b_mono, J.check_level, J.check_orientation);
            if (rc == YGZF_OK && J.mode == kStereo) rc = ygzf_stereo_batch(cx(k), J.mb, J.mbf);
            if (rc != YGZF_OK) d.err = ygzf_last_error(cx(k));
            return rc;
        };
        auto fetch = [&](int k) {        // waits for chunk k and brings its results into the page-locked staging of its context
            const int b = k & 1;
            int rc;
            const size_t F = (size_t) m->chunk, oK = (F * sizeof(int) + 255) & ~(size_t) 255, oD = oK + ((F * m->stride * sizeof(ygzf_kp) + 255) & ~(size_t) 255);
            if ((size_t) cnt(k) * m->stride * 60 <= (2u << 20)) {
                // a few frames (one Tracking frame, one stereo pair): the three parts gathered on the device and brought over in ONE copy
                size_t pk = 0, pd = 0;
                rc = ygzf_batch_fetch_packed(cx(k), d.hRes[b], d.hResBytes, &pk, &pd, &d.resRow[b], nullptr);
                d.hCnt[b] = (int *) d.hRes[b]; d.hKp[b] = (ygzf_kp *) (d.hRes[b] + pk); d.hDesc[b] = d.hRes[b] + pd;
            } else {
                d.hCnt[b] = (int *) d.hRes[b]; d.hKp[b] = (ygzf_kp *) (d.hRes[b] + oK); d.hDesc[b] = d.hRes[b] + oD;
                rc = ygzf_batch_fetch_all(cx(k), d.hKp[b], d.hDesc[b], d.hCnt[b], m->stride);
                d.resRow[b] = m->stride;
            }
            if (rc == YGZF_OK && J.mode == kMatch) rc = ygzf_match_counts(cx(k), nm.data() + lo(k));
            if (rc == YGZF_OK && J.mode == kMatch && J.match) rc = ygzf_match_fetch_all(cx(k), d.hAux[b], m->stride);
            if (rc == YGZF_OK && J.mode == kStereo) {
                float *u = (float *) d.hAux[b];
                rc = ygzf_stereo_fetch_all(cx(k), u, u + (size_t) (cnt(k) / 2) * m->stride, m->stride);
            }
            if (rc != YGZF_OK) d.err = ygzf_last_error(cx(k));
            return rc;
        };
        auto scatter = [&](int k) {      // to the caller's rows: input order
            const int b = k & 1;
            parallel(cnt(k), [&](int j) {
                const int i = lo(k) + j, f = fr[i];
                const int c = d.hCnt[b][j];
                J.n_kp[f] = c;
                memcpy(J.kps + (size_t) f * J.stride, d.hKp[b] + (size_t) j * d.resRow[b], sizeof(ygzf_kp) * (size_t) c);
                memcpy(J.desc + (size_t) f * J.stride * 32, d.hDesc[b] + (size_t) j * d.resRow[b] * 32, 32 * (size_t) c);
                if (J.mode == kMatch) {
                    const bool first = (f % unit) == 0;                           // no predecessor inside the unit
                    if (J.nmatches) J.nmatches[f] = first ? -1 : nm[i];
                    if (J.match) {
                        int *row = J.match + (size_t) f * J.stride;
                        if (first) for (int q = 0; q < J.stride; q++) row[q] = -1;
                        else {
                            memcpy(row, d.hAux[b] + (size_t) j * m->stride, sizeof(int) * (size_t) c);
                            for (int q = c; q < J.stride; q++) row[q] = -1;
                        }
                    }
                } else if (J.mode == kStereo && (f & 1) == 0) {                   // left eye of pair f / 2: row j / 2 of this chunk
                    const float *u = (const float *) d.hAux[b] + (size_t) (j / 2) * m->stride;
                    const float *dp = u + (size_t) (cnt(k) / 2) * m->stride;
                    float *ur = J.u_right + (size_t) (f / 2) * J.stride, *dr = J.depth + (size_t) (f / 2) * J.stride;
                    memcpy(ur, u, sizeof(float) * (size_t) c);
                    memcpy(dr, dp, sizeof(float) * (size_t) c);
                    for (int q = c; q < J.stride; q++) { ur[q] = -1.f; dr[q] = -1.f; }
                }
            });
        };
        if (J.mode == kStereo && n == 2 && pinned) {
            // exactly one (left, right) pair for this slot (BASELINE.json's 3840x2160 configuration, a pair per GPU): the eyes go to the slot's two
            // contexts, the right eye's upload beside the left eye's kernels (ygzf_stereo_pair_host)
            size_t ok = 0, od = 0;
            int row = 0;
            float *u = (float *) d.hAux[0], *dp = u + m->stride;
            int rc1 = ygzf_stereo_pair_host(d.ctx[0], d.ctx[1], J.frames + (size_t) fr[0] * J.frame_stride, J.frames + (size_t) fr[1] * J.frame_stride, w, h, J.row_pitch,
                                            J.mb, J.mbf, d.hRes[0], d.hRes[1], d.hResBytes, &ok, &od, &row, u, dp);
            if (rc1 != YGZF_OK) { d.err = ygzf_last_error(d.ctx[0]); d.rc = rc1; return; }
            for (int e = 0; e < 2; e++) {
                const int f = fr[e], c = *(const int *) d.hRes[e];
                J.n_kp[f] = c;
                memcpy(J.kps + (size_t) f * J.stride, d.hRes[e] + ok, sizeof(ygzf_kp) * (size_t) c);
                memcpy(J.desc + (size_t) f * J.stride * 32, d.hRes[e] + od, 32 * (size_t) c);
            }
            const int cl = *(const int *) d.hRes[0], pr = fr[0] / 2;
            float *ur = J.u_right + (size_t) pr * J.stride, *dr = J.depth + (size_t) pr * J.stride;
            memcpy(ur, u, sizeof(float) * (size_t) cl);
            memcpy(dr, dp, sizeof(float) * (size_t) cl);
            for (int q = cl; q < J.stride; q++) { ur[q] = -1.f; dr[q] = -1.f; }
            d.rc = YGZF_OK;
            return;
        }
        prepare(0);
        int rc = launch(0);
        if (alternate) {
            // Two contexts: chunk k + 1 is queued (upload, kernels) before chunk k is waited for, and the moment chunk k's results are in the staging
            // area its context takes chunk k + 2 -- BEFORE the host copies those results out: the link, which bounds the whole call, then never waits
            // for a host-side copy (scattering a chunk of 128 frames is 8 MB of memcpy, gathering a pageable one 46 MB: with either between two
            // uploads the entry point reached 127 k frames/s where the bare two-context pipeline does 153 k).
            if (nChunks > 1 && rc == YGZF_OK) { prepare(1); rc = launch(1); }
            for (int k = 0; k < nChunks && rc == YGZF_OK; k++) {
                if (k + 2 < nChunks) prepare(k + 2);             // into the third staging area, while the device works on chunks k and k + 1
                rc = fetch(k);
                if (rc == YGZF_OK && k + 2 < nChunks) rc = launch(k + 2);
                if (rc == YGZF_OK) scatter(k);
            }
        } else {
            for (int k = 0; k < nChunks && rc == YGZF_OK; k++) {
                const bool more = k + 1 < nChunks;
                if (more) prepare(k + 1);                        // the device works on chunk k meanwhile
                if (rc == YGZF_OK) rc = fetch(k);
                if (rc == YGZF_OK && more) rc = launch(k + 1);   // one context: its outputs had to be read first
                if (rc == YGZF_OK) scatter(k);                   // the device works on chunk k + 1 meanwhile
            }
        }
        if (rc != YGZF_OK) {                                     // leave no copy in flight behind the caller's frames
            (void) ygzf_sync(d.ctx[0]);
            (void) ygzf_sync(d.ctx[1]);
        }
        d.rc = rc;
    };
    if (nd == 1) work(0, false);
    else {
        std::vector<std::thread> th_;                            // one host thread per device slot (bound to the device's NUMA node)
        for (int s = 0; s < nd; s++) th_.emplace_back(work, s, true);
        for (auto &t : th_) t.join();
    }
    for (int s = 0; s < nd; s++)
        if (m->devs[s].rc != YGZF_OK) return mfail(m, m->devs[s].rc, "device slot %d (device %d): %s", s, m->devs[s].device, m->devs[s].err.c_str());
    return YGZF_OK;
}
}  // namespace

extern "C" {

int ygzf_mgpu_extract_match(ygzf_mgpu *m, const uint8_t *frames, int n_frames, int w, int h, int row_pitch, size_t frame_stride, int unit,
                            const ygzf_camera *cam, float th, int b_mono, int check_level, int check_orientation, ygzf_kp *kps, uint8_t *desc,
                            int *n_kp, int stride, int *match, int *nmatches) {
    if (!m || !frames || !kps || !desc || !n_kp) return mfail(m, YGZF_ERR_INVALID, "null argument");
    if (n_frames < 1 || unit < 1 || w < 1 || h < 1 || w > m->maxW || h > m->maxH || row_pitch < w || stride < m->stride)
        return mfail(m, YGZF_ERR_INVALID, "bad geometry (frames %d, unit %d, %dx%d, stride %d < %d)", n_frames, unit, w, h, stride, m->stride);
    if ((match || nmatches) && !cam) return mfail(m, YGZF_ERR_INVALID, "matching needs a camera");
    Job J;
    memset(&J, 0, sizeof J);
    J.mode = (match || nmatches) ? kMatch : kExtractOnly;
    J.frames = frames; J.n_frames = n_frames; J.w = w; J.h = h; J.row_pitch = row_pitch; J.frame_stride = frame_stride; J.unit = unit;
    J.cam = cam; J.th = th; J.b_mono = b_mono; J.check_level = check_level; J.check_orientation = check_orientation;
    J.kps = kps; J.desc = desc; J.n_kp = n_kp; J.stride = stride; J.match = match; J.nmatches = nmatches;
    return run_job(m, J);
}

int ygzf_mgpu_extract_stereo(ygzf_mgpu *m, const uint8_t *frames, int n_frames, int w, int h, int row_pitch, size_t frame_stride, float mb, float mbf,
                             ygzf_kp *kps, uint8_t *desc, int *n_kp, int stride, float *u_right, float *depth) {
    if (!m || !frames || !kps || !desc || !n_kp || !u_right || !depth) return mfail(m, YGZF_ERR_INVALID, "null argument");
    if (n_frames < 2 || (n_frames & 1) || w < 1 || h < 1 || w > m->maxW || h > m->maxH || row_pitch < w || stride < m->stride)
        return mfail(m, YGZF_ERR_INVALID, "bad geometry (frames %d: (left, right) pairs, %dx%d, stride %d < %d)", n_frames, w, h, stride, m->stride);
    if (!(mb > 0)) return mfail(m, YGZF_ERR_INVALID, "baseline mb must be positive");
    Job J;
    memset(&J, 0, sizeof J);
    J.mode = kStereo;
    J.frames = frames; J.n_frames = n_frames; J.w = w; J.h = h; J.row_pitch = row_pitch; J.frame_stride = frame_stride; J.unit = 2;
    J.mb = mb; J.mbf = mbf;
    J.kps = kps; J.desc = desc; J.n_kp = n_kp; J.stride = stride; J.u_right = u_right; J.depth = depth;
    return run_job(m, J);
}

}  // extern "C"
