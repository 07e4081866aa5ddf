// ygzf_pool.cc -- see ygzf_pool.h (product code, host side).
#include "ygzf_pool.h"

#include <atomic>
#include <cstdio>
#include <algorithm>
#include <cstring>
#include <map>
#include <mutex>
#include <vector>

#include "../../../include/ygzf.h"

namespace ygzf_host {
namespace {
struct Pool {
    std::mutex mu;
    std::map<int, std::vector<ygzf_ctx *>> free_;   // keyed by device: a context only ever serves the device it was created on
};
Pool &pool() {
    static Pool *p = new Pool();   // leaked on purpose (see header)
    return *p;
}
}  // namespace

namespace {
struct Failures {
    std::mutex mu;
    std::atomic<unsigned long> count{0};
    std::string last;
    failure_callback cb = nullptr;
    void *user = nullptr;
};
Failures &failures() {
    static Failures *f = new Failures();   // leaked on purpose, like the pool
    return *f;
}
}  // namespace

void report_failure(const char *who, const char *what) {
    Failures &F = failures();
    failure_callback cb;
    void *user;
    {
        std::lock_guard<std::mutex> lk(F.mu);
        F.count.fetch_add(1);
        F.last = std::string(who ? who : "libygzf") + ": " + (what ? what : "unknown error");
        cb = F.cb;
        user = F.user;
    }
    fprintf(stderr, "%s: %s\n", who ? who : "libygzf", what ? what : "unknown error");
    if (cb) cb(who, what, user);
}
unsigned long failure_count() { return failures().count.load(); }
std::string last_failure() {
    std::lock_guard<std::mutex> lk(failures().mu);
    return failures().last;
}
void set_failure_callback(failure_callback cb, void *user) {
    std::lock_guard<std::mutex> lk(failures().mu);
    failures().cb = cb;
    failures().user = user;
}

Lease::Lease(int device) : c_(nullptr), device_(device) {
    {
        std::lock_guard<std::mutex> lk(pool().mu);
        std::vector<ygzf_ctx *> &v = pool().free_[device];
        if (!v.empty()) { c_ = v.back(); v.pop_back(); return; }
    }
    ygzf_extractor_cfg cfg = {1000, 1.2f, 8, 20, 7, 0};   // matcher / aligner entry points only use the context's stream and scratch buffers
    if (ygzf_create(device, &cfg, 64, 64, 1, &c_) != YGZF_OK) {
        char who[64];
        snprintf(who, sizeof who, "libygzf context pool (device %d)", device);
        report_failure(who, ygzf_last_error(nullptr));
        c_ = nullptr;
    }
}

Lease::~Lease() {
    if (!c_) return;
    std::lock_guard<std::mutex> lk(pool().mu);
    pool().free_[device_].push_back(c_);
}

struct ImageCache::Impl {
    std::mutex mu;
    int w = 0, h = 0, nlevels = 0, device = -1;
    float scaleFactor = 0;
    static const int kSlots = 96;
    // (kind, id) alone is recycled -- Tracking::Reset restarts Frame::nNextId / KeyFrame::nNextId at 0 while KeyFrames 0..5 may still sit in the
    // cache (src/Tracking.cc:1926-1927) -- and so is the buffer address; the third component is a fingerprint of the image CONTENT (a sparse
    // 64-bit hash over ~32 KB of level 0): a Frame copy, whose pyramid is a deep clone at another address (src/Frame.cc:185-187), hits the slot
    // its original filled; the same id with other pixels never does.
    struct Key {
        int kind; unsigned long id; unsigned long long print;
        bool operator<(const Key &o) const { return kind != o.kind ? kind < o.kind : id != o.id ? id < o.id : print < o.print; }
    };
    std::map<Key, int> slotOf;
    std::vector<Key> keyOf;
    std::vector<char> used;
    std::vector<unsigned long> lastUse;
    unsigned long tick = 0;
};

int ImageCache::capacity() { return Impl::kSlots; }

ImageCache &ImageCache::instance() {
    static ImageCache *c = [] { ImageCache *p = new ImageCache(); p->impl_ = new Impl(); return p; }();   // never destroyed (HIP may be gone at exit)
    return *c;
}

ImageCache::Guard::Guard(ImageCache &cache) : c(cache) { c.impl_->mu.lock(); }
ImageCache::Guard::~Guard() { c.impl_->mu.unlock(); }

bool ImageCache::prepare(int device, int w, int h, int nlevels, float scale_factor, const char *who) {
    Impl &I = *impl_;
    if (ctx_ && device == I.device && w == I.w && h == I.h && nlevels == I.nlevels && scale_factor == I.scaleFactor) return true;
    if (ctx_) ygzf_destroy(ctx_);
    ctx_ = nullptr;
    ygzf_extractor_cfg cfg = {1000, scale_factor, nlevels, 20, 7, 0};   // the pyramid geometry is all that matters here
    if (ygzf_create(device, &cfg, w, h, 1, &ctx_) != YGZF_OK) {
        ygzf_host::report_failure(who, ygzf_last_error(nullptr));
        ctx_ = nullptr;
        return false;
    }
    if (ygzf_image_cache_reserve(ctx_, Impl::kSlots, w, h) != YGZF_OK) {
        ygzf_host::report_failure(who, ygzf_last_error(ctx_));
        ygzf_destroy(ctx_);
        ctx_ = nullptr;
        return false;
    }
    I.device = device; I.w = w; I.h = h; I.nlevels = nlevels; I.scaleFactor = scale_factor;
    I.slotOf.clear();
    I.keyOf.assign(Impl::kSlots, Impl::Key{0, 0, 0});
    I.used.assign(Impl::kSlots, 0);
    I.lastUse.assign(Impl::kSlots, 0);
    return true;
}

void ImageCache::clear() {
    Impl &I = *impl_;
    I.slotOf.clear();
    std::fill(I.used.begin(), I.used.end(), 0);
    std::fill(I.lastUse.begin(), I.lastUse.end(), 0ul);
}

// 64 rows x 64 eight-byte words spread over the image, mixed FNV-1a style with the size: two different camera images agree on all of them with
// negligible probability, and reading 32 KB costs a few microseconds against the 360 KB upload a wrong miss would cost
unsigned long long image_fingerprint(const unsigned char *data, int cols, int rows, int step) {
    unsigned long long hsh = 1469598103934665603ull ^ ((unsigned long long) cols << 32 | (unsigned) rows);
    const int ny = rows < 64 ? rows : 64, nx = cols / 8 < 64 ? cols / 8 : 64;
    for (int j = 0; j < ny; j++) {
        const unsigned char *row = data + (size_t) ((long long) j * (rows - 1) / (ny > 1 ? ny - 1 : 1)) * step;
        for (int i = 0; i < nx; i++) {
            unsigned long long v;
            memcpy(&v, row + (size_t) ((long long) i * (cols - 8) / (nx > 1 ? nx - 1 : 1)), 8);
            hsh = (hsh ^ v) * 1099511628211ull;
        }
    }
    return hsh;
}

int ImageCache::slot(Kind kind, unsigned long id, const unsigned char *data, int cols, int rows, int step, const char *who, ygzf_ctx *resident) {
    Impl &I = *impl_;
    if (!ctx_ || cols != I.w || rows != I.h || cols < 8) return -1;
    const Impl::Key key{(int) kind, id, image_fingerprint(data, cols, rows, step)};
    auto it = I.slotOf.find(key);
    if (it != I.slotOf.end()) { I.lastUse[it->second] = ++I.tick; return it->second; }
    int victim = 0;
    for (int s = 1; s < Impl::kSlots; s++)
        if (I.lastUse[s] < I.lastUse[victim]) victim = s;
    if (I.used[victim]) I.slotOf.erase(I.keyOf[victim]);
    I.used[victim] = 0;
    // the image is on the device already when the Frame's extractor still holds it: device-to-device, else upload + pyramid
    if (!(resident && ygzf_image_cache_put_resident(ctx_, victim, resident) == YGZF_OK) &&
        ygzf_image_cache_put(ctx_, victim, data, cols, rows, step) != YGZF_OK) {
        ygzf_host::report_failure(who, ygzf_last_error(ctx_));
        return -1;
    }
    I.used[victim] = 1;
    I.keyOf[victim] = key;
    I.slotOf[key] = victim;
    I.lastUse[victim] = ++I.tick;
    return victim;
}
}  // namespace ygzf_host
