// ygzf_pool.h -- per-device pool of small libygzf contexts for the class shells that have no member to keep one in: ygz::ORBmatcher objects
// live on the stacks of the Tracking, LocalMapping and LoopClosing threads at the same time and the reference's SparseImgAlign
// class has no spare member, so both lease a context (own HIP stream + scratch buffers) for the duration of one call.
// Contexts are created on demand, handed back on release and deliberately never destroyed: at process exit the HIP runtime may already be
// gone when static destructors run.
#ifndef YGZF_HOST_POOL_H
#define YGZF_HOST_POOL_H

#include <string>

struct ygzf_ctx;

namespace ygzf_host {
// ---- failure channel of the class shells ------------------------------------------------------------------------------------------------
// The reference's signatures have no error channel (void / count returns), so a shell whose device call fails can only return
// "0 matches" / an empty result -- which Tracking cannot tell from a frame with nothing to match.  Every such return goes through
// report_failure: it counts, remembers the message, calls the registered callback and prints to stderr.  A SLAM system polls failure_count()
// once per frame (or installs a callback that raises its own "lost" flag); the tests provoke failures and read it.
typedef void (*failure_callback)(const char *who, const char *what, void *user);
void report_failure(const char *who, const char *what);
unsigned long failure_count();                 // process-wide, monotone
std::string last_failure();                    // "who: what" of the most recent one ("" when none)
void set_failure_callback(failure_callback cb, void *user);   // nullptr removes it; called on the failing thread

class Lease {
public:
    explicit Lease(int device);
    ~Lease();
    Lease(const Lease &) = delete;
    Lease &operator=(const Lease &) = delete;
    ygzf_ctx *get() const { return c_; }
    explicit operator bool() const { return c_ != nullptr; }

private:
    ygzf_ctx *c_;
    int device_;
};

// 64-bit content hash over ~32 KB of an 8-bit image (64 rows x 64 eight-byte words spread over it, mixed with the size)
unsigned long long image_fingerprint(const unsigned char *data, int cols, int rows, int step);

// Device-resident level-0 images + pyramids of recent Frames and KeyFrames, shared by the shells that read images (FindDirectProjection,
// SparseImgAlign::run): an image is uploaded the first time it is referenced and its pyramid rebuilt on the device by the extractor's
// resize kernel (so it equals the host pyramid the Frame holds); slots are recycled least recently used.  One cache per process, one
// context of its own; every use holds the mutex for the whole call (the cache's context is single-threaded like any other).
class ImageCache {
public:
    enum Kind { kFrame = 0, kKeyFrame = 1 };
    static ImageCache &instance();
    // locks the cache for the caller's scope
    struct Guard {
        explicit Guard(ImageCache &c);
        ~Guard();
        ImageCache &c;
    };
    // (re)creates the context when the geometry changes; false on failure (message on stderr)
    bool prepare(int device, int w, int h, int nlevels, float scale_factor, const char *who);
    // slot holding this image (uploads it on a miss); -1 on failure.  The key is (kind, id, fingerprint of the level-0 pixels): a Frame copy
    // (same id, deep-cloned pyramid at another address, src/Frame.cc:185-187) hits the slot its original filled; an id or a buffer address
    // that comes back with other pixels (Tracking::Reset restarts the id counters, src/Tracking.cc:1926-1927) misses.
    // `resident`: a context of the same device that may still hold this very image with its pyramid (the Frame's extractor, which built the
    // pyramid in the Frame's constructor): on a miss the slot is then filled device to device instead of by an upload and a second pyramid.
    int slot(Kind kind, unsigned long id, const unsigned char *data, int cols, int rows, int step, const char *who, ygzf_ctx *resident = nullptr);
    // forgets every cached image (the slots stay allocated).  Not needed for correctness -- the content fingerprint already keeps a recycled id
    // from hitting a stale image -- but the natural call in Tracking::Reset beside mpMap->clear().
    void clear();
    ygzf_ctx *ctx() const { return ctx_; }
    static int capacity();                 // slots: a batch call may reference at most capacity() distinct images

private:
    ImageCache() = default;
    struct Impl;
    Impl *impl_ = nullptr;
    ygzf_ctx *ctx_ = nullptr;
};
}  // namespace ygzf_host
#endif
