// ygzf_pool.h -- per-device pool of small libygzf contexts for the class shells that have no member to keep one in: ygz::ORBmatcher objects
// live on the stacks of the Tracking, LocalMapping and LoopClosing threads at the same time and the reference's SparseImgAlign
// class has no spare member, so both lease a context (own HIP stream + scratch buffers) for the duration of one call.
// Contexts are created on demand, handed back on release and deliberately never destroyed: at process exit the HIP runtime may already be
// gone when static destructors run.
#ifndef YGZF_HOST_POOL_H
#define YGZF_HOST_POOL_H

struct ygzf_ctx;

namespace ygzf_host {
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
}  // namespace ygzf_host
#endif
