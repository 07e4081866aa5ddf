// ygzf_pool.cc -- see ygzf_pool.h (product code, host side).
#include "ygzf_pool.h"

#include <cstdio>
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

Lease::Lease(int device) : c_(nullptr), device_(device) {
    {
        std::lock_guard<std::mutex> lk(pool().mu);
        std::vector<ygzf_ctx *> &v = pool().free_[device];
        if (!v.empty()) { c_ = v.back(); v.pop_back(); return; }
    }
    ygzf_extractor_cfg cfg = {1000, 1.2f, 8, 20, 7, 0};   // matcher / aligner entry points only use the context's stream and scratch buffers
    if (ygzf_create(device, &cfg, 64, 64, 1, &c_) != YGZF_OK) {
        fprintf(stderr, "libygzf context pool (device %d): %s\n", device, ygzf_last_error(nullptr));
        c_ = nullptr;
    }
}

Lease::~Lease() {
    if (!c_) return;
    std::lock_guard<std::mutex> lk(pool().mu);
    pool().free_[device_].push_back(c_);
}
}  // namespace ygzf_host
