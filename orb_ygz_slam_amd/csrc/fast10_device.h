// fast10_device.h -- the FAST-10 segment test by its definition (shared by fast10_kernels.hip and dso_kernels.hip).
// Replaces the generated decision trees of Thirdparty/fast/src/fast_10.cpp and fast_10_score.cpp.
#ifndef YGZF_FAST10_DEVICE_H
#define YGZF_FAST10_DEVICE_H
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ygzf {

// max over the 16 arcs of 10 contiguous ring pixels of the min signed margin, both polarities
__device__ __forceinline__ int arc10_margin(const uint8_t *p, int pitch) {
    const int v = p[0];
    int d[16];
    d[0] = p[3 * pitch] - v;      d[1] = p[3 * pitch + 1] - v;  d[2] = p[2 * pitch + 2] - v;  d[3] = p[pitch + 3] - v;
    d[4] = p[3] - v;              d[5] = p[-pitch + 3] - v;     d[6] = p[-2 * pitch + 2] - v; d[7] = p[-3 * pitch + 1] - v;
    d[8] = p[-3 * pitch] - v;     d[9] = p[-3 * pitch - 1] - v; d[10] = p[-2 * pitch - 2] - v; d[11] = p[-pitch - 3] - v;
    d[12] = p[-3] - v;            d[13] = p[pitch - 3] - v;     d[14] = p[2 * pitch - 2] - v; d[15] = p[3 * pitch - 1] - v;
    int best = -256;
#pragma unroll
    for (int pol = 0; pol < 2; pol++) {
        int m2[16], m4[16], m8[16];
#pragma unroll
        for (int k = 0; k < 16; k++) m2[k] = min(d[k], d[(k + 1) & 15]);
#pragma unroll
        for (int k = 0; k < 16; k++) m4[k] = min(m2[k], m2[(k + 2) & 15]);
#pragma unroll
        for (int k = 0; k < 16; k++) m8[k] = min(m4[k], m4[(k + 4) & 15]);
#pragma unroll
        for (int k = 0; k < 16; k++) best = max(best, min(m8[k], m2[(k + 8) & 15]));   // 10 contiguous: 8 + 2
#pragma unroll
        for (int k = 0; k < 16; k++) d[k] = -d[k];
    }
    return best;   // corner at barrier b  <=>  best > b
}

}  // namespace ygzf
#endif
