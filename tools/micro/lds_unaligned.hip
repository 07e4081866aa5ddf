// tools/micro/lds_unaligned.hip -- what does a 16-byte LDS read at a 2-byte aligned address cost on gfx950 (the compiler emits ds_read_b128 for it: unaligned
// DS access is a target feature), against the seven ds_read_u16 it would replace in k_describe's column pass?  Every lane reads at its own pseudo-random
// offset inside a 3.4 KB per-wave region (the size of the row-blurred window); 8 waves per SIMD.
// Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 tools/micro/lds_unaligned.hip -o /tmp/lds_unaligned && /tmp/lds_unaligned
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u4 __attribute__((ext_vector_type(4)));
typedef u4 u4_a2 __attribute__((aligned(2)));
constexpr int kIter = 4000, kRegion = 1720;   // u16 elements per wave
template <int MODE>   // 0: seven u16 reads 40 elements apart (row-major hb), 1: one 16-byte read at a 2-byte aligned address, 2: the same at a 16-byte aligned address
__global__ __launch_bounds__(256) void k(unsigned *out, long long *clk, unsigned seed) {
    __shared__ __attribute__((aligned(16))) unsigned short l[4 * kRegion + 64];
    for (int i = threadIdx.x; i < 4 * kRegion + 64; i += 256) l[i] = (unsigned short) (i * 7 + seed);
    __syncthreads();
    const unsigned short *base = l + (threadIdx.x >> 6) * kRegion;
    unsigned acc = 0, o = (threadIdx.x * 2654435761u + seed) % 1400u;
    const long long t0 = clock64();
    for (int it = 0; it < kIter; it++) {
        if (MODE == 0) {
#pragma unroll
            for (int k2 = 0; k2 < 7; k2++) acc += base[o + 40 * k2];
        } else {
            const unsigned oo = MODE == 2 ? (o & ~7u) : o;
            const u4 v = *(const u4_a2 *) &base[oo];
            acc += v.x + v.y + v.z + (v.w & 0xFFFFu);
        }
        o = (o * 5u + 17u + (acc & 1u)) % 1400u;
    }
    const long long t1 = clock64();
    out[blockIdx.x * 256 + threadIdx.x] = acc;
    if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = t1 - t0;
}
int main() {
    unsigned *out; long long *clk;
    hipMalloc(&out, 256 * 8 * 256 * 4); hipMalloc(&clk, 8);
    const char *names[3] = {"7 x ds_read_u16 (column of a row-major u16 array)", "1 x ds_read_b128, 2-byte aligned", "1 x ds_read_b128, 16-byte aligned"};
    for (int m = 0; m < 3; m++) {
        for (int rep = 0; rep < 2; rep++) {
            if (m == 0) hipLaunchKernelGGL(k<0>, dim3(256 * 8), dim3(256), 0, 0, out, clk, 1u + rep);
            if (m == 1) hipLaunchKernelGGL(k<1>, dim3(256 * 8), dim3(256), 0, 0, out, clk, 1u + rep);
            if (m == 2) hipLaunchKernelGGL(k<2>, dim3(256 * 8), dim3(256), 0, 0, out, clk, 1u + rep);
            hipDeviceSynchronize();
        }
        long long c; hipMemcpy(&c, clk, 8, hipMemcpyDeviceToHost);
        printf("%-52s %7.1f shader cycles per iteration per wave (8 waves per SIMD share the CU's LDS)\n", names[m], (double) c / kIter);
    }
    return 0;
}
