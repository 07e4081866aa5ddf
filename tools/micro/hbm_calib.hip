// tools/micro/hbm_calib.hip -- known-size streams in the access widths libygzf's kernels use, to calibrate rocprofv3's FETCH_SIZE / WRITE_SIZE on
// gfx950 (MI355X_MICROARCH.md: FETCH_SIZE reports half the bytes of a wide coalesced read on this chip; other widths must be calibrated).
// Every kernel reads N bytes once and / or writes N bytes once (N = 1 GiB, four times the 256 MiB Infinity Cache); run under
//   rocprofv3 --pmc FETCH_SIZE --kernel-trace ...   and   rocprofv3 --pmc WRITE_SIZE --kernel-trace ...   (tools/hbm_calib.sh)
// factor = N / (counter * 1024).  Measurement tool, not product code.
#include <hip/hip_runtime.h>
#include <cstdio>

typedef __attribute__((address_space(3))) void lds_void_t;
constexpr size_t N = 1ull << 30;

__global__ __launch_bounds__(256) void calib_read_dword(const unsigned *__restrict__ src, unsigned *__restrict__ sink) {
    const size_t i = (size_t) blockIdx.x * 256 + threadIdx.x, n = N / 4, stride = (size_t) gridDim.x * 256;
    unsigned acc = 0;
    for (size_t k = i; k < n; k += stride) acc ^= src[k];
    if (acc == 0x12345678u) sink[0] = acc;
}
__global__ __launch_bounds__(256) void calib_read_dwordx4(const uint4 *__restrict__ src, unsigned *__restrict__ sink) {
    const size_t i = (size_t) blockIdx.x * 256 + threadIdx.x, n = N / 16, stride = (size_t) gridDim.x * 256;
    unsigned acc = 0;
    for (size_t k = i; k < n; k += stride) { const uint4 v = src[k]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345678u) sink[0] = acc;
}
__global__ __launch_bounds__(256) void calib_read_ldsdma16(const uint4 *__restrict__ src, unsigned *__restrict__ sink) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[4 * 1024];
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const size_t n = N / 16, stride = (size_t) gridDim.x * 256;
    for (size_t k = (size_t) blockIdx.x * 256 + wv * 64; k < n; k += stride) {
        __builtin_amdgcn_global_load_lds((const unsigned *) (src + k + lane), (lds_void_t *) (lds + wv * 1024), 16, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (((unsigned *) lds)[threadIdx.x] == 0x12345678u) sink[0] = 1;
}
// the FAST window pattern: rows of 48 bytes at byte-unaligned starts, 752 bytes apart, 16-byte pieces (three lanes per row)
__global__ __launch_bounds__(256) void calib_read_ldsdma16_rows48(const unsigned char *__restrict__ src, unsigned *__restrict__ sink) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[4 * 1024];
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const size_t rows = N / 752, stride = (size_t) gridDim.x * 4 * 21;
    for (size_t r0 = ((size_t) blockIdx.x * 4 + wv) * 21; r0 + 21 < rows; r0 += stride) {   // 21 rows x 3 pieces = 63 lanes
        const int r = lane / 3, c = lane - 3 * r;
        if (lane < 63)
            __builtin_amdgcn_global_load_lds((const unsigned *) (src + (r0 + r) * 752 + 15 + 30 * (r0 % 23) + 16 * c), (lds_void_t *) (lds + wv * 1024), 16, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (((unsigned *) lds)[threadIdx.x] == 0x12345678u) sink[0] = 1;
}
__global__ __launch_bounds__(256) void calib_write_dword(unsigned *__restrict__ dst) {
    const size_t i = (size_t) blockIdx.x * 256 + threadIdx.x, n = N / 4, stride = (size_t) gridDim.x * 256;
    for (size_t k = i; k < n; k += stride) dst[k] = (unsigned) k;
}
__global__ __launch_bounds__(256) void calib_write_dwordx4(uint4 *__restrict__ dst) {
    const size_t i = (size_t) blockIdx.x * 256 + threadIdx.x, n = N / 16, stride = (size_t) gridDim.x * 256;
    for (size_t k = i; k < n; k += stride) dst[k] = make_uint4((unsigned) k, 1, 2, 3);
}
__global__ __launch_bounds__(256) void calib_write_byte(unsigned char *__restrict__ dst) {
    const size_t i = (size_t) blockIdx.x * 256 + threadIdx.x, n = N / 4, stride = (size_t) gridDim.x * 256;   // N / 4 bytes written
    for (size_t k = i; k < n; k += stride) dst[k] = (unsigned char) k;
}

int main() {
    unsigned char *a, *b;
    unsigned *sink;
    if (hipMalloc(&a, N + 4096) != hipSuccess || hipMalloc(&b, N + 4096) != hipSuccess || hipMalloc(&sink, 64) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(a, 1, N + 4096);
    hipMemset(b, 2, N + 4096);
    const dim3 grid(256 * 8), block(256);
    for (int rep = 0; rep < 3; rep++) {
        hipLaunchKernelGGL(calib_read_dword, grid, block, 0, 0, (const unsigned *) a, sink);
        hipLaunchKernelGGL(calib_read_dwordx4, grid, block, 0, 0, (const uint4 *) a, sink);
        hipLaunchKernelGGL(calib_read_ldsdma16, grid, block, 0, 0, (const uint4 *) a, sink);
        hipLaunchKernelGGL(calib_read_ldsdma16_rows48, grid, block, 0, 0, (const unsigned char *) a, sink);
        hipLaunchKernelGGL(calib_write_dword, grid, block, 0, 0, (unsigned *) b);
        hipLaunchKernelGGL(calib_write_dwordx4, grid, block, 0, 0, (uint4 *) b);
        hipLaunchKernelGGL(calib_write_byte, grid, block, 0, 0, b);
        hipDeviceSynchronize();
    }
    const size_t rows = N / 752;
    printf("bytes: read_dword %zu read_dwordx4 %zu read_ldsdma16 %zu read_ldsdma16_rows48 %zu (requested; the rows cover %zu image bytes) write_dword %zu write_dwordx4 %zu write_byte %zu\n",
           N, N, N, (rows / 21) * 21 * 48, (rows / 21) * 21 * 752, N, N, N / 4);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}
