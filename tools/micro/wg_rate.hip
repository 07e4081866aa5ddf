// Microbenchmark: how fast does the MI355X start workgroups?  (grid = 300 x 256 workgroups of 256 threads, as k_fast_quads at 256 frames)
#include <hip/hip_runtime.h>
#include <cstdio>
struct Geom { int v[32]; };
__global__ __launch_bounds__(256) void k_empty(unsigned short *out) {
    if (threadIdx.x == 0 && blockIdx.x == 0x7fffffff) out[0] = 1;
}
__global__ __launch_bounds__(256) void k_lds(unsigned short *out) {
    extern __shared__ unsigned char dyn[];
    if (threadIdx.x == 0 && blockIdx.x == 0x7fffffff) out[0] = dyn[blockIdx.y];
}
__global__ __launch_bounds__(256) void k_store(unsigned short *out) {
    if ((threadIdx.x & 63) == 0) out[(blockIdx.y * gridDim.x + blockIdx.x) * 4 + (threadIdx.x >> 6)] = 1;
}
__global__ __launch_bounds__(256) void k_chain(unsigned short *out, const Geom *geom, int n) {
    int l = 0;
    const int grp = blockIdx.x;
    while (l + 1 < n && grp >= geom[l + 1].v[0]) l++;
    const Geom g = geom[l];
    int s = 0;
    for (int i = 0; i < 32; i++) s += g.v[i];
    if ((threadIdx.x & 63) == 0) out[(blockIdx.y * gridDim.x + blockIdx.x) * 4 + (threadIdx.x >> 6)] = (unsigned short) s;
}
template <class F> float timeit(F f) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    f(); hipDeviceSynchronize();
    hipEventRecord(a); for (int i = 0; i < 10; i++) f(); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms * 100.f;   // us per launch
}
int main() {
    unsigned short *out; hipMalloc(&out, 300 * 256 * 4 * 2 + 64);
    Geom h[8]; for (int i = 0; i < 8; i++) { for (int j = 0; j < 32; j++) h[i].v[j] = j; h[i].v[0] = i == 0 ? 0 : 100 + 30 * i; }
    Geom *dg; hipMalloc(&dg, sizeof h); hipMemcpy(dg, h, sizeof h, hipMemcpyHostToDevice);
    dim3 grid(300, 256);
    printf("empty           %8.1f us\n", timeit([&] { hipLaunchKernelGGL(k_empty, grid, dim3(256), 0, 0, out); }));
    printf("empty 64thr x4  %8.1f us\n", timeit([&] { hipLaunchKernelGGL(k_empty, dim3(1200, 256), dim3(64), 0, 0, out); }));
    for (int kb : {1, 8, 18, 40}) printf("lds %2d KB       %8.1f us\n", kb, timeit([&] { hipLaunchKernelGGL(k_lds, grid, dim3(256), kb * 1024, 0, out); }));
    printf("store/wave      %8.1f us\n", timeit([&] { hipLaunchKernelGGL(k_store, grid, dim3(256), 0, 0, out); }));
    printf("chain+store     %8.1f us\n", timeit([&] { hipLaunchKernelGGL(k_chain, grid, dim3(256), 0, 0, out, dg, 8); }));
    printf("empty 1-D grid  %8.1f us\n", timeit([&] { hipLaunchKernelGGL(k_empty, dim3(76800), dim3(256), 0, 0, out); }));
    printf("empty 1024thr   %8.1f us\n", timeit([&] { hipLaunchKernelGGL(k_empty, dim3(19200), dim3(1024), 0, 0, out); }));
    return 0;
}
