// glds_probe.hip -- what does gfx950's LDS-DMA (global_load_lds_*) accept?  (measurement tool, not product code)
//   a) dword pieces from byte-UNALIGNED global addresses     b) dwordx4 pieces from 4-byte aligned (not 16-byte aligned) addresses
//   c) dwordx4 from byte-unaligned addresses                 d) partially masked pieces (inactive lanes must not write)
// Every case fills 4 KB of LDS per wave from a known byte ramp and compares with what plain byte loads give.
// build: hipcc --offload-arch=gfx950 -O2 tools/micro/glds_probe.hip -o tools/micro/bin/glds_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __attribute__((address_space(3))) void lds_void;

template <int SZ>
__global__ void k_probe(const unsigned char *g, int byteOff, int rowPitch, int lanesActive, unsigned *bad, unsigned char *dump) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[4096];
    const int lane = threadIdx.x;
    for (int i = lane; i < 1024; i += 64) ((unsigned *) lds)[i] = 0xEEEEEEEEu;
    __syncthreads();
    // piece u: lane -> (row = (u*64+lane) / 3, chunk = % 3): a 3-chunk-wide window row like the FAST window, rows rowPitch apart
#pragma unroll
    for (int u = 0; u < 2; u++) {
        const int i = u * 64 + lane, r = i / 3, c = i - 3 * r;
        const unsigned char *src = g + byteOff + r * rowPitch + c * SZ;
        if (lane < lanesActive) {
#if defined(__HIP_DEVICE_COMPILE__)
            if constexpr (SZ == 4) __builtin_amdgcn_global_load_lds(src, (lds_void *) (lds + u * 64 * 4), 4, 0, 0);
            else __builtin_amdgcn_global_load_lds(src, (lds_void *) (lds + u * 64 * 16), 16, 0, 0);
#endif
        }
    }
    __builtin_amdgcn_s_waitcnt(0x0f70);   // vmcnt(0)
    __syncthreads();
    unsigned nbad = 0;
    for (int u = 0; u < 2; u++) {
        const int i = u * 64 + lane, r = i / 3, c = i - 3 * r;
        for (int b = 0; b < SZ; b++) {
            const unsigned char want = lane < lanesActive ? g[byteOff + r * rowPitch + c * SZ + b] : 0xEE;
            const unsigned char got = lds[(u * 64 + lane) * SZ + b];
            if (want != got) nbad++;
            dump[(u * 64 + lane) * SZ + b] = got;
        }
    }
    atomicAdd(bad, nbad);
}

int main() {
    const int N = 1 << 16;
    std::vector<unsigned char> h(N);
    for (int i = 0; i < N; i++) h[i] = (unsigned char) ((i * 7 + (i >> 8) * 13) & 0xFF);
    unsigned char *d, *dump;
    unsigned *bad;
    hipMalloc(&d, N); hipMalloc(&dump, 4096); hipMalloc(&bad, 4);
    hipMemcpy(d, h.data(), N, hipMemcpyHostToDevice);
    struct Case { int sz, off, pitch, lanes; const char *what; } cases[] = {
        {4, 0, 752, 64, "dword, aligned"}, {4, 1, 752, 64, "dword, byte offset 1"}, {4, 3, 753, 64, "dword, byte offset 3, odd pitch"},
        {16, 0, 752, 64, "dwordx4, 16-byte aligned"}, {16, 4, 752, 64, "dwordx4, 4-byte aligned"}, {16, 8, 628, 64, "dwordx4, 4-byte aligned rows"},
        {16, 5, 752, 64, "dwordx4, byte offset 5"}, {16, 13, 627, 64, "dwordx4, byte offset 13, odd pitch"},
        {4, 2, 752, 40, "dword, 40 lanes active"}, {16, 4, 752, 23, "dwordx4, 23 lanes active"},
    };
    for (const Case &c : cases) {
        hipMemset(bad, 0, 4);
        if (c.sz == 4) hipLaunchKernelGGL(k_probe<4>, dim3(1), dim3(64), 0, 0, d, c.off, c.pitch, c.lanes, bad, dump);
        else hipLaunchKernelGGL(k_probe<16>, dim3(1), dim3(64), 0, 0, d, c.off, c.pitch, c.lanes, bad, dump);
        unsigned nb = 0;
        hipError_t e = hipDeviceSynchronize();
        hipMemcpy(&nb, bad, 4, hipMemcpyDeviceToHost);
        printf("%-40s : %s, %u bad bytes\n", c.what, e == hipSuccess ? "ran" : hipGetErrorString(e), nb);
        if (e != hipSuccess) return 1;
    }
    return 0;
}
