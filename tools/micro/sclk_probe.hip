// tools/micro/sclk_probe.hip -- what shader clock do short, sparse launches run at?  One wave spins for a fixed number of dependent integer
// operations and reports shader-clock cycles (s_memtime) against the constant 100 MHz counter (s_memrealtime): cycles per 10 ns = clock / 100 MHz.
// Launched (a) once per 100 us with idle gaps, like a tracker that extracts one frame at a time, (b) back to back.
// build: hipcc --offload-arch=gfx950 -O2 tools/micro/sclk_probe.hip -o /tmp/sclk_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <thread>

__global__ void spin(long long *out, int n) {
    const long long c0 = clock64(), w0 = wall_clock64();
    unsigned v = threadIdx.x;
    for (int i = 0; i < n; i++) v = v * 1664525u + 1013904223u;
    const long long c1 = clock64(), w1 = wall_clock64();
    if (threadIdx.x == 0) { out[0] = c1 - c0; out[1] = w1 - w0; out[2] = v; }
}

int main() {
    long long *d, h[3];
    hipMalloc(&d, 64);
    for (int mode = 0; mode < 2; mode++) {
        double mhz = 0, us = 0;
        const int reps = 200;
        for (int r = 0; r < reps; r++) {
            hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, 0, d, 20000);
            hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
            mhz += 100.0 * (double) h[0] / (double) h[1];
            us += (double) h[1] / 100.0;
            if (mode == 0) std::this_thread::sleep_for(std::chrono::microseconds(100));
        }
        printf("%s: %.0f MHz effective shader clock, %.1f us per 20000-step spin\n", mode == 0 ? "sparse launches (100 us gaps)" : "back-to-back launches", mhz / reps, us / reps);
    }
    // under load: a long spin on all CUs first
    hipLaunchKernelGGL(spin, dim3(2048), dim3(256), 0, 0, d + 4, 2000000);
    hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, 0, d, 20000);
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    printf("right after a 2048-workgroup load: %.0f MHz, %.1f us\n", 100.0 * (double) h[0] / (double) h[1], (double) h[1] / 100.0);
    return 0;
}
