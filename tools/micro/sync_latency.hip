// tools/micro/sync_latency.hip -- what a host thread pays to learn that a short stream of work has finished: hipStreamSynchronize (the runtime's own
// wait) against polling (hipStreamQuery / hipEventQuery in a loop), for a ~20 us kernel followed by a small device-to-host copy -- the shape of every
// one-frame entry point of libygzf.  Prints the median wall time of submit + wait per variant.
//   hipcc --offload-arch=gfx950 -O2 tools/micro/sync_latency.hip -o tools/micro/bin/sync_latency
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <immintrin.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
__global__ void spin(float *out, int iters) {
    float a = threadIdx.x * 1e-3f;
    for (int i = 0; i < iters; i++) a = a * 1.0001f + 0.5f;
    out[threadIdx.x] = a;
}
int main() {
    float *d, *h;
    CK(hipMalloc((void **) &d, 65536));
    CK(hipHostMalloc((void **) &h, 65536));
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    hipEvent_t ev, evb;
    CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    CK(hipEventCreateWithFlags(&evb, hipEventDisableTiming | hipEventBlockingSync));
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto run = [&](const char *name, int iters, auto wait) {
        std::vector<double> us;
        for (int r = 0; r < 300; r++) {
            const auto t0 = now();
            hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s, d, iters);
            CK(hipMemcpyAsync(h, d, 60000, hipMemcpyDeviceToHost, s));
            wait();
            us.push_back(std::chrono::duration<double, std::micro>(now() - t0).count());
        }
        std::sort(us.begin(), us.end());
        printf("%-44s kernel iters %6d: median %7.1f us  p10 %7.1f  p90 %7.1f\n", name, iters, us[150], us[30], us[270]);
    };
    for (int iters : {100, 4000, 20000}) {
        run("hipStreamSynchronize", iters, [&] { CK(hipStreamSynchronize(s)); });
        run("poll hipStreamQuery", iters, [&] { while (hipStreamQuery(s) == hipErrorNotReady) _mm_pause(); });
        run("event + poll hipEventQuery", iters, [&] { CK(hipEventRecord(ev, s)); while (hipEventQuery(ev) == hipErrorNotReady) _mm_pause(); });
        run("event + hipEventSynchronize", iters, [&] { CK(hipEventRecord(ev, s)); CK(hipEventSynchronize(ev)); });
        run("blocking event + hipEventSynchronize", iters, [&] { CK(hipEventRecord(evb, s)); CK(hipEventSynchronize(evb)); });
        // a flag the device writes into page-locked memory, polled by the host (no runtime call in the wait)
        run("copy of a flag word, host polls memory", iters, [&] {
            volatile unsigned *f = (volatile unsigned *) (h + 15000);
            *f = 0xFFFFFFFFu;
            static unsigned *dflag = nullptr;
            if (!dflag) { CK(hipMalloc((void **) &dflag, 4)); CK(hipMemset(dflag, 0, 4)); }
            CK(hipMemcpyAsync((void *) f, dflag, 4, hipMemcpyDeviceToHost, s));
            while (*f == 0xFFFFFFFFu) _mm_pause();
        });
    }
    return 0;
}
