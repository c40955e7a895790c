// queue_gap: what a kernel boundary costs on one HIP stream when another stream of the process is busy too (a measurement).
// The order loop with two column groups keeps two streams busy with chains of dependent kernels; the kernel trace shows ~5 us
// between a kernel's end and its successor's start on each of them, against ~0 when one stream runs alone.  This times chains of
// small kernels (each spins for `us` microseconds on `wgs` workgroups) on one stream, on two streams at once, and on two streams of
// different priorities:   time per kernel of a chain - us   = the boundary's cost.
//   hipcc -O3 --offload-arch=gfx950 -o tools/queue_gap tools/queue_gap.hip && ./tools/queue_gap
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>

#define CHK(x)                                                                             \
    do {                                                                                   \
        hipError_t e_ = (x);                                                               \
        if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } \
    } while (0)

__global__ void k_spin(long long ticks, int* sink) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
    if (ticks < 0) *sink = 1;
}

static double chain_ms(hipStream_t s, int n, int wgs, long long ticks, int* sink) {
    for (int i = 0; i < n; ++i) hipLaunchKernelGGL(k_spin, dim3(wgs), dim3(256), 0, s, ticks, sink);
    return 0;
}

int main() {
    int* sink;
    CHK(hipMalloc(&sink, 4));
    int rate_khz = 0;
    CHK(hipDeviceGetAttribute(&rate_khz, hipDeviceAttributeWallClockRate, 0));
    hipStream_t a, b, hi, lo;
    CHK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking));
    CHK(hipStreamCreateWithFlags(&b, hipStreamNonBlocking));
    int least, greatest;
    CHK(hipDeviceGetStreamPriorityRange(&least, &greatest));
    CHK(hipStreamCreateWithPriority(&hi, hipStreamNonBlocking, greatest));
    CHK(hipStreamCreateWithPriority(&lo, hipStreamNonBlocking, least));
    const int n = 400;
    printf("wall clock %d kHz; chains of %d kernels; time per kernel minus the kernel's own spin = cost of a boundary\n", rate_khz, n);
    for (int wgs : {8, 256}) {
        for (double us : {10.0, 40.0}) {
            const long long ticks = (long long)(us * rate_khz / 1000.0);
            auto run = [&](std::vector<hipStream_t> ss, const char* what) {
                for (auto s : ss) chain_ms(s, 20, wgs, ticks, sink);
                hipDeviceSynchronize();
                const auto t0 = std::chrono::steady_clock::now();
                // the host feeds the streams in turn, as the order loop does
                for (int i = 0; i < n; ++i)
                    for (auto s : ss) hipLaunchKernelGGL(k_spin, dim3(wgs), dim3(256), 0, s, ticks, sink);
                hipDeviceSynchronize();
                const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
                printf("  %3d workgroups x %4.0f us  %-42s %7.2f us per kernel of a chain  (boundary %+6.2f us)\n", wgs, us, what, ms * 1e3 / n, ms * 1e3 / n - us);
            };
            run({a}, "one stream");
            run({a, b}, "two streams");
            run({hi, lo}, "two streams, priorities high / low");
            run({a, b, hi}, "three streams");
        }
    }
    return 0;
}
