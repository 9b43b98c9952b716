// How many kernel launches does the runtime accept on a stream before hipLaunchKernel blocks?  Launches N kernels that spin for
// ~D us each into one stream and records the host duration of every launch call; prints the first index whose call took
// > 20 us and the distribution behind it.   hipcc --offload-arch=gfx950 -O2 tools/probes/queue_depth_probe.hip -o queue_depth_probe
//   ./queue_depth_probe [n_launches] [spin_us] [arg_bytes]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

struct Big { char pad[4096]; };

__global__ void spin_kernel(long long ticks, int* sink) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) {}
    if (sink != nullptr && threadIdx.x == 1000) sink[0] = 1;
}
template <int BYTES>
struct Pad { char p[BYTES]; };
template <int BYTES>
__global__ void spin_kernel_args(long long ticks, int* sink, Pad<BYTES> pad) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) {}
    if (sink != nullptr && threadIdx.x == 1000) sink[0] = pad.p[0];
}

int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 4000;
    const double us = argc > 2 ? atof(argv[2]) : 50.0;
    const int big = argc > 3 ? atoi(argv[3]) : 0;
    hipStream_t s;
    hipStreamCreate(&s);
    int* sink;
    hipMalloc(&sink, 64);
    const long long ticks = (long long)(us * 100.0);      // wall_clock64: 100 MHz
    hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, s, 100LL, sink);
    hipStreamSynchronize(s);
    std::vector<double> d(n);
    Pad<2048> pad{};
    const auto t_begin = std::chrono::steady_clock::now();
    for (int i = 0; i < n; ++i) {
        const auto t0 = std::chrono::steady_clock::now();
        if (big) hipLaunchKernelGGL(spin_kernel_args<2048>, dim3(1), dim3(64), 0, s, ticks, sink, pad);
        else hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, s, ticks, sink);
        d[i] = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    }
    const double enq = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_begin).count();
    hipStreamSynchronize(s);
    const double tot = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_begin).count();
    int first = -1;
    for (int i = 0; i < n; ++i) if (d[i] > 20.0) { first = i; break; }
    int blocked = 0; double bsum = 0;
    for (int i = 0; i < n; ++i) if (d[i] > 20.0) { ++blocked; bsum += d[i]; }
    printf("%d launches of a %.0f us kernel (%s args): enqueue %.1f us total (%.2f us / launch), drained after %.1f us; first blocking call: index %d; "
           "%d calls > 20 us (avg %.1f us)\n", n, us, big ? "2 KB" : "16 B", enq, enq / n, tot, first, blocked, blocked ? bsum / blocked : 0.0);
    return 0;
}
