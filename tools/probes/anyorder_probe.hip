// Does hipExtAnyOrderLaunch let two independent kernels of ONE stream overlap on gfx950?  (hip_ext.h notes the flag as
// unsupported on GFX9xx for the module-launch API.)   hipcc --offload-arch=gfx950 -O2 anyorder_probe.hip -o anyorder_probe
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>

__global__ void spin(long long ticks, int* sink) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) {}
    if (sink != nullptr && threadIdx.x == 0 && blockIdx.x == 0) atomicAdd(sink, 1);
}

static float run(bool any_order, int n_pairs, hipStream_t s, int* sink) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipEventRecord(e0, s);
    for (int i = 0; i < n_pairs; ++i) {
        hipLaunchKernelGGL(spin, dim3(8), dim3(64), 0, s, 5000LL, sink);                       // 50 us at 100 MHz
        hipExtLaunchKernelGGL(spin, dim3(8), dim3(64), 0, s, nullptr, nullptr, any_order ? hipExtAnyOrderLaunch : 0, 5000LL, sink);
    }
    hipEventRecord(e1, s);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main() {
    hipStream_t s;
    hipStreamCreate(&s);
    int* sink;
    hipMalloc(&sink, 4);
    hipMemset(sink, 0, 4);
    run(false, 2, s, sink);
    const float in_order = run(false, 20, s, sink), any = run(true, 20, s, sink);
    printf("20 pairs of 50 us kernels on one stream: in order %.3f ms, second of each pair any-order %.3f ms\n", in_order, any);
    return 0;
}
