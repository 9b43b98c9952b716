// Cost of N workgroups each taking one device-scope ticket on ONE address vs on 32 addresses (gfx950: the L2s of the 8
// XCDs are not coherent, device-scope atomics resolve at the memory side).  hipcc --offload-arch=gfx950 -O2 atomic_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ void tickets(unsigned* counters, int n_addr, unsigned* out) {
    if (threadIdx.x == 0) {
        unsigned t = __hip_atomic_fetch_add(&counters[(blockIdx.x % n_addr) * 32], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        out[blockIdx.x] = t;
    }
}
__global__ void nothing(unsigned* out) { if (threadIdx.x == 0) out[blockIdx.x] = 1; }

int main() {
    unsigned *c, *o;
    hipMalloc(&c, 32 * 32 * 4);
    hipMalloc(&o, 4096 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int blocks : {256, 1024}) {
        for (int mode = 0; mode < 3; ++mode) {
            hipMemset(c, 0, 32 * 32 * 4);
            for (int w = 0; w < 3; ++w) {
                if (mode == 0) hipLaunchKernelGGL(nothing, dim3(blocks), dim3(256), 0, 0, o);
                else hipLaunchKernelGGL(tickets, dim3(blocks), dim3(256), 0, 0, c, mode == 1 ? 1 : 32, o);
            }
            hipEventRecord(e0, 0);
            for (int r = 0; r < 50; ++r) {
                if (mode == 0) hipLaunchKernelGGL(nothing, dim3(blocks), dim3(256), 0, 0, o);
                else hipLaunchKernelGGL(tickets, dim3(blocks), dim3(256), 0, 0, c, mode == 1 ? 1 : 32, o);
            }
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("%4d workgroups, %s: %.2f us per launch\n", blocks,
                   mode == 0 ? "no atomic          " : (mode == 1 ? "tickets on 1 address" : "tickets on 32 addr  "), ms * 1000 / 50);
        }
    }
    return 0;
}
