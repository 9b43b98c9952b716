// Host cost of the fork / join primitives between two HIP streams on this stack.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void tiny(int* p) { if (threadIdx.x == 0 && p) p[0] = 1; }
int main() {
    hipStream_t a, b;
    hipStreamCreateWithFlags(&a, hipStreamNonBlocking);
    hipStreamCreateWithFlags(&b, hipStreamNonBlocking);
    hipEvent_t e;
    hipEventCreateWithFlags(&e, hipEventDisableTiming);
    int* d; hipMalloc(&d, 4);
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto us = [](auto t0, auto t1, int n) { return std::chrono::duration<double, std::micro>(t1 - t0).count() / n; };
    const int N = 2000;
    for (int i = 0; i < 100; ++i) { hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, a, d); }
    hipDeviceSynchronize();
    auto t0 = now();
    for (int i = 0; i < N; ++i) hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, a, d);
    auto t1 = now();
    hipDeviceSynchronize();
    printf("kernel launch                      %.2f us\n", us(t0, t1, N));
    t0 = now();
    for (int i = 0; i < N; ++i) { hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, a, d); hipEventRecord(e, a); }
    t1 = now();
    hipDeviceSynchronize();
    printf("kernel launch + hipEventRecord     %.2f us\n", us(t0, t1, N));
    t0 = now();
    for (int i = 0; i < N; ++i) {
        hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, a, d);
        hipEventRecord(e, a);
        hipStreamWaitEvent(b, e, 0);
        hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, b, d);
    }
    t1 = now();
    hipDeviceSynchronize();
    printf("launch a, record, b waits, launch b %.2f us\n", us(t0, t1, N));
    return 0;
}
