// Probe for a launch thread (round 5): what a kernel launch costs the calling thread (a) directly, (b) when a second thread does
// the hipLaunchKernel calls from a ring; and whether hipStreamWaitValue64 / hipStreamWriteValue64 can order a stream behind
// work another thread has not enqueued yet.
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/asyncq_probe tools/probes/asyncq_probe.hip -lpthread && /tmp/asyncq_probe
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); return 1; } } while (0)

struct Big { float* p; int n; char pad[240]; };
__global__ void k_small(float* p, int n) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1.f; }
__global__ void k_big(Big b) { if (threadIdx.x == 0 && blockIdx.x == 0) b.p[0] += 1.f; }
__global__ void k_spin(float* p, long cycles) {
    const long t0 = wall_clock64();
    while (wall_clock64() - t0 < cycles) {}
    if (threadIdx.x == 0) p[0] += 1.f;
}
__global__ void k_copy(const float* a, float* b) { b[0] = a[0]; }

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

struct Job { const void* fn; dim3 g, b; hipStream_t s; int nargs; unsigned short off[4]; char blob[256]; };
constexpr int RING = 1 << 14;
static Job ring[RING];
static std::atomic<long> head{0}, tail{0};
static std::atomic<bool> stop{false};

static void worker(int dev) {
    (void)hipSetDevice(dev);
    long t = 0;
    while (true) {
        while (head.load(std::memory_order_acquire) == t) {
            if (stop.load()) return;
            __builtin_ia32_pause();
        }
        Job& j = ring[t & (RING - 1)];
        void* argv[4];
        for (int i = 0; i < j.nargs; ++i) argv[i] = j.blob + j.off[i];
        (void)hipLaunchKernel(j.fn, j.g, j.b, argv, 0, j.s);
        ++t;
        tail.store(t, std::memory_order_release);
    }
}

int main() {
    CK(hipSetDevice(0));
    float* p;
    CK(hipMalloc(&p, 4096));
    CK(hipMemset(p, 0, 4096));
    hipStream_t s, s2;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    const int N = 4000;
    // (a) direct
    for (int rep = 0; rep < 2; ++rep) {
        CK(hipDeviceSynchronize());
        double t0 = now();
        for (int i = 0; i < N; ++i) hipLaunchKernelGGL(k_small, dim3(1), dim3(64), 0, s, p, 1);
        double t1 = now();
        CK(hipDeviceSynchronize());
        double t2 = now();
        printf("direct small args : enqueue %.2f us / launch, with drain %.2f\n", (t1 - t0) / N * 1e6, (t2 - t0) / N * 1e6);
    }
    {
        Big b; b.p = p; b.n = 1;
        CK(hipDeviceSynchronize());
        double t0 = now();
        for (int i = 0; i < N; ++i) hipLaunchKernelGGL(k_big, dim3(1), dim3(64), 0, s, b);
        double t1 = now();
        CK(hipDeviceSynchronize());
        double t2 = now();
        printf("direct 256-byte arg: enqueue %.2f us / launch, with drain %.2f\n", (t1 - t0) / N * 1e6, (t2 - t0) / N * 1e6);
    }
    {   // alternating two streams with an event pair every 8 launches
        hipEvent_t ev;
        CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        CK(hipDeviceSynchronize());
        double t0 = now();
        for (int i = 0; i < N; ++i) {
            hipLaunchKernelGGL(k_small, dim3(1), dim3(64), 0, (i & 8) ? s2 : s, p + 64 * ((i & 8) != 0), 1);
            if ((i & 7) == 7) { (void)hipEventRecord(ev, (i & 8) ? s2 : s); (void)hipStreamWaitEvent((i & 8) ? s : s2, ev, 0); }
        }
        double t1 = now();
        CK(hipDeviceSynchronize());
        printf("direct, 2 streams, record+wait every 8: enqueue %.2f us / launch\n", (t1 - t0) / N * 1e6);
    }
    // (b) ring + worker
    std::thread th(worker, 0);
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipDeviceSynchronize());
        double t0 = now();
        long h = head.load();
        for (int i = 0; i < N; ++i) {
            while (h - tail.load(std::memory_order_acquire) >= RING) {}
            Job& j = ring[h & (RING - 1)];
            j.fn = (const void*)k_small; j.g = dim3(1); j.b = dim3(64); j.s = s; j.nargs = 2;
            j.off[0] = 0; j.off[1] = 8;
            memcpy(j.blob, &p, 8);
            int one = 1; memcpy(j.blob + 8, &one, 4);
            ++h;
            head.store(h, std::memory_order_release);
        }
        double t1 = now();
        while (tail.load() != h) {}
        double t2 = now();
        CK(hipDeviceSynchronize());
        double t3 = now();
        printf("ring: producer %.3f us / launch, worker done %.2f, with drain %.2f\n", (t1 - t0) / N * 1e6, (t2 - t0) / N * 1e6,
               (t3 - t0) / N * 1e6);
    }
    // (b2) the worker launching while THIS thread also launches on another stream (lock contention inside the runtime?)
    {
        CK(hipDeviceSynchronize());
        double t0 = now();
        long h = head.load();
        for (int i = 0; i < N; ++i) {
            Job& j = ring[h & (RING - 1)];
            j.fn = (const void*)k_small; j.g = dim3(1); j.b = dim3(64); j.s = s; j.nargs = 2;
            j.off[0] = 0; j.off[1] = 8;
            memcpy(j.blob, &p, 8);
            int one = 1; memcpy(j.blob + 8, &one, 4);
            ++h;
        }
        head.store(h, std::memory_order_release);
        for (int i = 0; i < N; ++i) hipLaunchKernelGGL(k_small, dim3(1), dim3(64), 0, s2, p + 64, 1);
        double t1 = now();
        while (tail.load() != h) {}
        double t2 = now();
        CK(hipDeviceSynchronize());
        printf("both threads launching %d each: this thread %.2f us / launch, worker done at %.2f us / launch\n", N, (t1 - t0) / N * 1e6,
               (t2 - t0) / N * 1e6);
    }
    // (c) value gates
    int can = 0;
    (void)hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0);
    printf("hipDeviceAttributeCanUseStreamWaitValue = %d\n", can);
    uint64_t* flag = nullptr;
    hipError_t e = hipExtMallocWithFlags((void**)&flag, 8, hipMallocSignalMemory);
    printf("hipExtMallocWithFlags(signal) -> %s\n", hipGetErrorString(e));
    if (e == hipSuccess) {
        CK(hipMemset(flag, 0, 8));
        CK(hipMemset(p, 0, 4096));
        CK(hipDeviceSynchronize());
        // stream s2 (the "user" stream) waits for value 1, then copies p[0] -> p[1]; 2 ms later the worker's stream s gets a
        // spin kernel (0.5 ms) that increments p[0], then the write of 1
        e = hipStreamWaitValue64(s2, flag, 1, hipStreamWaitValueGte, ~0ull);
        printf("hipStreamWaitValue64 -> %s\n", hipGetErrorString(e));
        hipLaunchKernelGGL(k_copy, dim3(1), dim3(1), 0, s2, p, p + 1);
        std::this_thread::sleep_for(std::chrono::milliseconds(2));
        hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, s, p, 50000L);
        e = hipStreamWriteValue64(s, flag, 1, 0);
        printf("hipStreamWriteValue64 -> %s\n", hipGetErrorString(e));
        CK(hipDeviceSynchronize());
        float out[2];
        CK(hipMemcpy(out, p, 8, hipMemcpyDeviceToHost));
        printf("gate: p[0] = %.0f, copy behind the gate saw %.0f (1 = ordered)\n", out[0], out[1]);
        // cost: N gate pairs with a kernel on each side
        const int G = 1000;
        CK(hipDeviceSynchronize());
        double t0 = now();
        for (int i = 0; i < G; ++i) {
            hipLaunchKernelGGL(k_small, dim3(1), dim3(64), 0, s, p, 1);
            (void)hipStreamWriteValue64(s, flag, 2 + i, 0);
            (void)hipStreamWaitValue64(s2, flag, 2 + i, hipStreamWaitValueGte, ~0ull);
            hipLaunchKernelGGL(k_small, dim3(1), dim3(64), 0, s2, p + 64, 1);
        }
        double t1 = now();
        CK(hipDeviceSynchronize());
        double t2 = now();
        printf("gate pair (write + wait + 2 launches): host %.2f us, with drain %.2f us per pair\n", (t1 - t0) / G * 1e6, (t2 - t0) / G * 1e6);
        // same with an event pair instead
        hipEvent_t ev;
        CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        t0 = now();
        for (int i = 0; i < G; ++i) {
            hipLaunchKernelGGL(k_small, dim3(1), dim3(64), 0, s, p, 1);
            (void)hipEventRecord(ev, s);
            (void)hipStreamWaitEvent(s2, ev, 0);
            hipLaunchKernelGGL(k_small, dim3(1), dim3(64), 0, s2, p + 64, 1);
        }
        t1 = now();
        CK(hipDeviceSynchronize());
        t2 = now();
        printf("event pair (record + wait + 2 launches): host %.2f us, with drain %.2f us per pair\n", (t1 - t0) / G * 1e6, (t2 - t0) / G * 1e6);
    }
    stop.store(true);
    th.join();
    return 0;
}
