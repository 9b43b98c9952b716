// Probe for the one-shot peer-write all-reduce of synchronised BatchNorm (csrc/peer.hip): do the mechanics hold on this stack?
//   1. hipExtMallocWithFlags(uncached / fine-grained) memory exported with hipIpcGetMemHandle and opened by ANOTHER process
//      on the SAME device (the two-ranks-on-one-GPU test) - which allocation kinds can be shared;
//   2. a kernel of process A spinning on a flag that a kernel of process B writes (both resident at once?) - bounded spin;
//   3. latency of one exchange (write payload + flag to every peer, spin on world flags, sum) next to an empty kernel.
// hipcc --offload-arch=gfx950 -O2 tools/probes/peer_probe.hip -o tools/probes/peer_probe && tools/probes/peer_probe [world]
#include <hip/hip_runtime.h>
#include <sys/wait.h>
#include <unistd.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#define CK(c) do { hipError_t e_ = (c); if (e_ != hipSuccess) { printf("[%d] %s: %s\n", g_rank, #c, hipGetErrorString(e_)); fflush(stdout); _exit(3); } } while (0)
static int g_rank = -1;
constexpr int MAXW = 8, NSLOT = 4, PAY = 1024;      // doubles per payload

struct Box {      // one rank's mailbox
    unsigned long long flag[NSLOT][MAXW];
    double pay[NSLOT][MAXW][PAY];
};
struct Dev { int world, rank; Box* box[MAXW]; };

__global__ void empty_kernel(int* p) { if (p) p[0] = 1; }

__global__ void __launch_bounds__(256) exchange(Dev d, double* buf, int count, unsigned long long seq, long timeout, int* status) {
    const int slot = (int)(seq % NSLOT);
    for (int p = 0; p < d.world; ++p)
        for (int i = threadIdx.x; i < count; i += 256)
            __hip_atomic_store(&d.box[p]->pay[slot][d.rank][i], buf[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
    __syncthreads();
    if (threadIdx.x < d.world)
        __hip_atomic_store(&d.box[threadIdx.x]->flag[slot][d.rank], seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (threadIdx.x < d.world) {
        const long t0 = wall_clock64();
        while (__hip_atomic_load(&d.box[d.rank]->flag[slot][threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != seq) {
            if (wall_clock64() - t0 > timeout) { status[0] = (int)seq; break; }
            __builtin_amdgcn_s_sleep(1);
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
    __syncthreads();
    for (int i = threadIdx.x; i < count; i += 256) {
        double s = 0.0;
        for (int q = 0; q < d.world; ++q) s += __hip_atomic_load(&d.box[d.rank]->pay[slot][q][i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        buf[i] = s;
    }
}

static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static void xfer(int fd_w, int fd_r, const void* out, void* in, size_t n) {
    if (write(fd_w, out, n) != (ssize_t)n) _exit(4);
    size_t got = 0;
    while (got < n) { ssize_t k = read(fd_r, (char*)in + got, n - got); if (k <= 0) _exit(5); got += k; }
}

static int run(int rank, int world, int kind, int to_peer[2], int from_peer[2]) {
    g_rank = rank;
    CK(hipSetDevice(0));
    Box* mine = nullptr;
    hipError_t e = kind == 2 ? hipMalloc((void**)&mine, sizeof(Box))
                             : hipExtMallocWithFlags((void**)&mine, sizeof(Box), kind == 0 ? hipDeviceMallocUncached : hipDeviceMallocFinegrained);
    if (e != hipSuccess) { printf("[%d] allocation kind %d failed: %s\n", rank, kind, hipGetErrorString(e)); return 2; }
    CK(hipMemset(mine, 0, sizeof(Box)));
    CK(hipDeviceSynchronize());
    Dev d; d.world = world; d.rank = rank;
    for (int i = 0; i < MAXW; ++i) d.box[i] = mine;
    if (world == 2) {
        hipIpcMemHandle_t h, other;
        e = hipIpcGetMemHandle(&h, mine);
        if (e != hipSuccess) { printf("[%d] hipIpcGetMemHandle (kind %d) failed: %s\n", rank, kind, hipGetErrorString(e)); return 2; }
        xfer(to_peer[1], from_peer[0], &h, &other, sizeof(h));
        void* p = nullptr;
        e = hipIpcOpenMemHandle(&p, other, hipIpcMemLazyEnablePeerAccess);
        if (e != hipSuccess) { printf("[%d] hipIpcOpenMemHandle (kind %d) failed: %s\n", rank, kind, hipGetErrorString(e)); return 2; }
        d.box[1 - rank] = (Box*)p;
    }
    int* status;
    CK(hipHostMalloc((void**)&status, 64, hipHostMallocMapped));
    status[0] = 0;
    double* buf;
    CK(hipMalloc((void**)&buf, PAY * 8));
    hipStream_t s;
    CK(hipStreamCreate(&s));
    const int count = 401;
    double host[PAY];
    unsigned long long seq = 0;
    const long timeout = 100000000L * 3;      // 3 s of the 100 MHz wall clock
    // correctness: 200 rounds of different values
    int bad = 0;
    for (int r = 0; r < 200; ++r) {
        for (int i = 0; i < count; ++i) host[i] = (rank + 1) * 1000.0 + r + i * 0.25;
        CK(hipMemcpyAsync(buf, host, count * 8, hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(exchange, dim3(1), dim3(256), 0, s, d, buf, count, ++seq, timeout, status);
        CK(hipMemcpyAsync(host, buf, count * 8, hipMemcpyDeviceToHost, s));
        CK(hipStreamSynchronize(s));
        for (int i = 0; i < count; ++i) {
            double want = 0;
            for (int q = 0; q < world; ++q) want += (q + 1) * 1000.0 + r + i * 0.25;
            if (host[i] != want) ++bad;
        }
        if (status[0]) { printf("[%d] kind %d: spin timed out at seq %d\n", rank, kind, status[0]); return 2; }
    }
    // latency: 200 back-to-back exchanges vs 200 empty kernels
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    float ms_x = 0, ms_e = 0;
    CK(hipEventRecord(a, s));
    for (int r = 0; r < 200; ++r) hipLaunchKernelGGL(exchange, dim3(1), dim3(256), 0, s, d, buf, count, ++seq, timeout, status);
    CK(hipEventRecord(b, s));
    CK(hipStreamSynchronize(s));
    CK(hipEventElapsedTime(&ms_x, a, b));
    CK(hipEventRecord(a, s));
    for (int r = 0; r < 200; ++r) hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(256), 0, s, (int*)nullptr);
    CK(hipEventRecord(b, s));
    CK(hipStreamSynchronize(s));
    CK(hipEventElapsedTime(&ms_e, a, b));
    printf("[%d] world %d kind %s: %s (%d wrong values), exchange %.2f us, empty kernel %.2f us per launch, status %d\n", rank, world,
           kind == 0 ? "uncached" : kind == 1 ? "fine-grained" : "hipMalloc", bad == 0 ? "OK" : "WRONG", bad, ms_x * 5.0, ms_e * 5.0, status[0]);
    fflush(stdout);
    return bad == 0 ? 0 : 1;
}

int main(int argc, char** argv) {
    const int world = argc > 1 ? atoi(argv[1]) : 2;
    for (int kind = 0; kind < 3; ++kind) {
        int p01[2], p10[2];
        if (pipe(p01) || pipe(p10)) return 9;
        pid_t pids[2];
        for (int r = 0; r < world; ++r) {
            pids[r] = fork();          // before any HIP call of this process
            if (pids[r] == 0) {
                int rc = r == 0 ? run(0, world, kind, p01, p10) : run(1, world, kind, p10, p01);
                fflush(stdout);
                _exit(rc);
            }
        }
        for (int r = 0; r < world; ++r) { int st; waitpid(pids[r], &st, 0); }
    }
    return 0;
}
