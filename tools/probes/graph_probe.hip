// Does a hipGraph take the host off the critical path of a chain of small dependent kernels?  Host cost and device span of
//   (a) 300 direct launches on one stream,
//   (b) the same 300 kernels as ONE instantiated hipGraph, replayed (static shapes: no update),
//   (c) re-captured every iteration + hipGraphExecUpdate + launch (what a step with per-batch shapes would need:
//       N, E and every grid change from batch to batch),
//   (d) (b) with hipGraphExecKernelNodeSetParams on every node before the replay (per-batch parameter patching).
// hipcc --offload-arch=gfx950 -O2 tools/probes/graph_probe.hip -o tools/probes/graph_probe && tools/probes/graph_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>

__global__ void small(float* x, int n, int spin) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        float v = x[i];
        for (int k = 0; k < spin; ++k) v = v * 1.0001f + 0.5f;
        x[i] = v;
    }
}

static double now() {
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
#define CK(c) do { hipError_t e_ = (c); if (e_ != hipSuccess) { printf("%s: %s\n", #c, hipGetErrorString(e_)); return 1; } } while (0)

int main() {
    const int K = 300, n = 1 << 18, iters = 30;
    float* x;
    CK(hipMalloc(&x, n * sizeof(float)));
    CK(hipMemset(x, 0, n * sizeof(float)));
    hipStream_t s;
    CK(hipStreamCreate(&s));
    for (int spin : {8, 400}) {        // ~3 us and ~12 us kernels
        auto direct = [&](int grid) { for (int k = 0; k < K; ++k) hipLaunchKernelGGL(small, dim3(grid), dim3(256), 0, s, x, n, spin); };
        direct(n / 256);
        CK(hipStreamSynchronize(s));
        double h = 0, t = 0;
        for (int it = 0; it < iters; ++it) {
            double t0 = now();
            direct(n / 256);
            double t1 = now();
            CK(hipStreamSynchronize(s));
            h += t1 - t0; t += now() - t0;
        }
        printf("spin %3d  (a) direct launches      : host %7.1f us  total %7.1f us  (%.2f us host per kernel)\n", spin, h / iters, t / iters, h / iters / K);
        // (b) static graph
        hipGraph_t g;
        hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        direct(n / 256);
        CK(hipStreamEndCapture(s, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        CK(hipGraphLaunch(ge, s));
        CK(hipStreamSynchronize(s));
        h = t = 0;
        for (int it = 0; it < iters; ++it) {
            double t0 = now();
            CK(hipGraphLaunch(ge, s));
            double t1 = now();
            CK(hipStreamSynchronize(s));
            h += t1 - t0; t += now() - t0;
        }
        printf("spin %3d  (b) graph replay         : host %7.1f us  total %7.1f us\n", spin, h / iters, t / iters);
        // (c) re-capture + exec update + launch (grid changes every iteration, like N / E of a batch)
        h = t = 0;
        double hc = 0, hu = 0;
        for (int it = 0; it < iters; ++it) {
            double t0 = now();
            hipGraph_t g2;
            CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
            direct(n / 256 - (it % 7));
            CK(hipStreamEndCapture(s, &g2));
            double tc = now();
            hipGraphNode_t err_node;
            hipGraphExecUpdateResult res;
            hipError_t e = hipGraphExecUpdate(ge, g2, &err_node, &res);
            if (e != hipSuccess) { printf("hipGraphExecUpdate: %s (result %d)\n", hipGetErrorString(e), (int)res); break; }
            double tu = now();
            CK(hipGraphLaunch(ge, s));
            double t1 = now();
            CK(hipStreamSynchronize(s));
            CK(hipGraphDestroy(g2));
            hc += tc - t0; hu += tu - tc; h += t1 - t0; t += now() - t0;
        }
        printf("spin %3d  (c) capture+update+launch: host %7.1f us  total %7.1f us  (capture %.1f, update %.1f)\n", spin, h / iters, t / iters, hc / iters, hu / iters);
        // (d) per-node parameter patching + replay
        size_t nn = 0;
        CK(hipGraphGetNodes(g, nullptr, &nn));
        std::vector<hipGraphNode_t> nodes(nn);
        CK(hipGraphGetNodes(g, nodes.data(), &nn));
        h = t = 0;
        bool ok = true;
        for (int it = 0; it < iters && ok; ++it) {
            int grid = n / 256 - (it % 5);
            int nv = n, sp = spin;
            void* args[3] = {&x, &nv, &sp};
            hipKernelNodeParams p = {};
            p.func = (void*)small; p.gridDim = dim3(grid); p.blockDim = dim3(256); p.kernelParams = args; p.sharedMemBytes = 0;
            double t0 = now();
            for (size_t k = 0; k < nn; ++k) {
                hipError_t e = hipGraphExecKernelNodeSetParams(ge, nodes[k], &p);
                if (e != hipSuccess) { printf("hipGraphExecKernelNodeSetParams: %s\n", hipGetErrorString(e)); ok = false; break; }
            }
            CK(hipGraphLaunch(ge, s));
            double t1 = now();
            CK(hipStreamSynchronize(s));
            h += t1 - t0; t += now() - t0;
        }
        if (ok) printf("spin %3d  (d) set params x%zu + replay: host %7.1f us  total %7.1f us\n", spin, nn, h / iters, t / iters);
        CK(hipGraphExecDestroy(ge));
        CK(hipGraphDestroy(g));
    }
    return 0;
}
