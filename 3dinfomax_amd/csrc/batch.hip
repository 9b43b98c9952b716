// Device-side construction of the complete 3D distance graphs of a batch.
//
// Replaces, per training step, B x { QM9Dataset.get_complete_graph (reference datasets/qm9_dataset.py:233-244:
// src = repeat_interleave(arange(n), n-1), dst = all j != src, d = ||x_src - x_dst||_2) } + dgl.batch
// (reference datasets/custom_collate.py:108-109) by one kernel over the E3 = sum n(n-1) edges: everything about a
// complete graph is analytic, so only the coordinates [N,3] and the node offsets [B+1] cross PCIe.
//
// For molecule g with n atoms and node/edge offsets (nb, eb), local edge (u -> v), u != v:
//   edge id  (reference order, source-major)      id   = eb + u*(n-1) + (v < u ? v : v-1)
//   epos     (destination-sorted, stable in id)   epos = eb + v*(n-1) + (u < v ? u : u-1)
// in_ptr = out_ptr = nb-th node starts at eb + local*(n-1).
#include "common.h"

namespace i3d {

__device__ __forceinline__ int find_graph(const int* __restrict__ ptr, int num_graphs, int x) {
    int lo = 0, hi = num_graphs;          // ptr[lo] <= x < ptr[hi]
    while (hi - lo > 1) {
        int mid = (lo + hi) >> 1;
        if (ptr[mid] <= x) lo = mid; else hi = mid;
    }
    return lo;
}

__global__ void __launch_bounds__(256)
complete_graph_kernel(const float* __restrict__ coords, const int* __restrict__ graph_ptr,
                      const int* __restrict__ edge_ptr, int num_graphs, int num_nodes, int num_edges,
                      int* __restrict__ in_ptr, int* __restrict__ src_s, int* __restrict__ dst_s,
                      int* __restrict__ perm, int* __restrict__ inv_perm, int* __restrict__ out_epos,
                      int64_t* __restrict__ src_id, int64_t* __restrict__ dst_id, float* __restrict__ d_id) {
    I3D_CHAIN_PRIO();
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t <= num_nodes) {     // node-level: CSR row pointer (identical by destination and by source)
        if (t == num_nodes) {
            in_ptr[t] = num_edges;
        } else {
            int g = find_graph(graph_ptr, num_graphs, t);
            int nb = graph_ptr[g], n = graph_ptr[g + 1] - nb;
            in_ptr[t] = edge_ptr[g] + (t - nb) * (n - 1);
        }
    }
    if (t >= num_edges) return;
    // t is an EDGE ID (source-major order of the reference)
    int g = find_graph(edge_ptr, num_graphs, t);
    int nb = graph_ptr[g], n = graph_ptr[g + 1] - nb, eb = edge_ptr[g];
    int loc = t - eb;
    int u = loc / (n - 1), j = loc - u * (n - 1);
    int v = j < u ? j : j + 1;
    int epos = eb + v * (n - 1) + (u < v ? u : u - 1);
    const float* a = coords + (long)(nb + u) * 3;
    const float* b = coords + (long)(nb + v) * 3;
    float dx = a[0] - b[0], dy = a[1] - b[1], dz = a[2] - b[2];
    float dist = sqrtf(dx * dx + dy * dy + dz * dz);
    src_id[t] = nb + u;
    dst_id[t] = nb + v;
    d_id[t] = dist;
    src_s[epos] = nb + u;
    dst_s[epos] = nb + v;
    perm[epos] = t;
    inv_perm[t] = epos;
    // out-edges of u grouped by source, ordered by destination: slot (u, j) holds the epos of edge u -> v
    out_epos[t] = epos;
}

}  // namespace i3d

using namespace i3d;

extern "C" int i3d_complete_graph_build(const float* coords, const int* graph_ptr, const int* edge_ptr, int num_graphs,
                                        int num_nodes, int num_edges, int* in_ptr, int* src_s, int* dst_s, int* perm,
                                        int* inv_perm, int* out_epos, int64_t* src_id, int64_t* dst_id, float* d_id,
                                        void* stream) {
    I3D_CHECK_ARG(num_graphs > 0 && num_nodes > 0 && num_edges >= 0, "bad shape");
    int items = num_edges > num_nodes + 1 ? num_edges : num_nodes + 1;
    hipLaunchKernelGGL(complete_graph_kernel, dim3(cdiv(items, 256)), dim3(256), 0, (hipStream_t)stream, coords, graph_ptr,
                       edge_ptr, num_graphs, num_nodes, num_edges, in_ptr, src_s, dst_s, perm, inv_perm, out_epos, src_id,
                       dst_id, d_id);
    I3D_CHECK_LAUNCH();
    return I3D_OK;
}
