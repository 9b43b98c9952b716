// Block copies between per-tower parameter tensors and the stacked ("all towers of a layer as one wide layer") buffers of the
// tower variant: reference models/pna_original.py:264-319 runs `towers` small PNATowers per layer and concatenates their
// outputs; here a layer's towers are ONE edge block [2 in + edge -> towers * F_t], ONE aggregation over [E, towers * F_t]
// and ONE posttrans block [in + 12 towers F_t -> towers * F_out] whose weight matrices hold the towers' weights as blocks
// (zeros elsewhere: a tower reads only its own columns).  The stacked buffers are rebuilt from the parameters by ONE
// launch per model forward and the stacked gradients are scattered back by ONE launch per backward; the table of blocks is
// built once per model (3dinfomax_amd/pna_original.py: _TowerStacks).
#include "common.h"

#include <algorithm>

namespace i3d {

namespace {

__global__ void __launch_bounds__(256) block_copy_kernel(const I3dCopyBlock* __restrict__ table, int reverse) {
    const I3dCopyBlock b = table[blockIdx.x];
    const float* src = reverse ? b.dst : b.src;
    float* dst = reverse ? const_cast<float*>(b.src) : b.dst;
    const long ls = reverse ? b.ld_dst : b.ld_src, ld = reverse ? b.ld_src : b.ld_dst;
    const int n = b.rows * b.cols;
    for (int i = threadIdx.x; i < n; i += 256) {
        const int r = i / b.cols, c = i - r * b.cols;
        dst[r * ld + c] = src[r * ls + c];
    }
}

// dst[r, c] = c < cols_src ? src[r, c] : 0 for c < cols_dst: widens rows to a pitch the 16-byte kernels take (zeros in the new
// columns) or crops them back; a thread moves one float4 of dst when cols_dst % 4 == 0
__global__ void __launch_bounds__(256) copy_cols_kernel(const float* __restrict__ src, int rows, int cols_src, float* __restrict__ dst,
                                                        int cols_dst, int vec) {
    const long n = vec ? (long)rows * (cols_dst / 4) : (long)rows * cols_dst;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        if (vec) {
            const int q = cols_dst / 4;
            const long r = i / q;
            const int c = (int)(i - r * q) * 4;
            const float* s = src + r * cols_src + c;
            float4 v;
            v.x = c < cols_src ? s[0] : 0.f;
            v.y = c + 1 < cols_src ? s[1] : 0.f;
            v.z = c + 2 < cols_src ? s[2] : 0.f;
            v.w = c + 3 < cols_src ? s[3] : 0.f;
            *reinterpret_cast<float4*>(dst + r * cols_dst + c) = v;
        } else {
            const long r = i / cols_dst;
            const int c = (int)(i - r * cols_dst);
            dst[i] = c < cols_src ? src[r * cols_src + c] : 0.f;
        }
    }
}

// out[j * ld + col] = |x[src_j] - x[dst_j]|^2 (three coordinates; squares, then sums in order, no contraction: torch's
// sum((a - b) ** 2, dim=-1) bit for bit)
__global__ void __launch_bounds__(256) edge_sqdist_kernel(const float* __restrict__ x, const int* __restrict__ src, const int* __restrict__ dst,
                                                          int E, float* __restrict__ out, int ld, int col, int take_sqrt) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= E) return;
    const float* a = x + 3L * src[j];
    const float* b = x + 3L * dst[j];
    const float d0 = a[0] - b[0], d1 = a[1] - b[1], d2 = a[2] - b[2];
    const float sq = __fadd_rn(__fadd_rn(__fmul_rn(d0, d0), __fmul_rn(d1, d1)), __fmul_rn(d2, d2));
    out[(long)j * ld + col] = take_sqrt ? __fsqrt_rn(sq) : sq;
}

}  // namespace
}  // namespace i3d

using namespace i3d;

extern "C" int i3d_edge_sqdist(const float* x, const int* src, const int* dst, int num_edges, float* out, int ld, int col, int take_sqrt,
                               void* stream) {
    I3D_CHECK_ARG(num_edges >= 0 && ld > col && col >= 0 && (num_edges == 0 || (x != nullptr && src != nullptr && dst != nullptr && out != nullptr)),
                  "bad arguments");
    if (num_edges == 0) return I3D_OK;
    hipLaunchKernelGGL(edge_sqdist_kernel, dim3(cdiv(num_edges, 256)), dim3(256), 0, (hipStream_t)stream, x, src, dst, num_edges, out, ld, col,
                       take_sqrt);
    I3D_CHECK_LAUNCH();
    return I3D_OK;
}

extern "C" int i3d_copy_cols(const float* src, int rows, int cols_src, float* dst, int cols_dst, void* stream) {
    I3D_CHECK_ARG(rows >= 0 && cols_src >= 0 && cols_dst >= 0 && (rows == 0 || cols_dst == 0 || (src != nullptr && dst != nullptr)),
                  "bad arguments");
    if (rows == 0 || cols_dst == 0) return I3D_OK;
    const int vec = (cols_dst % 4 == 0) && (((uintptr_t)dst & 15) == 0);
    const long n = vec ? (long)rows * (cols_dst / 4) : (long)rows * cols_dst;
    const int grid = (int)std::min<long>((n + 255) / 256, 4096);
    hipLaunchKernelGGL(copy_cols_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, src, rows, cols_src, dst, cols_dst, vec);
    I3D_CHECK_LAUNCH();
    return I3D_OK;
}

// table: n_blocks entries in DEVICE memory; reverse != 0 copies dst -> src (the same table scatters results back)
extern "C" int i3d_block_copy(const I3dCopyBlock* table, int n_blocks, int reverse, void* stream) {
    I3D_CHECK_ARG(table != nullptr && n_blocks >= 0, "bad arguments");
    if (n_blocks == 0) return I3D_OK;
    hipLaunchKernelGGL(block_copy_kernel, dim3(n_blocks), dim3(256), 0, (hipStream_t)stream, table, reverse);
    I3D_CHECK_LAUNCH();
    return I3D_OK;
}
