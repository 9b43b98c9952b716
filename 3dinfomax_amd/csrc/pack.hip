// Block copies between per-tower parameter tensors and the stacked ("all towers of a layer as one wide layer") buffers of the
// tower variant: reference models/pna_original.py:264-319 runs `towers` small PNATowers per layer and concatenates their
// outputs; here a layer's towers are ONE edge block [2 in + edge -> towers * F_t], ONE aggregation over [E, towers * F_t]
// and ONE posttrans block [in + 12 towers F_t -> towers * F_out] whose weight matrices hold the towers' weights as blocks
// (zeros elsewhere: a tower reads only its own columns).  The stacked buffers are rebuilt from the parameters by ONE
// launch per model forward and the stacked gradients are scattered back by ONE launch per backward; the table of blocks is
// built once per model (3dinfomax_amd/pna_original.py: _TowerStacks).
#include "common.h"

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

}  // namespace
}  // namespace i3d

using namespace i3d;

// table: n_blocks entries in DEVICE memory; reverse != 0 copies dst -> src (the same table scatters results back)
extern "C" int i3d_block_copy(const I3dCopyBlock* table, int n_blocks, int reverse, void* stream) {
    I3D_CHECK_ARG(table != nullptr && n_blocks >= 0, "bad arguments");
    if (n_blocks == 0) return I3D_OK;
    hipLaunchKernelGGL(block_copy_kernel, dim3(n_blocks), dim3(256), 0, (hipStream_t)stream, table, reverse);
    I3D_CHECK_LAUNCH();
    return I3D_OK;
}
