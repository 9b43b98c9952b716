// The gates of one GRU step (reference models/pna_original.py:64-84: `GRU`, a one-step nn.GRU between the layers of PNAGNNOriginal
// with gru_enable=True, :190-193).  The two products GI = x W_ih^T + b_ih and GH = h0 W_hh^T + b_hh are the library's GEMMs
// (3dinfomax_amd/pna_original.py: _GRUCellFn); this file is the elementwise part, torch's gate order (r | z | n):
//   r = sigmoid(GI_r + GH_r), z = sigmoid(GI_z + GH_z), n = tanh(GI_n + r GH_n), h' = (1 - z) n + z h0
#include "common.h"

namespace i3d {
namespace {

__global__ void __launch_bounds__(256) gru_gates_fwd_kernel(const float* __restrict__ GI, const float* __restrict__ GH,
                                                            const float* __restrict__ h0, long items, int H, float* __restrict__ out,
                                                            float* __restrict__ S) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= items) return;
    const long row = i / H;
    const int c = (int)(i - row * H);
    const float* gi = GI + row * 3 * H;
    const float* gh = GH + row * 3 * H;
    const float r = 1.f / (1.f + expf(-(gi[c] + gh[c])));
    const float z = 1.f / (1.f + expf(-(gi[H + c] + gh[H + c])));
    const float n = tanhf(gi[2 * H + c] + r * gh[2 * H + c]);
    out[i] = (1.f - z) * n + z * h0[i];
    float* s = S + row * 3 * H;
    s[c] = r; s[H + c] = z; s[2 * H + c] = n;
}

__global__ void __launch_bounds__(256) gru_gates_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ S,
                                                            const float* __restrict__ GH, const float* __restrict__ h0, long items, int H,
                                                            float* __restrict__ dGI, float* __restrict__ dGH, float* __restrict__ dh0) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= items) return;
    const long row = i / H;
    const int c = (int)(i - row * H);
    const float* s = S + row * 3 * H;
    const float r = s[c], z = s[H + c], n = s[2 * H + c];
    const float hn = GH[row * 3 * H + 2 * H + c];
    const float g = dout[i];
    const float dpn = g * (1.f - z) * (1.f - n * n);
    const float dpr = dpn * hn * r * (1.f - r);
    const float dpz = g * (h0[i] - n) * z * (1.f - z);
    float* a = dGI + row * 3 * H;
    float* b = dGH + row * 3 * H;
    a[c] = dpr; a[H + c] = dpz; a[2 * H + c] = dpn;
    b[c] = dpr; b[H + c] = dpz; b[2 * H + c] = dpn * r;
    dh0[i] = g * z;
}

}  // namespace
}  // namespace i3d

using namespace i3d;

extern "C" int i3d_gru_gates_fwd(const float* GI, const float* GH, const float* h0, int rows, int hidden, float* out, float* saved,
                                 void* stream) {
    I3D_CHECK_ARG(rows >= 0 && hidden > 0 && (rows == 0 || (GI && GH && h0 && out && saved)), "bad arguments");
    if (rows == 0) return I3D_OK;
    const long items = (long)rows * hidden;
    hipLaunchKernelGGL(gru_gates_fwd_kernel, dim3(cdiv(items, 256)), dim3(256), 0, (hipStream_t)stream, GI, GH, h0, items, hidden, out, saved);
    I3D_CHECK_LAUNCH();
    return I3D_OK;
}

extern "C" int i3d_gru_gates_bwd(const float* grad_out, const float* saved, const float* GH, const float* h0, int rows, int hidden,
                                 float* grad_GI, float* grad_GH, float* grad_h0, void* stream) {
    I3D_CHECK_ARG(rows >= 0 && hidden > 0 && (rows == 0 || (grad_out && saved && GH && h0 && grad_GI && grad_GH && grad_h0)), "bad arguments");
    if (rows == 0) return I3D_OK;
    const long items = (long)rows * hidden;
    hipLaunchKernelGGL(gru_gates_bwd_kernel, dim3(cdiv(items, 256)), dim3(256), 0, (hipStream_t)stream, grad_out, saved, GH, h0, items, hidden,
                       grad_GI, grad_GH, grad_h0);
    I3D_CHECK_LAUNCH();
    return I3D_OK;
}
