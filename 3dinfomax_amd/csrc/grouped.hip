// Degree-combined posttrans weights of the PNA layer.
//
// The reference concatenates [h | a | amp(D) a | att(D) a] (a = the n_agg aggregators, models/pna.py:229-233, 207) and
// multiplies by W3 [F_out, F_in + S*A] (A = n_agg*F_in).  The scalers are per-node scalars that only depend on the
// in-degree D, so for all nodes of one degree
//     [a | amp a | att a] W_agg^T  =  a ( sum_s c_s(D) W_s )^T  =:  a W_D^T            (re-association only)
// which cuts the K of the dominant GEMM of the layer from S*A to A (12F -> 4F) and lets the aggregation kernel write
// [N, 4F] instead of [N, 12F].  W_D is built here per degree group (a handful of [F_out, A] matrices, ~1 MB), its
// gradient is folded back into the S blocks of dW3:  dW_s = sum_D c_s(D) dW_D.
#include "common.h"

namespace i3d {

constexpr int MAX_GROUPS = 32, MAX_SCALERS = 4;
struct Coef {
    float c[MAX_GROUPS][MAX_SCALERS];
};

// WD[g][n][k] = sum_s coef[g][s] * W[n][f_in + s*A + k]
// S (number of scalers) is a template parameter: the per-scaler registers are then statically indexed (a runtime-indexed
// float4 w[4] went through scratch memory: 14-23 us for 5 MB of traffic)
template <int S>
__global__ void __launch_bounds__(256)
combine_weights_fwd_kernel(const float4* __restrict__ W, int ldw4, int f_in4, int f_out, int A4, int n_groups, Coef coef,
                           float4* __restrict__ WD) {
    I3D_CHAIN_PRIO();
    long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    long per = (long)f_out * A4;
    if (t >= per) return;
    int n = (int)(t / A4), k = (int)(t - (long)n * A4);
    float4 w[S];
#pragma unroll
    for (int s = 0; s < S; ++s) w[s] = W[(long)n * ldw4 + f_in4 + (long)s * A4 + k];
    for (int g = 0; g < n_groups; ++g) {
        float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int s = 0; s < S; ++s) {
            const float c = coef.c[g][s];
            o.x += c * w[s].x; o.y += c * w[s].y; o.z += c * w[s].z; o.w += c * w[s].w;
        }
        WD[(long)g * per + t] = o;
    }
}

// dW[n][f_in + s*A + k] = sum_g coef[g][s] * dWD[g][n][k]
template <int S>
__global__ void __launch_bounds__(256)
combine_weights_bwd_kernel(const float4* __restrict__ dWD, int ldw4, int f_in4, int f_out, int A4, int n_groups, Coef coef,
                           float4* __restrict__ dW) {
    I3D_CHAIN_PRIO();
    long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    long per = (long)f_out * A4;
    if (t >= per) return;
    int n = (int)(t / A4), k = (int)(t - (long)n * A4);
    float4 acc[S];
#pragma unroll
    for (int s = 0; s < S; ++s) acc[s] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int g = 0; g < n_groups; ++g) {
        const float4 d = dWD[(long)g * per + t];
#pragma unroll
        for (int s = 0; s < S; ++s) {
            const float c = coef.c[g][s];
            acc[s].x += c * d.x; acc[s].y += c * d.y; acc[s].z += c * d.z; acc[s].w += c * d.w;
        }
    }
#pragma unroll
    for (int s = 0; s < S; ++s) dW[(long)n * ldw4 + f_in4 + (long)s * A4 + k] = acc[s];
}

// Wcat [2 Fo + Fp, Fh] = [W_s ; W_d ; W_h] (row blocks), bcat = [0 | 0 | b_post]: the three products of a PNA layer that read
// the node features h - P = h [W_s | W_d]^T of the edge block and lin_h = h W_h^T + b of the posttrans block - as ONE GEMM
__global__ void __launch_bounds__(256)
pack_h_weights_kernel(const float4* __restrict__ We, int ldwe4, int Fo, const float4* __restrict__ Wp, int ldwp4, int Fp,
                      const float* __restrict__ bias_p, int Fh4, float4* __restrict__ Wcat, float* __restrict__ bcat) {
    I3D_CHAIN_PRIO();
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int rows = 2 * Fo + Fp;
    if (t < rows) bcat[t] = t < 2 * Fo ? 0.f : (bias_p != nullptr ? bias_p[t - 2 * Fo] : 0.f);
    if (t >= (long)rows * Fh4) return;
    const int r = (int)(t / Fh4), k = (int)(t - (long)r * Fh4);
    float4 v;
    if (r < Fo) v = We[(long)r * ldwe4 + k];
    else if (r < 2 * Fo) v = We[(long)(r - Fo) * ldwe4 + Fh4 + k];
    else v = Wp[(long)(r - 2 * Fo) * ldwp4 + k];
    Wcat[t] = v;
}

#define I3D_COMBINE_LAUNCH(KERNEL, ...)                                                                         \
    switch (n_scalers) {                                                                                        \
        case 1: hipLaunchKernelGGL(KERNEL<1>, dim3(cdiv(items, 256)), dim3(256), 0, (hipStream_t)stream, __VA_ARGS__); break; \
        case 2: hipLaunchKernelGGL(KERNEL<2>, dim3(cdiv(items, 256)), dim3(256), 0, (hipStream_t)stream, __VA_ARGS__); break; \
        case 3: hipLaunchKernelGGL(KERNEL<3>, dim3(cdiv(items, 256)), dim3(256), 0, (hipStream_t)stream, __VA_ARGS__); break; \
        default: hipLaunchKernelGGL(KERNEL<4>, dim3(cdiv(items, 256)), dim3(256), 0, (hipStream_t)stream, __VA_ARGS__); break; \
    }

static int fill_coef(const float* coef_host, int n_groups, int n_scalers, Coef& c) {
    if (n_groups < 1 || n_groups > MAX_GROUPS || n_scalers < 1 || n_scalers > MAX_SCALERS) return -1;
    for (int g = 0; g < n_groups; ++g)
        for (int s = 0; s < n_scalers; ++s) c.c[g][s] = coef_host[g * n_scalers + s];
    return 0;
}

}  // namespace i3d

using namespace i3d;

extern "C" int i3d_pna_combine_weights_fwd(const float* W, int ldw, int f_in, int f_out, int agg_width, int n_groups,
                                           int n_scalers, const float* coef, float* WD, void* stream) {
    I3D_CHECK_ARG(f_in % 4 == 0 && agg_width % 4 == 0 && ldw % 4 == 0, "dimensions must be multiples of 4");
    I3D_CHECK_ARG((((uintptr_t)W | (uintptr_t)WD) & 15) == 0, "16-byte aligned pointers required");
    Coef c;
    I3D_CHECK_ARG(fill_coef(coef, n_groups, n_scalers, c) == 0, "1..32 groups and 1..4 scalers supported");
    long items = (long)f_out * (agg_width / 4);
    I3D_COMBINE_LAUNCH(combine_weights_fwd_kernel, (const float4*)W, ldw / 4, f_in / 4, f_out, agg_width / 4, n_groups, c,
                       (float4*)WD);
    I3D_CHECK_LAUNCH();
    return I3D_OK;
}

extern "C" int i3d_pna_combine_weights_bwd(const float* dWD, int ldw, int f_in, int f_out, int agg_width, int n_groups,
                                           int n_scalers, const float* coef, float* dW, void* stream) {
    I3D_CHECK_ARG(f_in % 4 == 0 && agg_width % 4 == 0 && ldw % 4 == 0, "dimensions must be multiples of 4");
    I3D_CHECK_ARG((((uintptr_t)dW | (uintptr_t)dWD) & 15) == 0, "16-byte aligned pointers required");
    Coef c;
    I3D_CHECK_ARG(fill_coef(coef, n_groups, n_scalers, c) == 0, "1..32 groups and 1..4 scalers supported");
    long items = (long)f_out * (agg_width / 4);
    I3D_COMBINE_LAUNCH(combine_weights_bwd_kernel, (const float4*)dWD, ldw / 4, f_in / 4, f_out, agg_width / 4, n_groups, c,
                       (float4*)dW);
    I3D_CHECK_LAUNCH();
    return I3D_OK;
}

extern "C" int i3d_pna_pack_h_weights(const float* W_edge, int ldw_edge, int f_out_edge, const float* W_post, int ldw_post,
                                      int f_out_post, const float* bias_post, int f_h, float* Wcat, float* bcat, void* stream) {
    I3D_CHECK_ARG(W_edge != nullptr && W_post != nullptr && Wcat != nullptr && bcat != nullptr, "null");
    I3D_CHECK_ARG(f_h % 4 == 0 && ldw_edge % 4 == 0 && ldw_post % 4 == 0 && ldw_edge >= 2 * f_h && ldw_post >= f_h,
                  "dimensions must be multiples of 4");
    I3D_CHECK_ARG((((uintptr_t)W_edge | (uintptr_t)W_post | (uintptr_t)Wcat) & 15) == 0, "16-byte aligned pointers required");
    const long items = (long)(2 * f_out_edge + f_out_post) * (f_h / 4);
    hipLaunchKernelGGL(pack_h_weights_kernel, dim3(cdiv(items, 256)), dim3(256), 0, (hipStream_t)stream, (const float4*)W_edge,
                       ldw_edge / 4, f_out_edge, (const float4*)W_post, ldw_post / 4, f_out_post, bias_post, f_h / 4, (float4*)Wcat,
                       bcat);
    I3D_CHECK_LAUNCH();
    return I3D_OK;
}
