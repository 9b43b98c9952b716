// Edge-side kernels: gather-combine (the fused replacement of "gather src/dst rows, concat, first Linear"),
// segmented sums for the gather backward and Net3D's mean reduce, Fourier distance features and Net3D's
// soft edge gate.  All HBM-bound; one lane owns one (row, 4-feature) item with 16-byte accesses, neighbour
// rows of a molecule sit next to each other so the P[src] / P[dst] gathers hit L2.
#include "common.h"

namespace i3d {

// pre[j,:] = P[src[j], 0:F] + P[dst[j], F:2F] + Q[q_code ? q_code[j] : j, :] + bias
// reference models/pna.py:237-252 (cat[src,dst,edge] -> Linear) via  [a|b|c] W^T = a Ws^T + b Wd^T + c Wq^T
template <int V>
__global__ void __launch_bounds__(256)
edge_combine_fwd_kernel(const float* __restrict__ P, int ldp, const float* __restrict__ Q, const int* __restrict__ q_code,
                        const float* __restrict__ bias, const int* __restrict__ src, const int* __restrict__ dst,
                        int E, int feat, float* __restrict__ pre) {
    I3D_CHAIN_PRIO();
    const int FV = feat / V;
    long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long)E * FV) return;
    int j = (int)(t / FV), c = (int)(t - (long)j * FV) * V;
    const float* ps = P + (long)src[j] * ldp + c;
    const float* pd = P + (long)dst[j] * ldp + feat + c;
    float* o = pre + (long)j * feat + c;
    const long qrow = q_code ? q_code[j] : j;
    if (V == 4) {
        float4 a = *reinterpret_cast<const float4*>(ps);
        float4 b = *reinterpret_cast<const float4*>(pd);
        float4 d = Q ? *reinterpret_cast<const float4*>(Q + qrow * feat + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        float4 r = make_float4(a.x + b.x + d.x, a.y + b.y + d.y, a.z + b.z + d.z, a.w + b.w + d.w);
        if (bias) {
            float4 bb = *reinterpret_cast<const float4*>(bias + c);
            r.x += bb.x; r.y += bb.y; r.z += bb.z; r.w += bb.w;
        }
        *reinterpret_cast<float4*>(o) = r;
    } else {
        float r = ps[0] + pd[0] + (Q ? Q[qrow * feat + c] : 0.f);
        if (bias) r += bias[c];
        o[0] = r;
    }
}

// Categorical edge features with a small joint vocabulary (bonds: 5 x 6 x 2 = 60 combinations): the embedding sum of an
// edge is a row of the [V, F] table of all combinations, so  ef W_q^T  is a gather from (table W_q^T) and the
// gradients reduce over the one-hot matrix.  codes[j] = sum_c idx[row(j), c] * stride[c];  onehot[j, codes[j]] = 1.
struct CodeStrides { int s[8]; };
__global__ void __launch_bounds__(256)
edge_codes_kernel(const long* __restrict__ idx, const int* __restrict__ row_perm, int rows, int n_cols, CodeStrides st,
                  int v_pad, int* __restrict__ codes, float* __restrict__ onehot) {
    I3D_CHAIN_PRIO();
    const int VQ = v_pad / 4;
    long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long)rows * VQ) return;
    const int j = (int)(t / VQ), q = (int)(t - (long)j * VQ);
    const long row = row_perm ? row_perm[j] : j;
    int code = 0;
    for (int c = 0; c < n_cols; ++c) code += (int)idx[row * n_cols + c] * st.s[c];
    if (q == 0) codes[j] = code;
    const int c0 = q * 4;
    *reinterpret_cast<float4*>(onehot + (long)j * v_pad + c0) =
        make_float4(code == c0 ? 1.f : 0.f, code == c0 + 1 ? 1.f : 0.f, code == c0 + 2 ? 1.f : 0.f, code == c0 + 3 ? 1.f : 0.f);
}

// Multi-hot encoding of the categorical feature columns: out[j, offset[c] + idx[row(j), c]] = 1 for every column c, else 0
// (offset = prefix sum of the table sizes).  The gradient of ALL embedding tables of an encoder is then one product
// out^T dY ([sum of table sizes, F], split-K through the scratch: deterministic) instead of atomics.
struct ColOffsets { int o[16]; };
__global__ void __launch_bounds__(256)
multihot_kernel(const long* __restrict__ idx, const int* __restrict__ row_perm, int rows, int n_cols, ColOffsets off, int v_pad,
                float* __restrict__ out) {
    I3D_CHAIN_PRIO();
    const int VQ = v_pad / 4;
    long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long)rows * VQ) return;
    const int j = (int)(t / VQ), q = (int)(t - (long)j * VQ);
    const long row = row_perm ? row_perm[j] : j;
    const int c0 = q * 4;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    for (int c = 0; c < n_cols; ++c) {
        const int pos = off.o[c] + (int)idx[row * n_cols + c] - c0;
        if (pos >= 0 && pos < 4) v[pos] = 1.f;
    }
    *reinterpret_cast<float4*>(out + (long)j * v_pad + c0) = make_float4(v[0], v[1], v[2], v[3]);
}

// out[v, :] = scale * sum_{j in [ptr[v], ptr[v+1])} x[idx ? idx[j] : j, :]
// PAIR: two segmentations of the same rows in one launch (the edge block's dP[src] over the out-edges and dP[dst] over the
// in-edges): items [0, nseg FV) take (ptr, idx, out), the next nseg FV take (ptr1, idx1, out1)
template <int V, bool PAIR = false, bool XB16 = false>
__global__ void __launch_bounds__(256)
segment_sum_kernel(const float* __restrict__ x, int ldx, const int* __restrict__ ptr, const int* __restrict__ idx,
                   int nseg, int feat, int scale_mode, float* __restrict__ out, int ldo,
                   const int* __restrict__ ptr1 = nullptr, const int* __restrict__ idx1 = nullptr, float* __restrict__ out1 = nullptr) {
    I3D_CHAIN_PRIO();
    const int FV = feat / V;
    long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (PAIR) {
        if (t >= 2L * nseg * FV) return;
        if (t >= (long)nseg * FV) { t -= (long)nseg * FV; ptr = ptr1; idx = idx1; out = out1; }
    } else if (t >= (long)nseg * FV) return;
    int v = (int)(t / FV), c = (int)(t - (long)v * FV) * V;
    int beg = ptr[v], end = ptr[v + 1];
    float acc[V];
#pragma unroll
    for (int i = 0; i < V; ++i) acc[i] = 0.f;
    // four rows per trip, loaded unconditionally (clamped index; the duplicates hit L1) so that four loads - and their
    // four index loads - are in flight instead of one load-add per trip; the additions keep the order j = beg .. end-1
    for (int j = beg; j < end; j += 4) {
        long rows[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int jj = min(j + k, end - 1);
            rows[k] = idx ? idx[jj] : jj;
        }
        float a[4][V];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float* p = x + rows[k] * ldx + c;
            if constexpr (XB16) {      // rows of bf16 (V == 4): 4 values = 8 bytes
                const uint2 t2 = *reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(x) + rows[k] * ldx + c);
                a[k][0] = __uint_as_float(t2.x << 16); a[k][1 % V] = __uint_as_float(t2.x & 0xffff0000u);
                a[k][2 % V] = __uint_as_float(t2.y << 16); a[k][3 % V] = __uint_as_float(t2.y & 0xffff0000u);
            } else if (V == 4) {
                const float4 t4 = *reinterpret_cast<const float4*>(p);
                a[k][0] = t4.x; a[k][1 % V] = t4.y; a[k][2 % V] = t4.z; a[k][3 % V] = t4.w;
            } else {
                a[k][0] = p[0];
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (j + k < end) {
#pragma unroll
                for (int i = 0; i < V; ++i) acc[i] += a[k][i];
            }
        }
    }
    if (scale_mode == 1) {   // DGL fn.mean divides the sum by the in-degree
#pragma unroll
        for (int i = 0; i < V; ++i) acc[i] = acc[i] / (float)max(end - beg, 1);
    }
    float* o = out + (long)v * ldo + c;
    if (V == 4) *reinterpret_cast<float4*>(o) = make_float4(acc[0], acc[1 % V], acc[2 % V], acc[3 % V]);
    else o[0] = acc[0];
}

// out[j,:] = scale(seg) * g[seg,:]  with seg = seg_of_row[j]
template <int V>
__global__ void __launch_bounds__(256)
segment_bcast_kernel(const float* __restrict__ g, const int* __restrict__ ptr, const int* __restrict__ seg_of_row,
                     int rows, int feat, int scale_mode, float* __restrict__ out) {
    I3D_CHAIN_PRIO();
    const int FV = feat / V;
    long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long)rows * FV) return;
    int j = (int)(t / FV), c = (int)(t - (long)j * FV) * V;
    int v = seg_of_row[j];
    float s = 1.f;
    if (scale_mode == 1) s = (float)max(ptr[v + 1] - ptr[v], 1);
    const float* p = g + (long)v * feat + c;
    float* o = out + (long)j * feat + c;
    if (V == 4) {
        float4 a = *reinterpret_cast<const float4*>(p);
        *reinterpret_cast<float4*>(o) = make_float4(a.x / s, a.y / s, a.z / s, a.w / s);
    } else {
        o[0] = p[0] / s;
    }
}

template <int V>
__global__ void __launch_bounds__(256)
gather_rows_kernel(const float* __restrict__ x, const int* __restrict__ idx, int rows, int feat,
                   float* __restrict__ out) {
    I3D_CHAIN_PRIO();
    const int FV = feat / V;
    long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long)rows * FV) return;
    int j = (int)(t / FV), c = (int)(t - (long)j * FV) * V;
    const float* p = x + (long)idx[j] * feat + c;
    float* o = out + (long)j * feat + c;
    if (V == 4) *reinterpret_cast<float4*>(o) = *reinterpret_cast<const float4*>(p);
    else o[0] = p[0];
}

// reference commons/utils.py:103-110: [sin(d/2^k)]_k | [cos(d/2^k)]_k | d
__global__ void __launch_bounds__(256)
fourier_encode_kernel(const float* __restrict__ d, int E, int n_enc, float* __restrict__ out) {
    I3D_CHAIN_PRIO();
    int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= E) return;
    float x = d[j];
    float* o = out + (long)j * (2 * n_enc + 1);
    float scale = 1.f;
    for (int k = 0; k < n_enc; ++k) {
        float v = x / scale;            // torch: x / 2**k
        o[k] = sinf(v);
        o[n_enc + k] = cosf(v);
        scale *= 2.f;
    }
    o[2 * n_enc] = x;
}

// reference models/net3d.py:117-118: w = sigmoid(m . ws + bs), msg = m * w.  One lane per edge (feat is small).
__global__ void __launch_bounds__(256)
soft_edge_fwd_kernel(const float* __restrict__ m, const float* __restrict__ ws, const float* __restrict__ bs, int E,
                     int feat, float* __restrict__ msg, float* __restrict__ w) {
    I3D_CHAIN_PRIO();
    int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= E) return;
    const float* p = m + (long)j * feat;
    float dot = bs[0];
    for (int f = 0; f < feat; ++f) dot += p[f] * ws[f];
    float g = 1.f / (1.f + expf(-dot));
    w[j] = g;
    float* o = msg + (long)j * feat;
    for (int f = 0; f < feat; ++f) o[f] = p[f] * g;
}

__global__ void __launch_bounds__(256)
soft_edge_bwd_kernel(const float* __restrict__ gmsg, const float* __restrict__ m, const float* __restrict__ w,
                     const float* __restrict__ ws, int E, int feat, float* __restrict__ gm,
                     float* __restrict__ ggate) {
    I3D_CHAIN_PRIO();
    int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= E) return;
    const float* p = m + (long)j * feat;
    const float* gp = gmsg + (long)j * feat;
    float g = w[j];
    float dot = 0.f;
    for (int f = 0; f < feat; ++f) dot += gp[f] * p[f];
    float gg = dot * g * (1.f - g);
    ggate[j] = gg;
    float* o = gm + (long)j * feat;
    for (int f = 0; f < feat; ++f) o[f] = gp[f] * g + gg * ws[f];
}

// feat <= 32, feat % 4 == 0 (the 3D network: 20): 8 lanes share an edge, one float4 each (lanes >= feat/4 idle), the dot
// products are reduced with three xor-shuffles - 16-byte coalesced accesses instead of 20 strided scalar loads per lane
__global__ void __launch_bounds__(256)
soft_edge_fwd_v8_kernel(const float* __restrict__ m, const float* __restrict__ ws, const float* __restrict__ bs, int E,
                        int feat, float* __restrict__ msg, float* __restrict__ w) {
    I3D_CHAIN_PRIO();
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long j = t >> 3;
    const int q = (int)(t & 7);
    const bool live = j < E && q * 4 < feat;
    float4 x = make_float4(0.f, 0.f, 0.f, 0.f), wv = x;
    if (live) {
        x = *reinterpret_cast<const float4*>(m + j * feat + q * 4);
        wv = *reinterpret_cast<const float4*>(ws + q * 4);
    }
    float dot = x.x * wv.x + x.y * wv.y + x.z * wv.z + x.w * wv.w;
    dot += __shfl_xor(dot, 1, 8);
    dot += __shfl_xor(dot, 2, 8);
    dot += __shfl_xor(dot, 4, 8);
    const float g = 1.f / (1.f + expf(-(dot + bs[0])));
    if (!live) return;
    if (q == 0) w[j] = g;
    *reinterpret_cast<float4*>(msg + j * feat + q * 4) = make_float4(x.x * g, x.y * g, x.z * g, x.w * g);
}

__global__ void __launch_bounds__(256)
soft_edge_bwd_v8_kernel(const float* __restrict__ gmsg, const float* __restrict__ m, const float* __restrict__ w,
                        const float* __restrict__ ws, int E, int feat, float* __restrict__ gm, float* __restrict__ ggate) {
    I3D_CHAIN_PRIO();
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long j = t >> 3;
    const int q = (int)(t & 7);
    const bool live = j < E && q * 4 < feat;
    float4 x = make_float4(0.f, 0.f, 0.f, 0.f), gy = x, wv = x;
    float g = 0.f;
    if (live) {
        x = *reinterpret_cast<const float4*>(m + j * feat + q * 4);
        gy = *reinterpret_cast<const float4*>(gmsg + j * feat + q * 4);
        wv = *reinterpret_cast<const float4*>(ws + q * 4);
        g = w[j];
    }
    float dot = gy.x * x.x + gy.y * x.y + gy.z * x.z + gy.w * x.w;
    dot += __shfl_xor(dot, 1, 8);
    dot += __shfl_xor(dot, 2, 8);
    dot += __shfl_xor(dot, 4, 8);
    if (!live) return;
    const float gg = dot * g * (1.f - g);
    if (q == 0) ggate[j] = gg;
    *reinterpret_cast<float4*>(gm + j * feat + q * 4) =
        make_float4(gy.x * g + gg * wv.x, gy.y * g + gg * wv.y, gy.z * g + gg * wv.z, gy.w * g + gg * wv.w);
}

}  // namespace i3d

using namespace i3d;

#define LAUNCH_V(kernel, items_rows, feat, ...)                                                             \
    do {                                                                                                    \
        if ((feat) % 4 == 0) {                                                                              \
            long items_ = (long)(items_rows) * ((feat) / 4);                                                \
            hipLaunchKernelGGL(kernel<4>, dim3(cdiv(items_, 256)), dim3(256), 0, (hipStream_t)stream, __VA_ARGS__); \
        } else {                                                                                            \
            long items_ = (long)(items_rows) * (feat);                                                      \
            hipLaunchKernelGGL(kernel<1>, dim3(cdiv(items_, 256)), dim3(256), 0, (hipStream_t)stream, __VA_ARGS__); \
        }                                                                                                   \
    } while (0)

extern "C" int i3d_edge_combine_fwd(const float* P, int ldp, const float* Q, const int* q_code, const float* bias,
                                    const int* src_s, const int* dst_s, int num_edges, int feat, float* pre,
                                    void* stream) {
    I3D_CHECK_ARG(num_edges >= 0 && feat > 0 && ldp >= 2 * feat, "bad shape");
    if (num_edges == 0) return I3D_OK;
    if (feat % 4 == 0 && ldp % 4 == 0) {
        long items = (long)num_edges * (feat / 4);
        hipLaunchKernelGGL(edge_combine_fwd_kernel<4>, dim3(cdiv(items, 256)), dim3(256), 0, (hipStream_t)stream, P, ldp, Q,
                           q_code, bias, src_s, dst_s, num_edges, feat, pre);
    } else {
        long items = (long)num_edges * feat;
        hipLaunchKernelGGL(edge_combine_fwd_kernel<1>, dim3(cdiv(items, 256)), dim3(256), 0, (hipStream_t)stream, P, ldp, Q,
                           q_code, bias, src_s, dst_s, num_edges, feat, pre);
    }
    I3D_CHECK_LAUNCH();
    return I3D_OK;
}

extern "C" int i3d_segment_sum(const float* x, int ldx, const int* ptr, const int* idx, int num_segments, int feat,
                               int scale_mode, float* out, int ldo, void* stream) {
    I3D_CHECK_ARG(num_segments >= 0 && feat > 0 && ldx >= feat && ldo >= feat, "bad shape");
    if (num_segments == 0) return I3D_OK;
    if (feat % 4 == 0 && ldx % 4 == 0 && ldo % 4 == 0 && (((uintptr_t)x | (uintptr_t)out) & 15) == 0) {
        long items = (long)num_segments * (feat / 4);
        hipLaunchKernelGGL(segment_sum_kernel<4>, dim3(cdiv(items, 256)), dim3(256), 0, (hipStream_t)stream, x, ldx, ptr,
                           idx, num_segments, feat, scale_mode, out, ldo);
    } else {
        long items = (long)num_segments * feat;
        hipLaunchKernelGGL(segment_sum_kernel<1>, dim3(cdiv(items, 256)), dim3(256), 0, (hipStream_t)stream, x, ldx, ptr,
                           idx, num_segments, feat, scale_mode, out, ldo);
    }
    I3D_CHECK_LAUNCH();
    return I3D_OK;
}

extern "C" int i3d_segment_sum_bf16(const void* x, int ldx, const int* ptr, const int* idx, int num_segments, int feat,
                                    int scale_mode, float* out, int ldo, void* stream) {
    I3D_CHECK_ARG(num_segments >= 0 && feat > 0 && ldx >= feat && ldo >= feat, "bad shape");
    I3D_CHECK_ARG(feat % 4 == 0 && ldx % 4 == 0 && ldo % 4 == 0 && (((uintptr_t)x & 7) | ((uintptr_t)out & 15)) == 0, "bf16 rows: multiples of 4, aligned");
    if (num_segments == 0) return I3D_OK;
    long items = (long)num_segments * (feat / 4);
    hipLaunchKernelGGL((segment_sum_kernel<4, false, true>), dim3(cdiv(items, 256)), dim3(256), 0, (hipStream_t)stream, (const float*)x,
                       ldx, ptr, idx, num_segments, feat, scale_mode, out, ldo);
    I3D_CHECK_LAUNCH();
    return I3D_OK;
}

extern "C" int i3d_segment_sum_pair(const float* x, int ldx, const int* ptr0, const int* idx0, float* out0, const int* ptr1,
                                    const int* idx1, float* out1, int num_segments, int feat, int ldo, void* stream) {
    I3D_CHECK_ARG(num_segments >= 0 && feat > 0 && ldx >= feat && ldo >= feat, "bad shape");
    if (num_segments == 0) return I3D_OK;
    if (feat % 4 == 0 && ldx % 4 == 0 && ldo % 4 == 0 && (((uintptr_t)x | (uintptr_t)out0 | (uintptr_t)out1) & 15) == 0) {
        long items = 2L * num_segments * (feat / 4);
        hipLaunchKernelGGL((segment_sum_kernel<4, true>), dim3(cdiv(items, 256)), dim3(256), 0, (hipStream_t)stream, x, ldx, ptr0,
                           idx0, num_segments, feat, 0, out0, ldo, ptr1, idx1, out1);
    } else {
        long items = 2L * num_segments * feat;
        hipLaunchKernelGGL((segment_sum_kernel<1, true>), dim3(cdiv(items, 256)), dim3(256), 0, (hipStream_t)stream, x, ldx, ptr0,
                           idx0, num_segments, feat, 0, out0, ldo, ptr1, idx1, out1);
    }
    I3D_CHECK_LAUNCH();
    return I3D_OK;
}

extern "C" int i3d_edge_codes(const int64_t* idx, const int* row_perm, int rows, int n_cols, const int* strides, int v_pad,
                              int* codes, float* onehot, void* stream) {
    I3D_CHECK_ARG(rows >= 0 && n_cols >= 1 && n_cols <= 8 && v_pad > 0 && v_pad % 4 == 0, "bad shape");
    if (rows == 0) return I3D_OK;
    CodeStrides st;
    for (int c = 0; c < 8; ++c) st.s[c] = c < n_cols ? strides[c] : 0;
    long items = (long)rows * (v_pad / 4);
    hipLaunchKernelGGL(edge_codes_kernel, dim3(cdiv(items, 256)), dim3(256), 0, (hipStream_t)stream, (const long*)idx, row_perm,
                       rows, n_cols, st, v_pad, codes, onehot);
    I3D_CHECK_LAUNCH();
    return I3D_OK;
}

extern "C" int i3d_multihot(const int64_t* idx, const int* row_perm, int rows, int n_cols, const int* offsets, int v_pad,
                            float* out, void* stream) {
    I3D_CHECK_ARG(rows >= 0 && n_cols >= 1 && n_cols <= 16 && v_pad > 0 && v_pad % 4 == 0, "bad shape");
    if (rows == 0) return I3D_OK;
    ColOffsets off;
    for (int c = 0; c < 16; ++c) off.o[c] = c < n_cols ? offsets[c] : 0;
    long items = (long)rows * (v_pad / 4);
    hipLaunchKernelGGL(multihot_kernel, dim3(cdiv(items, 256)), dim3(256), 0, (hipStream_t)stream, (const long*)idx, row_perm, rows,
                       n_cols, off, v_pad, out);
    I3D_CHECK_LAUNCH();
    return I3D_OK;
}

extern "C" int i3d_segment_bcast(const float* g, const int* ptr, const int* seg_of_row, int rows, int feat,
                                 int scale_mode, float* out, void* stream) {
    I3D_CHECK_ARG(rows >= 0 && feat > 0, "bad shape");
    if (rows == 0) return I3D_OK;
    LAUNCH_V(segment_bcast_kernel, rows, feat, g, ptr, seg_of_row, rows, feat, scale_mode, out);
    I3D_CHECK_LAUNCH();
    return I3D_OK;
}

extern "C" int i3d_gather_rows(const float* x, const int* idx, int rows, int feat, float* out, void* stream) {
    I3D_CHECK_ARG(rows >= 0 && feat > 0, "bad shape");
    if (rows == 0) return I3D_OK;
    LAUNCH_V(gather_rows_kernel, rows, feat, x, idx, rows, feat, out);
    I3D_CHECK_LAUNCH();
    return I3D_OK;
}

extern "C" int i3d_fourier_encode(const float* d, int num_edges, int n_enc, float* out, void* stream) {
    I3D_CHECK_ARG(num_edges >= 0 && n_enc >= 1 && n_enc <= 16, "bad shape");
    if (num_edges == 0) return I3D_OK;
    hipLaunchKernelGGL(fourier_encode_kernel, dim3(cdiv(num_edges, 256)), dim3(256), 0, (hipStream_t)stream, d, num_edges,
                       n_enc, out);
    I3D_CHECK_LAUNCH();
    return I3D_OK;
}

extern "C" int i3d_soft_edge_fwd(const float* m, const float* ws, const float* bs, int num_edges, int feat, float* msg,
                                 float* w, void* stream) {
    I3D_CHECK_ARG(num_edges >= 0 && feat > 0, "bad shape");
    if (num_edges == 0) return I3D_OK;
    if (feat % 4 == 0 && feat <= 32 && (((uintptr_t)m | (uintptr_t)ws | (uintptr_t)msg) & 15) == 0)
        hipLaunchKernelGGL(soft_edge_fwd_v8_kernel, dim3(cdiv((long)num_edges * 8, 256)), dim3(256), 0, (hipStream_t)stream, m, ws,
                           bs, num_edges, feat, msg, w);
    else
        hipLaunchKernelGGL(soft_edge_fwd_kernel, dim3(cdiv(num_edges, 256)), dim3(256), 0, (hipStream_t)stream, m, ws, bs,
                           num_edges, feat, msg, w);
    I3D_CHECK_LAUNCH();
    return I3D_OK;
}

extern "C" int i3d_soft_edge_bwd(const float* grad_msg, const float* m, const float* w, const float* ws, int num_edges,
                                 int feat, float* grad_m, float* g_gate, void* stream) {
    I3D_CHECK_ARG(num_edges >= 0 && feat > 0, "bad shape");
    if (num_edges == 0) return I3D_OK;
    if (feat % 4 == 0 && feat <= 32 && (((uintptr_t)m | (uintptr_t)ws | (uintptr_t)grad_msg | (uintptr_t)grad_m) & 15) == 0)
        hipLaunchKernelGGL(soft_edge_bwd_v8_kernel, dim3(cdiv((long)num_edges * 8, 256)), dim3(256), 0, (hipStream_t)stream,
                           grad_msg, m, w, ws, num_edges, feat, grad_m, g_gate);
    else
        hipLaunchKernelGGL(soft_edge_bwd_kernel, dim3(cdiv(num_edges, 256)), dim3(256), 0, (hipStream_t)stream, grad_msg, m, w,
                           ws, num_edges, feat, grad_m, g_gate);
    I3D_CHECK_LAUNCH();
    return I3D_OK;
}
