// NT-Xent contrastive loss (single positive and multiple-positives variants), forward and backward.
//
// Replaces NTXent.forward (reference commons/losses.py:143-155) and NTXentMultiplePositives.forward
// (reference commons/losses.py:225-247):
//     S = z1 z2^T;  S' = S / (|z1_i||z2_j| + eps);  P = exp(S'/tau);  (multi: P summed over conformers)
//     loss = - mean_i log( pos_i / (rowsum_i - pos_i) )
// The similarity GEMM runs on the MFMA GEMM (i3d_gemm_f32); these kernels fuse normalisation, exp, the row
// reductions and the log, and produce dL/dS plus the rank-1 norm-path terms for the backward GEMMs.
// Row-sharded for data parallelism: z1 are the LOCAL rows, z2 the all-gathered batch, pos_offset = rank*b1.
#include "common.h"

namespace i3d {

__device__ __forceinline__ float block_sum(float v, float* sm) {
    // 256 threads = 4 waves
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) sm[w] = v;
    __syncthreads();
    return sm[0] + sm[1] + sm[2] + sm[3];
}

__global__ void __launch_bounds__(256)
row_norms_kernel(const float* __restrict__ z, int rows, int dim, float* __restrict__ norms) {
    I3D_CHAIN_PRIO();
    __shared__ float sm[4];
    int r = blockIdx.x;
    float acc = 0.f;
    for (int c = threadIdx.x; c < dim; c += 256) {
        float v = z[(long)r * dim + c];
        acc += v * v;
    }
    float s = block_sum(acc, sm);
    if (threadIdx.x == 0) norms[r] = sqrtf(s);
}

__global__ void __launch_bounds__(256)
ntxent_fwd_kernel(const float* __restrict__ sim, const float* __restrict__ n1, const float* __restrict__ n2, int b1,
                  int ncol, int conf, int pos_offset, float inv_tau, float eps, float* __restrict__ row_sum,
                  float* __restrict__ row_pos) {
    I3D_CHAIN_PRIO();
    __shared__ float sm[4];
    int i = blockIdx.x;
    float a = n1[i];
    int p0 = (pos_offset + i) * conf, p1 = p0 + conf;
    float rs = 0.f, ps = 0.f;
    for (int j = threadIdx.x; j < ncol; j += 256) {
        float s = sim[(long)i * ncol + j] / (a * n2[j] + eps);
        float p = expf(s * inv_tau);
        rs += p;
        if (j >= p0 && j < p1) ps += p;
    }
    rs = block_sum(rs, sm);
    ps = block_sum(ps, sm);
    if (threadIdx.x == 0) {
        row_sum[i] = rs;
        row_pos[i] = ps;
    }
}

__global__ void __launch_bounds__(256)
ntxent_loss_kernel(const float* __restrict__ row_sum, const float* __restrict__ row_pos, int b1, float scale,
                   float* __restrict__ loss_sum) {
    I3D_CHAIN_PRIO();
    __shared__ float sm[4];
    float acc = 0.f;
    for (int i = threadIdx.x; i < b1; i += 256) acc += -logf(row_pos[i] / (row_sum[i] - row_pos[i]));
    float s = block_sum(acc, sm);
    if (threadIdx.x == 0) loss_sum[0] = s * scale;
}

// dL/dP_ij = gs * ( j positive ? -1/pos_i : 1/(rowsum_i - pos_i) ),  G = dL/dP * P / tau,  H = G / (a_i b_j + eps)
__global__ void __launch_bounds__(256)
ntxent_bwd_row_kernel(const float* __restrict__ sim, const float* __restrict__ n1, const float* __restrict__ n2,
                      const float* __restrict__ row_sum, const float* __restrict__ row_pos, int b1, int ncol, int conf,
                      int pos_offset, float inv_tau, float eps, float gs, const float* __restrict__ gs_dev,
                      float* __restrict__ dsim, float* __restrict__ ca) {
    I3D_CHAIN_PRIO();
    __shared__ float sm[4];
    int i = blockIdx.x;
    if (gs_dev != nullptr) gs *= gs_dev[0];      // the upstream scalar gradient stays on the device
    float a = n1[i];
    int p0 = (pos_offset + i) * conf, p1 = p0 + conf;
    float pos = row_pos[i], den = row_sum[i] - pos;
    float g_neg = gs / den, g_pos = -gs / pos;
    float da = 0.f;
    for (int j = threadIdx.x; j < ncol; j += 256) {
        float b = n2[j];
        float nrm = a * b + eps;
        float s = sim[(long)i * ncol + j] / nrm;
        float p = expf(s * inv_tau);
        float G = ((j >= p0 && j < p1) ? g_pos : g_neg) * p * inv_tau;
        float H = G / nrm;
        dsim[(long)i * ncol + j] = H;
        da -= H * s * b;
    }
    da = block_sum(da, sm);
    if (threadIdx.x == 0) ca[i] = a > 0.f ? da / a : 0.f;
}

// cb_j = -(1/b_j) sum_i H_ij * S'_ij * a_i ;  16 columns per block, 16 row lanes (B = 512: 32 workgroups, 32 rows per
// lane; with 64 columns x 4 lanes it was 8 workgroups walking 128 rows each: 39 us)
constexpr int COL_W = 16, COL_L = 16;
__global__ void __launch_bounds__(256)
ntxent_bwd_col_kernel(const float* __restrict__ sim, const float* __restrict__ dsim, const float* __restrict__ n1,
                      const float* __restrict__ n2, int b1, int ncol, float eps, float* __restrict__ cb) {
    I3D_CHAIN_PRIO();
    __shared__ float sm[COL_L][COL_W];
    int cx = threadIdx.x % COL_W, ry = threadIdx.x / COL_W;
    int j = blockIdx.x * COL_W + cx;
    float acc = 0.f;
    if (j < ncol) {
        float b = n2[j];
        for (int i = ry; i < b1; i += COL_L) {
            float a = n1[i];
            float s = sim[(long)i * ncol + j] / (a * b + eps);
            acc -= dsim[(long)i * ncol + j] * s * a;
        }
    }
    sm[ry][cx] = acc;
    __syncthreads();
    if (ry == 0 && j < ncol) {
        float b = n2[j];
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < COL_L; ++k) t += sm[k][cx];
        cb[j] = b > 0.f ? t / b : 0.f;
    }
}

template <int V>
__global__ void __launch_bounds__(256)
row_axpy_kernel(const float* __restrict__ z, const float* __restrict__ coef, int rows, int dim, float* __restrict__ out) {
    I3D_CHAIN_PRIO();
    const int DV = dim / V;
    long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long)rows * DV) return;
    int r = (int)(t / DV);
    float c = coef[r];
    long off = t * V;
    if (V == 4) {
        float4 a = *reinterpret_cast<const float4*>(z + off);
        float4 o = *reinterpret_cast<const float4*>(out + off);
        o.x += c * a.x; o.y += c * a.y; o.z += c * a.z; o.w += c * a.w;
        *reinterpret_cast<float4*>(out + off) = o;
    } else {
        out[off] += c * z[off];
    }
}

template <int V>
__global__ void __launch_bounds__(256)
row_scale_kernel(const float* __restrict__ z, const float* __restrict__ coef, int rows, int dim, float* __restrict__ out) {
    I3D_CHAIN_PRIO();
    const int DV = dim / V;
    long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long)rows * DV) return;
    float c = coef[(int)(t / DV)];
    long off = t * V;
    if (V == 4) {
        float4 a = *reinterpret_cast<const float4*>(z + off);
        *reinterpret_cast<float4*>(out + off) = make_float4(c * a.x, c * a.y, c * a.z, c * a.w);
    } else {
        out[off] = c * z[off];
    }
}


// ---- fewer launches for i3d_ntxent_loss_fwd / _bwd: the step's smallest kernels are a chain of launch latencies ----
// Both row-norm vectors from one launch, one wave per row (no LDS, no barrier).
__global__ void __launch_bounds__(256)
row_norms_pair_kernel(const float* __restrict__ z1, int rows1, const float* __restrict__ z2, int rows2, int dim,
                      float* __restrict__ n1, float* __restrict__ n2) {
    I3D_CHAIN_PRIO();
    const int lane = threadIdx.x & 63;
    int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows1 + rows2) return;
    const float* z = r < rows1 ? z1 + (long)r * dim : z2 + (long)(r - rows1) * dim;
    float acc = 0.f;
    if ((dim & 3) == 0) {
        for (int c = lane * 4; c < dim; c += 256) {
            float4 v = *reinterpret_cast<const float4*>(z + c);
            acc += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
        }
    } else {
        for (int c = lane; c < dim; c += 64) acc += z[c] * z[c];
    }
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
    if (lane == 0) {
        if (r < rows1) n1[r] = sqrtf(acc);
        else n2[r - rows1] = sqrtf(acc);
    }
}

// ntxent_bwd_row_kernel and ntxent_bwd_col_kernel as ONE launch, and the norm-path terms of the two gradients written by
// it: workgroups [0, b1) are the rows (dsim row i, ca_i, dz1_i = ca_i z1_i), the rest FC_W columns each (cb_j from sim and the
// row sums - the same H_ij, recomputed instead of read back from the rows' workgroups -, dz2_j = cb_j z2_j); the two GEMMs
// behind it accumulate dS z2 and dS^T z1 on top (no row_axpy launches).  The column workgroups walk the rows FC_U at a time
// with all loads of a group in flight together (one L2 round trip per group, not per row: the first version's column kernel
// was 14 us of serial load latency).
constexpr int FC_W = 8, FC_L = 32, FC_U = 8;
__global__ void __launch_bounds__(256)
ntxent_bwd_fused_kernel(const float* __restrict__ sim, const float* __restrict__ n1, const float* __restrict__ n2,
                        const float* __restrict__ row_sum, const float* __restrict__ row_pos, const float* __restrict__ z1,
                        const float* __restrict__ z2, int b1, int ncol, int conf, int dim, int pos_offset, float inv_tau,
                        float eps, float gs, const float* __restrict__ gs_dev, float* __restrict__ dsim,
                        float* __restrict__ dz1, float* __restrict__ dz2) {
    I3D_CHAIN_PRIO();
    __shared__ float sm[FC_L][FC_W];
    __shared__ float coef[FC_W];
    if (gs_dev != nullptr) gs *= gs_dev[0];      // the upstream scalar gradient stays on the device
    if ((int)blockIdx.x < b1) {
        const int i = blockIdx.x;
        const float zv = (int)threadIdx.x < dim ? z1[(long)i * dim + threadIdx.x] : 0.f;     // in flight under the row pass
        float a = n1[i];
        int p0 = (pos_offset + i) * conf, p1 = p0 + conf;
        float pos = row_pos[i], den = row_sum[i] - pos;
        float g_neg = gs / den, g_pos = -gs / pos;
        float da = 0.f;
        for (int j = threadIdx.x; j < ncol; j += 256) {
            float b = n2[j];
            float nrm = a * b + eps;
            float s = sim[(long)i * ncol + j] / nrm;
            float p = expf(s * inv_tau);
            float G = ((j >= p0 && j < p1) ? g_pos : g_neg) * p * inv_tau;
            float H = G / nrm;
            dsim[(long)i * ncol + j] = H;
            da -= H * s * b;
        }
        da = block_sum(da, &sm[0][0]);
        const float ca = a > 0.f ? da / a : 0.f;
        if ((int)threadIdx.x < dim) dz1[(long)i * dim + threadIdx.x] = ca * zv;
        for (int c = threadIdx.x + 256; c < dim; c += 256) dz1[(long)i * dim + c] = ca * z1[(long)i * dim + c];
        return;
    }
    const int j0 = ((int)blockIdx.x - b1) * FC_W;
    const int cx = threadIdx.x % FC_W, ry = threadIdx.x / FC_W;
    const int j = j0 + cx;
    const bool col_ok = j < ncol;
    const float b = col_ok ? n2[j] : 1.f;
    float acc = 0.f;
    for (int i0 = ry; i0 < b1; i0 += FC_L * FC_U) {
        float sv[FC_U], av[FC_U], pv[FC_U], rv[FC_U];
#pragma unroll
        for (int u = 0; u < FC_U; ++u) {
            const int i = i0 + u * FC_L;
            const bool ok = col_ok && i < b1;
            sv[u] = ok ? sim[(long)i * ncol + j] : 0.f;
            av[u] = ok ? n1[i] : 0.f;
            pv[u] = ok ? row_pos[i] : 1.f;
            rv[u] = ok ? row_sum[i] : 3.f;
        }
#pragma unroll
        for (int u = 0; u < FC_U; ++u) {
            const int i = i0 + u * FC_L;
            const int p0 = (pos_offset + i) * conf;
            const float a = av[u], pos = pv[u], den = rv[u] - pos;
            const float nrm = a * b + eps;
            const float s = sv[u] / nrm;
            const float p = expf(s * inv_tau);
            const float G = ((j >= p0 && j < p0 + conf) ? -gs / pos : gs / den) * p * inv_tau;
            if (col_ok && i < b1) acc -= (G / nrm) * s * a;
        }
    }
    sm[ry][cx] = acc;
    __syncthreads();
    if (ry == 0) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < FC_L; ++k) t += sm[k][cx];
        coef[cx] = (col_ok && b > 0.f) ? t / b : 0.f;
    }
    __syncthreads();
    const int items = min(FC_W, ncol - j0) * dim;
    for (int t = threadIdx.x; t < items; t += 256) dz2[(long)j0 * dim + t] = coef[t / dim] * z2[(long)j0 * dim + t];
}

}  // namespace i3d

using namespace i3d;

extern "C" int i3d_row_norms(const float* z, int rows, int dim, float* norms, void* stream) {
    I3D_CHECK_ARG(rows >= 0 && dim > 0, "bad shape");
    if (rows == 0) return I3D_OK;
    hipLaunchKernelGGL(row_norms_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, z, rows, dim, norms);
    I3D_CHECK_LAUNCH();
    return I3D_OK;
}

extern "C" int i3d_ntxent_fwd(const float* sim, const float* n1, const float* n2, int b1, int b2, int conf,
                              int pos_offset, float tau, float eps, float loss_scale, float* row_sum, float* row_pos,
                              float* loss_sum, void* stream) {
    I3D_CHECK_ARG(b1 > 0 && b2 > 0 && conf > 0 && tau > 0.f, "bad shape");
    I3D_CHECK_ARG(pos_offset >= 0 && pos_offset + b1 <= b2, "positive columns out of range");
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(ntxent_fwd_kernel, dim3(b1), dim3(256), 0, s, sim, n1, n2, b1, b2 * conf, conf, pos_offset,
                       1.f / tau, eps, row_sum, row_pos);
    I3D_CHECK_LAUNCH();
    hipLaunchKernelGGL(ntxent_loss_kernel, dim3(1), dim3(256), 0, s, row_sum, row_pos, b1, loss_scale, loss_sum);
    I3D_CHECK_LAUNCH();
    return I3D_OK;
}

extern "C" int i3d_ntxent_bwd(const float* sim, const float* n1, const float* n2, const float* row_sum,
                              const float* row_pos, int b1, int b2, int conf, int pos_offset, float tau, float eps,
                              float grad_scale, const float* grad_scale_dev, float* dsim, float* ca, float* cb,
                              void* stream) {
    I3D_CHECK_ARG(b1 > 0 && b2 > 0 && conf > 0 && tau > 0.f, "bad shape");
    hipStream_t s = (hipStream_t)stream;
    int ncol = b2 * conf;
    hipLaunchKernelGGL(ntxent_bwd_row_kernel, dim3(b1), dim3(256), 0, s, sim, n1, n2, row_sum, row_pos, b1, ncol, conf,
                       pos_offset, 1.f / tau, eps, grad_scale, grad_scale_dev, dsim, ca);
    I3D_CHECK_LAUNCH();
    hipLaunchKernelGGL(ntxent_bwd_col_kernel, dim3(cdiv(ncol, COL_W)), dim3(256), 0, s, sim, dsim, n1, n2, b1, ncol, eps, cb);
    I3D_CHECK_LAUNCH();
    return I3D_OK;
}

extern "C" int i3d_row_axpy(const float* z, const float* coef, int rows, int dim, float* out, void* stream) {
    I3D_CHECK_ARG(rows >= 0 && dim > 0, "bad shape");
    if (rows == 0) return I3D_OK;
    if (dim % 4 == 0) {
        long items = (long)rows * dim / 4;
        hipLaunchKernelGGL(row_axpy_kernel<4>, dim3(cdiv(items, 256)), dim3(256), 0, (hipStream_t)stream, z, coef, rows, dim,
                           out);
    } else {
        long items = (long)rows * dim;
        hipLaunchKernelGGL(row_axpy_kernel<1>, dim3(cdiv(items, 256)), dim3(256), 0, (hipStream_t)stream, z, coef, rows, dim,
                           out);
    }
    I3D_CHECK_LAUNCH();
    return I3D_OK;
}

extern "C" int i3d_row_scale(const float* z, const float* coef, int rows, int dim, float* out, void* stream) {
    I3D_CHECK_ARG(rows >= 0 && dim > 0, "bad shape");
    if (rows == 0) return I3D_OK;
    if (dim % 4 == 0) {
        long items = (long)rows * dim / 4;
        hipLaunchKernelGGL(row_scale_kernel<4>, dim3(cdiv(items, 256)), dim3(256), 0, (hipStream_t)stream, z, coef, rows, dim,
                           out);
    } else {
        long items = (long)rows * dim;
        hipLaunchKernelGGL(row_scale_kernel<1>, dim3(cdiv(items, 256)), dim3(256), 0, (hipStream_t)stream, z, coef, rows, dim,
                           out);
    }
    I3D_CHECK_LAUNCH();
    return I3D_OK;
}

// ---- the whole loss from one C call per direction (host sequencing only: the kernels above + the similarity GEMMs) ----
// scratch layout (floats): n1[b1] | n2[b2c] | row_sum[b1] | row_pos[b1] | sim[b1, b2c]  (kept for the backward pass)
extern "C" long i3d_ntxent_loss_scratch_floats(int b1, int b2c) {
    auto al = [](long n) { return (n + 3) & ~3L; };
    return al(b1) + al(b2c) + 2 * al(b1) + al((long)b1 * b2c);
}

// the loss in four + three launches (norms, GEMM, rows, sum; rows and columns, GEMM, GEMM).  I3D_LOSS_FUSED=0: the five + six of
// the first version (row norms x2, GEMM, rows, sum; rows, columns, GEMM, axpy, GEMM, axpy)
static bool loss_fused() {
    static const bool on = [] { const char* e = getenv("I3D_LOSS_FUSED"); return e == nullptr || e[0] != '0'; }();
    return on;
}

extern "C" int i3d_ntxent_loss_fwd(const float* z1, const float* z2, int b1, int b2, int conf, int dim, int pos_offset, float tau,
                                   float eps, float loss_scale, float* scratch, float* loss, void* stream) {
    I3D_CHECK_ARG(z1 != nullptr && z2 != nullptr && scratch != nullptr && loss != nullptr && b1 > 0 && b2 > 0 && conf > 0, "bad arguments");
    auto al = [](long n) { return (n + 3) & ~3L; };
    const int b2c = b2 * conf;
    float* n1 = scratch;
    float* n2 = n1 + al(b1);
    float* row_sum = n2 + al(b2c);
    float* row_pos = row_sum + al(b1);
    float* sim = row_pos + al(b1);
    int rc;
    if (loss_fused()) {
        I3D_CHECK_ARG(dim > 0 && tau > 0.f, "bad shape");
        I3D_CHECK_ARG(pos_offset >= 0 && pos_offset + b1 <= b2, "positive columns out of range");
        hipStream_t s = (hipStream_t)stream;
        hipLaunchKernelGGL(row_norms_pair_kernel, dim3(cdiv(b1 + b2c, 4)), dim3(256), 0, s, z1, b1, z2, b2c, dim, n1, n2);
        I3D_CHECK_LAUNCH();
        if ((rc = i3d_gemm_f32(0, 1, b1, b2c, dim, z1, dim, z2, dim, sim, b2c, nullptr, 0, stream)) != I3D_OK) return rc;
        // the rows' kernel and the one-workgroup sum of their terms stay two launches: a last-arriving-workgroup sum in the
        // rows' kernel needs a device-scope release per workgroup (an L2 write-back on the 8-XCD part): 14.9 us against 7.1 + 5.7
        hipLaunchKernelGGL(ntxent_fwd_kernel, dim3(b1), dim3(256), 0, s, sim, n1, n2, b1, b2c, conf, pos_offset, 1.f / tau, eps, row_sum,
                           row_pos);
        I3D_CHECK_LAUNCH();
        hipLaunchKernelGGL(ntxent_loss_kernel, dim3(1), dim3(256), 0, s, row_sum, row_pos, b1, loss_scale, loss);
        I3D_CHECK_LAUNCH();
        return I3D_OK;
    }
    if ((rc = i3d_row_norms(z1, b1, dim, n1, stream)) != I3D_OK) return rc;
    if ((rc = i3d_row_norms(z2, b2c, dim, n2, stream)) != I3D_OK) return rc;
    if ((rc = i3d_gemm_f32(0, 1, b1, b2c, dim, z1, dim, z2, dim, sim, b2c, nullptr, 0, stream)) != I3D_OK) return rc;
    return i3d_ntxent_fwd(sim, n1, n2, b1, b2, conf, pos_offset, tau, eps, loss_scale, row_sum, row_pos, loss, stream);
}

// dz1 [b1, dim], dz2 [b2c, dim]; work: b1*b2c + b1 + b2c floats (dsim, ca, cb); grad_scale_dev: the upstream scalar gradient
extern "C" int i3d_ntxent_loss_bwd(const float* z1, const float* z2, int b1, int b2, int conf, int dim, int pos_offset, float tau,
                                   float eps, float loss_scale, const float* scratch, const float* grad_scale_dev, float* work,
                                   float* dz1, float* dz2, void* stream) {
    I3D_CHECK_ARG(z1 != nullptr && z2 != nullptr && scratch != nullptr && work != nullptr && dz1 != nullptr && dz2 != nullptr, "null");
    auto al = [](long n) { return (n + 3) & ~3L; };
    const int b2c = b2 * conf;
    const float* n1 = scratch;
    const float* n2 = n1 + al(b1);
    const float* row_sum = n2 + al(b2c);
    const float* row_pos = row_sum + al(b1);
    const float* sim = row_pos + al(b1);
    float* dsim = work;
    float* ca = dsim + al((long)b1 * b2c);
    float* cb = ca + al(b1);
    int rc;
    if (loss_fused()) {
        I3D_CHECK_ARG(b1 > 0 && b2 > 0 && conf > 0 && dim > 0 && tau > 0.f, "bad shape");
        hipLaunchKernelGGL(ntxent_bwd_fused_kernel, dim3(b1 + cdiv(b2c, FC_W)), dim3(256), 0, (hipStream_t)stream, sim, n1, n2, row_sum,
                           row_pos, z1, z2, b1, b2c, conf, dim, pos_offset, 1.f / tau, eps, loss_scale, grad_scale_dev, dsim, dz1, dz2);
        I3D_CHECK_LAUNCH();
        if ((rc = i3d_gemm_f32(0, 0, b1, dim, b2c, dsim, b2c, z2, dim, dz1, dim, nullptr, 1, stream)) != I3D_OK) return rc;
        return i3d_gemm_f32(1, 0, b2c, dim, b1, dsim, b2c, z1, dim, dz2, dim, nullptr, 1, stream);
    }
    if ((rc = i3d_ntxent_bwd(sim, n1, n2, row_sum, row_pos, b1, b2, conf, pos_offset, tau, eps, loss_scale, grad_scale_dev, dsim, ca,
                             cb, stream)) != I3D_OK) return rc;
    if ((rc = i3d_gemm_f32(0, 0, b1, dim, b2c, dsim, b2c, z2, dim, dz1, dim, nullptr, 0, stream)) != I3D_OK) return rc;
    if ((rc = i3d_row_axpy(z1, ca, b1, dim, dz1, stream)) != I3D_OK) return rc;
    if ((rc = i3d_gemm_f32(1, 0, b2c, dim, b1, dsim, b2c, z1, dim, dz2, dim, nullptr, 0, stream)) != I3D_OK) return rc;
    return i3d_row_axpy(z2, cb, b2c, dim, dz2, stream);
}
