// Contrastive monitoring metrics of the pre-training loop in one device pass.
//
// Replaces the nine metric modules of configs_clean/pre-train_QM9.yml:15-24 - reference trainer/metrics.py:161-174
// (DimensionCovariance, BatchVariance), :212-333 (Alignment, Uniformity, TruePositiveRate, TrueNegativeRate,
// ContrastiveAccuracy, PositiveSimilarity), :443-463 (NegativeSimilarity) with commons/losses.py:946-959 - which the
// trainer evaluates every `log_iterations` (= 2) steps, each through its own chain of small ops and its own `.item()`
// (trainer/self_supervised_trainer.py:31-50).  Here: five small GEMMs (similarity, two Gram matrices, two second-moment
// matrices; gemm.hip), the two row kernels below, deterministic column sums (bn.hip) and ONE device-to-host copy.
#include "common.h"

namespace i3d {

constexpr int RS = 8;     // statistics per row of contrastive_rowstats_kernel

__device__ __forceinline__ float block_sum(float v, float* sm) {
    // fixed-order tree over the 256 threads of the block
    sm[threadIdx.x] = v;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) sm[threadIdx.x] += sm[threadIdx.x + s];
        __syncthreads();
    }
    float r = sm[0];
    __syncthreads();
    return r;
}

// row i (one workgroup): S = x1 x2^T [B1, lds] (x2: first B1 rows), G1 = x1 x1^T [B1, B1], G2 = x2 x2^T [B2, B2]
//   out[i][0] = sum_j cos_ij   out[i][1] = cos_ii   out[i][2] = [(cos_ii + 1)/2 > thr]   out[i][3] = #{j != i : (cos_ij+1)/2 <= thr}
//   out[i][4] = |x1_i - x2_i|^alpha   out[i][5] = sum_{j>i} exp(-t |x1_i - x1_j|^2)   out[i][6] = the same for x2 (i < B2)
__global__ void __launch_bounds__(256)
contrastive_rowstats_kernel(const float* __restrict__ S, const float* __restrict__ G1, const float* __restrict__ G2, int B1,
                            int B2, float thr, float t, float alpha, float* __restrict__ out) {
    I3D_CHAIN_PRIO();
    __shared__ float sm[256];
    const int i = blockIdx.x;
    float s_cos = 0.f, s_tn = 0.f, s_u1 = 0.f, s_u2 = 0.f;
    if (i < B1) {
        const float n1 = sqrtf(G1[(long)i * B1 + i]);
        const float g1ii = G1[(long)i * B1 + i];
        for (int j = threadIdx.x; j < B1; j += 256) {
            const float c = S[(long)i * B1 + j] / (n1 * sqrtf(G2[(long)j * B2 + j]));
            s_cos += c;
            if (j != i && !((c + 1.f) * 0.5f > thr)) s_tn += 1.f;
            if (j > i) s_u1 += expf(-t * fmaxf(g1ii + G1[(long)j * B1 + j] - 2.f * G1[(long)i * B1 + j], 0.f));
        }
    }
    if (i < B2) {
        const float g2ii = G2[(long)i * B2 + i];
        for (int j = i + 1 + threadIdx.x; j < B2; j += 256)
            s_u2 += expf(-t * fmaxf(g2ii + G2[(long)j * B2 + j] - 2.f * G2[(long)i * B2 + j], 0.f));
    }
    s_cos = block_sum(s_cos, sm);
    s_tn = block_sum(s_tn, sm);
    s_u1 = block_sum(s_u1, sm);
    s_u2 = block_sum(s_u2, sm);
    if (threadIdx.x == 0) {
        float* o = out + (long)i * RS;
        float cii = 0.f, tp = 0.f, al = 0.f;
        if (i < B1) {
            const float g1 = G1[(long)i * B1 + i], g2 = G2[(long)i * B2 + i], sii = S[(long)i * B1 + i];
            cii = sii / (sqrtf(g1) * sqrtf(g2));
            tp = ((cii + 1.f) * 0.5f > thr) ? 1.f : 0.f;
            al = powf(sqrtf(fmaxf(g1 + g2 - 2.f * sii, 0.f)), alpha);
        }
        o[0] = s_cos; o[1] = cii; o[2] = tp; o[3] = s_tn; o[4] = al; o[5] = s_u1; o[6] = s_u2; o[7] = 0.f;
    }
}

// row a of the second-moment matrix C = X^T X [D, D] with the column sums s = sum_rows X:
//   out[a][0] = sum_{b != a} ((C_ab - s_a s_b / n) / (n - 1))^2      (squared off-diagonal covariances, cov_loss)
//   out[a][1] = C_aa                                                 (sum of squares of column a)
__global__ void __launch_bounds__(256)
cov_rowstats_kernel(const float* __restrict__ C, const float* __restrict__ colsum, int n, int D, float* __restrict__ out) {
    I3D_CHAIN_PRIO();
    __shared__ float sm[256];
    const int a = blockIdx.x;
    const float sa = colsum[a], inv_n = 1.f / (float)n, inv_n1 = 1.f / (float)(n > 1 ? n - 1 : 1);
    float acc = 0.f;
    for (int b = threadIdx.x; b < D; b += 256) {
        if (b == a) continue;
        const float cov = (C[(long)a * D + b] - sa * colsum[b] * inv_n) * inv_n1;
        acc += cov * cov;
    }
    acc = block_sum(acc, sm);
    if (threadIdx.x == 0) {
        out[(long)a * 2] = acc;
        out[(long)a * 2 + 1] = C[(long)a * D + a];
    }
}

}  // namespace i3d

using namespace i3d;

extern "C" int i3d_contrastive_rowstats(const float* S, const float* G1, const float* G2, int B1, int B2, float threshold,
                                        float t, float alpha, float* out, void* stream) {
    I3D_CHECK_ARG(B1 > 0 && B2 >= B1, "need 0 < B1 <= B2");
    hipLaunchKernelGGL(contrastive_rowstats_kernel, dim3(B2), dim3(256), 0, (hipStream_t)stream, S, G1, G2, B1, B2, threshold, t,
                       alpha, out);
    I3D_CHECK_LAUNCH();
    return I3D_OK;
}

extern "C" int i3d_cov_rowstats(const float* C, const float* colsum, int n, int D, float* out, void* stream) {
    I3D_CHECK_ARG(n > 0 && D > 0, "bad shape");
    hipLaunchKernelGGL(cov_rowstats_kernel, dim3(D), dim3(256), 0, (hipStream_t)stream, C, colsum, n, D, out);
    I3D_CHECK_LAUNCH();
    return I3D_OK;
}
