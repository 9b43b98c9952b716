// Dense tower GEMMs on the fp32 matrix cores of gfx950.
//
// Replaces aten::addmm / aten::mm behind nn.Linear in FCLayer (reference models/base_layers.py:101)
// for the forward (Y = X W^T + b), the data gradient (dX = dY W) and the weight gradient
// (dW = dY^T X).  Exact fp32: v_mfma_f32_16x16x4_f32 is bit-for-bit an fp32 fmaf chain
// (MI355X guide 3), so parity with the reference's CPU fp32 matmul is a summation-order question only.
//
// Structure (wave64, 4 waves / workgroup):
//  * both operand tiles are staged in LDS k-major  T[k][idx]  (idx = m or n), so the MFMA fragment of
//    lane l - T[4*kk + (l>>4)][idx0 + (l&15)] - is a conflict-free ds_read_b32: the two 16-lane halves of a
//    32-lane group are steered to different bank halves by XOR-ing bit 4 of idx with ((k ^ (k>>2)) & 1).
//  * register-staged double buffering: global loads of K-tile t+1 are issued before the MFMAs of tile t
//    and written to the other LDS buffer afterwards: one barrier per K-tile.
//  * the MFMA is issued with (W-fragment, X-fragment) so the accumulator holds C^T tiles: a lane owns 4
//    consecutive output columns of one row -> the epilogue (bias, accumulate, split-K atomics) is one
//    16-byte access per tile.
//  * k-contiguous operands (X[m][k], W[n][k]) are loaded with 16-byte loads along k and transposed on the
//    LDS write; idx-contiguous operands (dY^T, W for dX) are loaded along idx and written as b128.
//  * split-K (grid.z) with fp32 atomics for the weight gradients, whose reduction axis is the row count.
#include "common.h"

namespace i3d {

typedef float floatx4 __attribute__((ext_vector_type(4)));

struct GemmArgs {
    const float* A;
    const float* B;
    float* C;
    const float* bias;
    int M, N, K;
    int lda, ldb, ldc;
    int a_kcontig;   // 1: A[m*lda + k]   0: A[k*lda + m]
    int b_kcontig;   // 1: B[n*ldb + k]   0: B[k*ldb + n]
    int accumulate;  // C += ...
    int k_per_split; // multiple of BK
    int atomic_out;  // split-K: atomicAdd into C
    int a_vec, b_vec, c_vec;  // 16-byte access allowed (pointer + leading dimension aligned)
};

constexpr int BK = 16;

__device__ __forceinline__ int swz(int k, int idx) { return idx ^ ((((k) ^ (k >> 2)) & 1) << 4); }

template <int R, int LD>
struct TileStage {
    static constexpr int SLOTS = R * BK / 4;                 // float4 slots in a tile
    static constexpr int PER_THREAD = (SLOTS + 255) / 256;
    float4 v[PER_THREAD];

    // kcontig: slot -> (idx = s / 4, kq = s % 4), 4 consecutive k of one row
    // else   : slot -> (k = s / (R/4), iq = s % (R/4)), 4 consecutive idx of one k
    __device__ __forceinline__ void load(const float* __restrict__ p, int ld, int kcontig, int vec, int idx0,
                                         int idx_max, int k0, int k_end) {
#pragma unroll
        for (int it = 0; it < PER_THREAD; ++it) {
            int s = threadIdx.x + it * 256;
            float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
            if (s < SLOTS) {
                if (kcontig) {
                    int idx = idx0 + (s >> 2), k = k0 + (s & 3) * 4;
                    if (idx < idx_max && k < k_end) {
                        const float* q = p + (long)idx * ld + k;
                        if (vec && k + 3 < k_end) {
                            r = *reinterpret_cast<const float4*>(q);
                        } else {
                            r.x = q[0];
                            if (k + 1 < k_end) r.y = q[1];
                            if (k + 2 < k_end) r.z = q[2];
                            if (k + 3 < k_end) r.w = q[3];
                        }
                    }
                } else {
                    int k = k0 + s / (R / 4), idx = idx0 + (s % (R / 4)) * 4;
                    if (k < k_end && idx < idx_max) {
                        const float* q = p + (long)k * ld + idx;
                        if (vec && idx + 3 < idx_max) {
                            r = *reinterpret_cast<const float4*>(q);
                        } else {
                            r.x = q[0];
                            if (idx + 1 < idx_max) r.y = q[1];
                            if (idx + 2 < idx_max) r.z = q[2];
                            if (idx + 3 < idx_max) r.w = q[3];
                        }
                    }
                }
            }
            v[it] = r;
        }
    }

    __device__ __forceinline__ void store(float* __restrict__ T, int kcontig) const {
#pragma unroll
        for (int it = 0; it < PER_THREAD; ++it) {
            int s = threadIdx.x + it * 256;
            if (s < SLOTS) {
                if (kcontig) {
                    int idx = s >> 2, k = (s & 3) * 4;
                    T[(k + 0) * LD + swz(k + 0, idx)] = v[it].x;
                    T[(k + 1) * LD + swz(k + 1, idx)] = v[it].y;
                    T[(k + 2) * LD + swz(k + 2, idx)] = v[it].z;
                    T[(k + 3) * LD + swz(k + 3, idx)] = v[it].w;
                } else {
                    int k = s / (R / 4), idx = (s % (R / 4)) * 4;
                    *reinterpret_cast<float4*>(&T[k * LD + swz(k, idx)]) = v[it];
                }
            }
        }
    }
};

template <int WAVES_M, int WAVES_N, int WM_T, int WN_T>
__global__ void __launch_bounds__(256)
gemm_f32_kernel(GemmArgs g) {
    constexpr int BM = WAVES_M * WM_T * 16, BN = WAVES_N * WN_T * 16;
    constexpr int LDA = (BM + 31) / 32 * 32, LDB = (BN + 31) / 32 * 32;
    __shared__ __attribute__((aligned(16))) float As[2][BK * LDA];
    __shared__ __attribute__((aligned(16))) float Bs[2][BK * LDB];

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
    const int k_begin = blockIdx.z * g.k_per_split;
    const int k_end = min(g.K, k_begin + g.k_per_split);
    if (k_begin >= k_end && !(blockIdx.z == 0)) return;

    floatx4 acc[WM_T][WN_T];
#pragma unroll
    for (int i = 0; i < WM_T; ++i)
#pragma unroll
        for (int j = 0; j < WN_T; ++j) acc[i][j] = (floatx4){0.f, 0.f, 0.f, 0.f};

    TileStage<BM, LDA> sa;
    TileStage<BN, LDB> sb;
    const int nk = (k_end - k_begin + BK - 1) / BK;
    if (nk > 0) {
        sa.load(g.A, g.lda, g.a_kcontig, g.a_vec, m0, g.M, k_begin, k_end);
        sb.load(g.B, g.ldb, g.b_kcontig, g.b_vec, n0, g.N, k_begin, k_end);
        sa.store(As[0], g.a_kcontig);
        sb.store(Bs[0], g.b_kcontig);
    }
    __syncthreads();
    const int l15 = lane & 15, lk = lane >> 4;
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) {
            sa.load(g.A, g.lda, g.a_kcontig, g.a_vec, m0, g.M, k_begin + (kt + 1) * BK, k_end);
            sb.load(g.B, g.ldb, g.b_kcontig, g.b_vec, n0, g.N, k_begin + (kt + 1) * BK, k_end);
        }
        const float* as = As[cur];
        const float* bs = Bs[cur];
#pragma unroll
        for (int kk = 0; kk < BK / 4; ++kk) {
            const int kr = kk * 4 + lk;
            float af[WM_T], bf[WN_T];
#pragma unroll
            for (int i = 0; i < WM_T; ++i) af[i] = as[kr * LDA + swz(kr, (wm * WM_T + i) * 16 + l15)];
#pragma unroll
            for (int j = 0; j < WN_T; ++j) bf[j] = bs[kr * LDB + swz(kr, (wn * WN_T + j) * 16 + l15)];
#pragma unroll
            for (int i = 0; i < WM_T; ++i)
#pragma unroll
                for (int j = 0; j < WN_T; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[j], af[i], acc[i][j], 0, 0, 0);
        }
        if (kt + 1 < nk) {
            sa.store(As[cur ^ 1], g.a_kcontig);
            sb.store(Bs[cur ^ 1], g.b_kcontig);
        }
        __syncthreads();
    }

    // epilogue: lane owns C[m][n..n+3], m = tile row (lane&15), n = tile col group (lane>>4)*4
    const bool add_bias = g.bias != nullptr && blockIdx.z == 0;
#pragma unroll
    for (int i = 0; i < WM_T; ++i) {
        const int m = m0 + (wm * WM_T + i) * 16 + l15;
        if (m >= g.M) continue;
#pragma unroll
        for (int j = 0; j < WN_T; ++j) {
            const int n = n0 + (wn * WN_T + j) * 16 + lk * 4;
            if (n >= g.N) continue;
            float r[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
            float* c = g.C + (long)m * g.ldc + n;
            const bool full = (n + 3 < g.N);
            if (add_bias) {
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (n + q < g.N) r[q] += g.bias[n + q];
            }
            if (g.atomic_out) {
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (n + q < g.N) unsafeAtomicAdd(c + q, r[q]);
            } else if (full && g.c_vec) {
                float4 o = make_float4(r[0], r[1], r[2], r[3]);
                if (g.accumulate) {
                    float4 old = *reinterpret_cast<const float4*>(c);
                    o.x += old.x; o.y += old.y; o.z += old.z; o.w += old.w;
                }
                *reinterpret_cast<float4*>(c) = o;
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (n + q < g.N) c[q] = g.accumulate ? c[q] + r[q] : r[q];
            }
        }
    }
}

template <int WAVES_M, int WAVES_N, int WM_T, int WN_T>
static void launch(const GemmArgs& g, int splits, hipStream_t s) {
    constexpr int BM = WAVES_M * WM_T * 16, BN = WAVES_N * WN_T * 16;
    dim3 grid(cdiv(g.M, BM), cdiv(g.N, BN), splits);
    hipLaunchKernelGGL((gemm_f32_kernel<WAVES_M, WAVES_N, WM_T, WN_T>), grid, dim3(256), 0, s, g);
}

}  // namespace i3d

using namespace i3d;

extern "C" int i3d_gemm_f32(int trans_a, int trans_b, int M, int N, int K, const float* A, int lda, const float* B,
                            int ldb, float* C, int ldc, const float* bias, int accumulate, void* stream) {
    I3D_CHECK_ARG(M >= 0 && N >= 0 && K >= 0, "negative dimension");
    I3D_CHECK_ARG(lda >= (trans_a ? M : K) && ldb >= (trans_b ? K : N) && ldc >= N, "leading dimension too small");
    if (M == 0 || N == 0) return I3D_OK;
    hipStream_t s = (hipStream_t)stream;
    GemmArgs g;
    g.A = A; g.B = B; g.C = C; g.bias = bias;
    g.M = M; g.N = N; g.K = K;
    g.lda = lda; g.ldb = ldb; g.ldc = ldc;
    g.a_kcontig = trans_a ? 0 : 1;
    g.b_kcontig = trans_b ? 1 : 0;
    g.accumulate = accumulate ? 1 : 0;
    g.a_vec = (((uintptr_t)A & 15) == 0) && (lda % 4 == 0);
    g.b_vec = (((uintptr_t)B & 15) == 0) && (ldb % 4 == 0);
    g.c_vec = (((uintptr_t)C & 15) == 0) && (ldc % 4 == 0);

    // tile configuration: the towers have N in {20, 200, 256, 400}; weight gradients have small M,N and huge K.
    int cfg;  // 0: 64x208  1: 128x208  2: 128x128  3: 256x32  4: 64x64
    int bm, bn;
    if (N <= 32) { cfg = 3; bm = 256; bn = 32; }
    else if (N > 128 && N <= 208) {
        if (cdiv(M, 64) >= 1024) { cfg = 1; bm = 128; bn = 208; } else { cfg = 0; bm = 64; bn = 208; }
    } else if ((long)cdiv(M, 128) * cdiv(N, 128) >= 256) { cfg = 2; bm = 128; bn = 128; }
    else { cfg = 4; bm = 64; bn = 64; }
    int tiles = cdiv(M, bm) * cdiv(N, bn);
    int splits = 1;
    if (tiles < 256 && K >= 1024) {
        splits = cdiv(512, tiles);
        int max_splits = K / 256;
        if (splits > max_splits) splits = max_splits;
        if (splits < 1) splits = 1;
    }
    int kps = cdiv(cdiv(K, splits), BK) * BK;
    if (kps < BK) kps = BK;
    splits = K > 0 ? cdiv(K, kps) : 1;
    g.k_per_split = kps;
    g.atomic_out = splits > 1;
    if (splits > 1 && !accumulate) {
        // atomics accumulate on top of zeros
        if (ldc == N) {
            if (hipMemsetAsync(C, 0, (size_t)M * N * sizeof(float), s) != hipSuccess) {
                set_error("i3d_gemm_f32: memset failed");
                return I3D_ERR_LAUNCH;
            }
        } else {
            if (hipMemset2DAsync(C, (size_t)ldc * sizeof(float), 0, (size_t)N * sizeof(float), M, s) != hipSuccess) {
                set_error("i3d_gemm_f32: memset2d failed");
                return I3D_ERR_LAUNCH;
            }
        }
    }
    switch (cfg) {
        case 0: launch<4, 1, 1, 13>(g, splits, s); break;
        case 1: launch<4, 1, 2, 13>(g, splits, s); break;
        case 2: launch<2, 2, 4, 4>(g, splits, s); break;
        case 3: launch<4, 1, 4, 2>(g, splits, s); break;
        default: launch<2, 2, 2, 2>(g, splits, s); break;
    }
    I3D_CHECK_LAUNCH();
    return I3D_OK;
}
