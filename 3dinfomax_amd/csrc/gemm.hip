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
    unsigned a_bytes, b_bytes;  // extent of the operand views in bytes (buffer descriptor num_records)
};


__device__ __forceinline__ int swz(int k, int idx) { return idx ^ ((((k) ^ (k >> 2)) & 1) << 4); }

template <int R, int LD, int BK>
struct TileStage {
    static constexpr int SLOTS = R * BK / 4;                 // float4 slots in a tile
    static constexpr int KQ = BK / 4;                        // float4 slots along k of one row
    static constexpr int PER_THREAD = (SLOTS + 255) / 256;
    float4 v[PER_THREAD];

    // kcontig: slot -> (idx = s / 4, kq = s % 4), 4 consecutive k of one row
    // else   : slot -> (k = s / (R/4), iq = s % (R/4)), 4 consecutive idx of one k
    // BRANCH-FREE through the buffer descriptor: every lane issues the load, lanes outside the matrix pass an
    // offset beyond num_records and the hardware returns 0 - no per-lane branch, no select, so hipcc keeps the
    // loads in flight across the MFMA block and waits (counted vmcnt) only in front of the LDS store.  Per-lane
    // branches around the loads made it serialise them behind vmcnt(0) (MI355X guide 5, trap (c)): 3.5x slower.
    // VEC: one 16-byte load per slot (pointer/ld 16-byte aligned and the contiguous extent a multiple of 4).
    template <bool VEC>
    __device__ __forceinline__ void load(__amdgpu_buffer_rsrc_t rsrc, unsigned oob, int ld, int kcontig, int idx0,
                                         int idx_max, int k0, int k_end) {
#pragma unroll
        for (int it = 0; it < PER_THREAD; ++it) {
            int s = threadIdx.x + it * 256;
            int idx, k;
            if (kcontig) { idx = idx0 + s / KQ; k = k0 + (s % KQ) * 4; }
            else { k = k0 + s / (R / 4); idx = idx0 + (s % (R / 4)) * 4; }
            if (VEC) {
                // contiguous extent is a multiple of 4: a valid first element implies a valid float4
                const bool ok = idx < idx_max && k < k_end && s < SLOTS;
                const unsigned off = kcontig ? (unsigned)(idx * ld + k) * 4u : (unsigned)(k * ld + idx) * 4u;
                auto r = __builtin_amdgcn_raw_buffer_load_b128(rsrc, ok ? off : oob, 0, 0);
                static_assert(sizeof(r) == 16, "b128 load");
                v[it] = __builtin_bit_cast(float4, r);
            } else {
                float e[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int kk = kcontig ? k + u : k, ii = kcontig ? idx : idx + u;
                    const bool ok = ii < idx_max && kk < k_end && s < SLOTS;
                    const unsigned off = kcontig ? (unsigned)(ii * ld + kk) * 4u : (unsigned)(kk * ld + ii) * 4u;
                    e[u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, ok ? off : oob, 0, 0));
                }
                v[it] = make_float4(e[0], e[1], e[2], e[3]);
            }
        }
    }

    __device__ __forceinline__ void store(float* __restrict__ T, int kcontig) const {
#pragma unroll
        for (int it = 0; it < PER_THREAD; ++it) {
            int s = threadIdx.x + it * 256;
            if (s < SLOTS) {
                if (kcontig) {
                    int idx = s / KQ, k = (s % KQ) * 4;
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

template <int WAVES_M, int WAVES_N, int WM_T, int WN_T, int BK, bool VEC>
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

    TileStage<BM, LDA, BK> sa;
    TileStage<BN, LDB, BK> sb;
    // descriptors are built from kernel arguments only (provably wave-uniform: no waterfall loops, guide T20)
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.A), 0, g.a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.B), 0, g.b_bytes, 0x00020000);
    const int nk = (k_end - k_begin + BK - 1) / BK;
    if (nk > 0) {
        sa.template load<VEC>(ra, g.a_bytes, g.lda, g.a_kcontig, m0, g.M, k_begin, k_end);
        sb.template load<VEC>(rb, g.b_bytes, g.ldb, g.b_kcontig, n0, g.N, k_begin, k_end);
        sa.store(As[0], g.a_kcontig);
        sb.store(Bs[0], g.b_kcontig);
    }
    __syncthreads();
    const int l15 = lane & 15, lk = lane >> 4;
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) {
            sa.template load<VEC>(ra, g.a_bytes, g.lda, g.a_kcontig, m0, g.M, k_begin + (kt + 1) * BK, k_end);
            sb.template load<VEC>(rb, g.b_bytes, g.ldb, g.b_kcontig, n0, g.N, k_begin + (kt + 1) * BK, k_end);
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

template <int WAVES_M, int WAVES_N, int WM_T, int WN_T, int BK>
static void launch(const GemmArgs& g, int splits, bool vec, hipStream_t s) {
    constexpr int BM = WAVES_M * WM_T * 16, BN = WAVES_N * WN_T * 16;
    dim3 grid(cdiv(g.M, BM), cdiv(g.N, BN), splits);
    if (vec) hipLaunchKernelGGL((gemm_f32_kernel<WAVES_M, WAVES_N, WM_T, WN_T, BK, true>), grid, dim3(256), 0, s, g);
    else hipLaunchKernelGGL((gemm_f32_kernel<WAVES_M, WAVES_N, WM_T, WN_T, BK, false>), grid, dim3(256), 0, s, g);
}

// ---- variant on v_mfma_f32_32x32x2_f32: one 32x32 accumulator tile per MFMA (64-cycle issue = dependent latency, so
// a single accumulator chain keeps the pipe busy), half as many MFMA instructions per FLOP.  Same staging, same LDS
// image (the bit-4 swizzle is a permutation inside a 32-wide read, so the 32-lane fragment reads stay conflict-free).
typedef float floatx16 __attribute__((ext_vector_type(16)));

template <int WAVES_M, int WAVES_N, int WM_T, int WN_T, int BK, bool VEC>
__global__ void __launch_bounds__(256)
gemm_f32_m32_kernel(GemmArgs g) {
    constexpr int BM = WAVES_M * WM_T * 32, BN = WAVES_N * WN_T * 32;
    constexpr int LDA = BM, LDB = BN;
    __shared__ __attribute__((aligned(16))) float As[2][BK * LDA];
    __shared__ __attribute__((aligned(16))) float Bs[2][BK * LDB];

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
    const int k_begin = blockIdx.z * g.k_per_split;
    const int k_end = min(g.K, k_begin + g.k_per_split);
    if (k_begin >= k_end && !(blockIdx.z == 0)) return;

    floatx16 acc[WM_T][WN_T];
#pragma unroll
    for (int i = 0; i < WM_T; ++i)
#pragma unroll
        for (int j = 0; j < WN_T; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    TileStage<BM, LDA, BK> sa;
    TileStage<BN, LDB, BK> sb;
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.A), 0, g.a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.B), 0, g.b_bytes, 0x00020000);
    const int nk = (k_end - k_begin + BK - 1) / BK;
    if (nk > 0) {
        sa.template load<VEC>(ra, g.a_bytes, g.lda, g.a_kcontig, m0, g.M, k_begin, k_end);
        sb.template load<VEC>(rb, g.b_bytes, g.ldb, g.b_kcontig, n0, g.N, k_begin, k_end);
        sa.store(As[0], g.a_kcontig);
        sb.store(Bs[0], g.b_kcontig);
    }
    __syncthreads();
    const int l31 = lane & 31, lk = lane >> 5;
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) {
            sa.template load<VEC>(ra, g.a_bytes, g.lda, g.a_kcontig, m0, g.M, k_begin + (kt + 1) * BK, k_end);
            sb.template load<VEC>(rb, g.b_bytes, g.ldb, g.b_kcontig, n0, g.N, k_begin + (kt + 1) * BK, k_end);
        }
        const float* as = As[cur];
        const float* bs = Bs[cur];
#pragma unroll
        for (int kk = 0; kk < BK / 2; ++kk) {
            const int kr = kk * 2 + lk;
            float af[WM_T], bf[WN_T];
#pragma unroll
            for (int i = 0; i < WM_T; ++i) af[i] = as[kr * LDA + swz(kr, (wm * WM_T + i) * 32 + l31)];
#pragma unroll
            for (int j = 0; j < WN_T; ++j) bf[j] = bs[kr * LDB + swz(kr, (wn * WN_T + j) * 32 + l31)];
#pragma unroll
            for (int i = 0; i < WM_T; ++i)
#pragma unroll
                for (int j = 0; j < WN_T; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(bf[j], af[i], acc[i][j], 0, 0, 0);
        }
        if (kt + 1 < nk) {
            sa.store(As[cur ^ 1], g.a_kcontig);
            sb.store(Bs[cur ^ 1], g.b_kcontig);
        }
        __syncthreads();
    }

    // epilogue: D[row = n][col = m]: lane owns m = tile col (lane&31), n = 8*grp + 4*(lane>>5) + 0..3 per register group
    const bool add_bias = g.bias != nullptr && blockIdx.z == 0;
#pragma unroll
    for (int i = 0; i < WM_T; ++i) {
        const int m = m0 + (wm * WM_T + i) * 32 + l31;
        if (m >= g.M) continue;
#pragma unroll
        for (int j = 0; j < WN_T; ++j) {
#pragma unroll
            for (int grp = 0; grp < 4; ++grp) {
                const int n = n0 + (wn * WN_T + j) * 32 + 8 * grp + 4 * lk;
                if (n >= g.N) continue;
                float r[4] = {acc[i][j][4 * grp + 0], acc[i][j][4 * grp + 1], acc[i][j][4 * grp + 2], acc[i][j][4 * grp + 3]};
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
}

template <int WAVES_M, int WAVES_N, int WM_T, int WN_T, int BK>
static void launch32(const GemmArgs& g, int splits, bool vec, hipStream_t s) {
    constexpr int BM = WAVES_M * WM_T * 32, BN = WAVES_N * WN_T * 32;
    dim3 grid(cdiv(g.M, BM), cdiv(g.N, BN), splits);
    if (vec) hipLaunchKernelGGL((gemm_f32_m32_kernel<WAVES_M, WAVES_N, WM_T, WN_T, BK, true>), grid, dim3(256), 0, s, g);
    else hipLaunchKernelGGL((gemm_f32_m32_kernel<WAVES_M, WAVES_N, WM_T, WN_T, BK, false>), grid, dim3(256), 0, s, g);
}

// tile configurations: {BM, BN, BK}
constexpr int N_CFG = 12;   // 9..11: the 32x32x2 MFMA variant
static const int CFG_BM[N_CFG] = {64, 128, 128, 256, 64, 64, 128, 128, 32, 64, 128, 64};
static const int CFG_BN[N_CFG] = {208, 208, 128, 32, 64, 64, 128, 64, 64, 64, 128, 64};
static const int CFG_BK[N_CFG] = {16, 16, 16, 16, 16, 32, 32, 32, 32, 16, 16, 32};

}  // namespace i3d

using namespace i3d;

static int gemm_impl(int trans_a, int trans_b, int M, int N, int K, const float* A, int lda, const float* B, int ldb,
                     float* C, int ldc, const float* bias, int accumulate, int force_cfg, int force_splits,
                     void* stream);

extern "C" int i3d_gemm_f32(int trans_a, int trans_b, int M, int N, int K, const float* A, int lda, const float* B,
                            int ldb, float* C, int ldc, const float* bias, int accumulate, void* stream) {
    return gemm_impl(trans_a, trans_b, M, N, K, A, lda, B, ldb, C, ldc, bias, accumulate, -1, 0, stream);
}

extern "C" int i3d_gemm_f32_ex(int trans_a, int trans_b, int M, int N, int K, const float* A, int lda, const float* B,
                               int ldb, float* C, int ldc, const float* bias, int accumulate, int tile_cfg,
                               int splits, void* stream) {
    return gemm_impl(trans_a, trans_b, M, N, K, A, lda, B, ldb, C, ldc, bias, accumulate, tile_cfg, splits, stream);
}

static int gemm_impl(int trans_a, int trans_b, int M, int N, int K, const float* A, int lda, const float* B, int ldb,
                     float* C, int ldc, const float* bias, int accumulate, int force_cfg, int force_splits,
                     void* stream) {
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
    {
        const long a_rows = trans_a ? K : M, a_cols = trans_a ? M : K, b_rows = trans_b ? N : K, b_cols = trans_b ? K : N;
        const long ab = a_rows > 0 ? ((a_rows - 1) * lda + a_cols) * 4 : 0, bb = b_rows > 0 ? ((b_rows - 1) * ldb + b_cols) * 4 : 0;
        I3D_CHECK_ARG(ab < (1L << 32) - 16 && bb < (1L << 32) - 16, "operand view larger than 4 GiB (32-bit buffer offsets)");
        g.a_bytes = (unsigned)ab;
        g.b_bytes = (unsigned)bb;
    }
    g.a_vec = (((uintptr_t)A & 15) == 0) && (lda % 4 == 0);
    g.b_vec = (((uintptr_t)B & 15) == 0) && (ldb % 4 == 0);
    g.c_vec = (((uintptr_t)C & 15) == 0) && (ldc % 4 == 0);

    // tile configuration: the towers have N in {20, 200, 256, 400}; weight gradients have small M,N and huge K.
    int cfg;  // 0: 64x208  1: 128x208  2: 128x128  3: 256x32  4: 64x64
    int bm, bn;
    // measured on MI355X at the step's shapes (tools/gemm_bench.py, profiles/gemm_bench_r1.log): the batch is small
    // for a 256-CU chip, so many 64x64 tiles beat fewer big ones until there are thousands of tiles.
    const long tiles64 = (long)cdiv(M, 64) * cdiv(N, 64);
    if (N <= 32) { cfg = 3; bm = 256; bn = 32; }
    else if (tiles64 >= 4096) { cfg = 2; bm = 128; bn = 128; }
    else if (trans_a && tiles64 < 64) { cfg = 8; bm = 32; bn = 64; }    // tiny weight-gradient outputs: 32x64x32
    else { cfg = 9; bm = 64; bn = 64; }    // 64x64x16 on v_mfma_f32_32x32x2_f32: 3-5 % ahead of the 16x16x4 form
    if (force_cfg >= 0) {
        I3D_CHECK_ARG(force_cfg < N_CFG, "tile_cfg out of range");
        cfg = force_cfg; bm = CFG_BM[cfg]; bn = CFG_BN[cfg];
    }
    const int BK = CFG_BK[cfg];
    int tiles = cdiv(M, bm) * cdiv(N, bn);
    int splits = 1;
    // split-K (fp32 atomics, order not deterministic) ONLY for the row-reduction GEMMs of the backward pass
    // (trans_a: dW = dY^T X, K = number of rows).  Forward GEMMs stay single-pass and bit-deterministic, so that
    // the arg-max/arg-min routing of the aggregators and readouts cannot flip from run to run on near-ties.
    if (trans_a && tiles < 768 && K >= 1024) {   // until ~1024 workgroups, >= 512 of K per split
        splits = (1024 + tiles / 2) / tiles;
        int max_splits = K / 512;
        if (splits > max_splits) splits = max_splits;
        if (splits < 1) splits = 1;
    }
    if (force_splits > 0) splits = force_splits;
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
    // fast path: 16-byte loads need aligned pointers / leading dimensions and contiguous extents that are
    // multiples of 4 (K for k-contiguous operands, M or N for the others)
    const bool vec = g.a_vec && g.b_vec && ((g.a_kcontig ? K : M) % 4 == 0) && ((g.b_kcontig ? K : N) % 4 == 0);
    switch (cfg) {
        case 0: launch<4, 1, 1, 13, 16>(g, splits, vec, s); break;
        case 1: launch<4, 1, 2, 13, 16>(g, splits, vec, s); break;
        case 2: launch<2, 2, 4, 4, 16>(g, splits, vec, s); break;
        case 3: launch<4, 1, 4, 2, 16>(g, splits, vec, s); break;
        case 4: launch<2, 2, 2, 2, 16>(g, splits, vec, s); break;
        case 5: launch<2, 2, 2, 2, 32>(g, splits, vec, s); break;
        case 6: launch<2, 2, 4, 4, 32>(g, splits, vec, s); break;
        case 7: launch<2, 2, 4, 2, 32>(g, splits, vec, s); break;
        case 8: launch<2, 2, 1, 2, 32>(g, splits, vec, s); break;
        case 9: launch32<2, 2, 1, 1, 16>(g, splits, vec, s); break;
        case 10: launch32<2, 2, 2, 2, 16>(g, splits, vec, s); break;
        default: launch32<2, 2, 1, 1, 32>(g, splits, vec, s); break;
    }
    I3D_CHECK_LAUNCH();
    return I3D_OK;
}
