// Dense tower GEMMs on the fp32 matrix cores of gfx950.
//
// Replaces aten::addmm / aten::mm behind nn.Linear in FCLayer (reference models/base_layers.py:101)
// for the forward (Y = X W^T + b), the data gradient (dX = dY W) and the weight gradient
// (dW = dY^T X).  Exact fp32: v_mfma_f32_16x16x4_f32 / v_mfma_f32_32x32x2_f32 are bit-for-bit an fp32 fmaf
// chain (MI355X guide 3), so parity with the reference's CPU fp32 matmul is a summation-order question only.
//
// Structure (wave64, 4 waves / workgroup):
//  * both operand tiles are staged in LDS k-major  T[k][idx]  (idx = m or n), so the MFMA fragment of a lane
//    is one conflict-free ds_read_b32: for the 16-wide MFMA the two 16-lane halves of a 32-lane group are
//    steered to different bank halves by XOR-ing bit 4 of idx with ((k ^ (k>>2)) & 1); a 32-wide fragment read is
//    a permutation of 32 consecutive banks either way.
//  * register-staged double buffering: global loads of K-tile t+1 are issued before the MFMAs of tile t
//    and written to the other LDS buffer afterwards: one barrier per K-tile.
//  * loads go through buffer descriptors: lanes outside the matrix pass an offset beyond num_records and the
//    hardware returns 0 - no per-lane branch around a load (those made hipcc serialise the loads behind vmcnt(0):
//    3.5x slower, MI355X guide 5 trap (c)), per-slot address parts are hoisted out of the K loop.
//  * the MFMA is issued with (W-fragment, X-fragment) so the accumulator holds C^T tiles: a lane owns 4
//    consecutive output columns of one row -> the epilogue (bias, accumulate, split-K atomics) is 16-byte accesses.
//  * k-contiguous operands (X[m][k], W[n][k]) are loaded with 16-byte loads along k and transposed on the
//    LDS write; idx-contiguous operands (dY^T, W for dX) are loaded along idx and written as b128.
//  * split-K (grid.z) with fp32 atomics ONLY for the row-reduction GEMMs of the backward pass.
//  * row indirection (m_rows / k_rows) and per-m-tile weight selection (tile_group) for the degree-grouped
//    posttrans GEMMs of the PNA layer (i3d_gemm_f32_grouped): rows of one in-degree share the combined weight
//    W_D = W_id + amp(D) W_amp + att(D) W_att, which cuts K from 12F to 4F.
#include "common.h"

namespace i3d {

typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

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
    int c_vec;       // 16-byte access to C allowed
    unsigned a_bytes, b_bytes;  // extent of the operand views in bytes (buffer descriptor num_records)
    const int* m_rows;      // [M] or null: logical row m lives at row m_rows[m] of A (k-contiguous A only) and of C;
                            //          -1 = padding row (loads return 0, nothing is stored)
    const int* k_rows;      // [K] or null: reduction index k lives at row k_rows[k] of the idx-contiguous operands
    const int* tile_group;  // [ceil(M/BM)] or null: B of m-tile t is g.B + tile_group[t] * b_group_stride
    long b_group_stride;    // floats
};

__device__ __forceinline__ int swz(int k, int idx) { return idx ^ ((((k) ^ (k >> 2)) & 1) << 4); }

template <int R, int LD, int BK>
struct TileStage {
    static constexpr int SLOTS = R * BK / 4;                 // float4 slots in a tile
    static constexpr int KQ = BK / 4;                        // float4 slots along k of one row
    static constexpr int PER_THREAD = (SLOTS + 255) / 256;
    float4 v[PER_THREAD];
    unsigned base[PER_THREAD];   // loop-invariant byte offset of the slot (row part or idx part)
    int kloc[PER_THREAD];        // k of the slot inside a K-tile
    int iloc[PER_THREAD];        // idx of the slot (idx-contiguous: first of 4)
    bool ok[PER_THREAD];

    // kcontig: slot -> (idx = s / KQ, kq = s % KQ), 4 consecutive k of one row
    // else   : slot -> (k = s / (R/4), iq = s % (R/4)), 4 consecutive idx of one k
    __device__ __forceinline__ void prepare(int ld, int kcontig, int idx0, int idx_max, const int* __restrict__ rows) {
#pragma unroll
        for (int it = 0; it < PER_THREAD; ++it) {
            const int s = threadIdx.x + it * 256;
            if (kcontig) {
                const int idx = idx0 + s / KQ;
                kloc[it] = (s % KQ) * 4;
                int row = -1;
                if (idx < idx_max && s < SLOTS) row = rows ? rows[idx] : idx;
                ok[it] = row >= 0;
                iloc[it] = idx;
                base[it] = (unsigned)(max(row, 0) * ld + kloc[it]) * 4u;
            } else {
                kloc[it] = s / (R / 4);
                iloc[it] = idx0 + (s % (R / 4)) * 4;
                ok[it] = iloc[it] < idx_max && s < SLOTS;
                base[it] = (unsigned)iloc[it] * 4u;
            }
        }
    }

    template <bool VEC>
    __device__ __forceinline__ void load(__amdgpu_buffer_rsrc_t rsrc, unsigned oob, int ld, int kcontig, int idx_max,
                                         int k0, int k_end, const int* __restrict__ k_rows) {
#pragma unroll
        for (int it = 0; it < PER_THREAD; ++it) {
            const int k = k0 + kloc[it];
            unsigned off;
            const bool valid = ok[it] && k < k_end;
            if (kcontig) {
                off = base[it] + (unsigned)k0 * 4u;
            } else {
                int krow = k;
                if (k_rows != nullptr) krow = k_rows[min(k, k_end - 1)];   // unconditional (clamped) index load
                off = (unsigned)(krow * ld) * 4u + base[it];
            }
            if (VEC) {   // contiguous extent is a multiple of 4: a valid first element implies a valid float4
                auto r = __builtin_amdgcn_raw_buffer_load_b128(rsrc, valid ? off : oob, 0, 0);
                static_assert(sizeof(r) == 16, "b128 load");
                v[it] = __builtin_bit_cast(float4, r);
            } else {
                float e[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const bool oku = valid && (kcontig ? (k + u < k_end) : (iloc[it] + u < idx_max));
                    e[u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, oku ? off + 4u * u : oob, 0, 0));
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

template <int MT> struct Acc;
template <> struct Acc<16> { typedef floatx4 type; static constexpr int REGS = 4; };
template <> struct Acc<32> { typedef floatx16 type; static constexpr int REGS = 16; };

// MT = MFMA tile (16: v_mfma_f32_16x16x4_f32, 32: v_mfma_f32_32x32x2_f32); a wave computes WM_T x WN_T such tiles
template <int MT, int WAVES_M, int WAVES_N, int WM_T, int WN_T, int BK, bool VEC>
__global__ void __launch_bounds__(256)
gemm_f32_kernel(GemmArgs g) {
    constexpr int BM = WAVES_M * WM_T * MT, BN = WAVES_N * WN_T * MT;
    constexpr int LDA = (BM + 31) / 32 * 32, LDB = (BN + 31) / 32 * 32;
    constexpr int KSTEP = (MT == 16) ? 4 : 2;         // k per MFMA
    __shared__ __attribute__((aligned(16))) float As[2][BK * LDA];
    __shared__ __attribute__((aligned(16))) float Bs[2][BK * LDB];

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
    const int k_begin = blockIdx.z * g.k_per_split;
    const int k_end = min(g.K, k_begin + g.k_per_split);
    if (k_begin >= k_end && !(blockIdx.z == 0)) return;

    typename Acc<MT>::type acc[WM_T][WN_T];
#pragma unroll
    for (int i = 0; i < WM_T; ++i)
#pragma unroll
        for (int j = 0; j < WN_T; ++j)
#pragma unroll
            for (int r = 0; r < Acc<MT>::REGS; ++r) acc[i][j][r] = 0.f;

    TileStage<BM, LDA, BK> sa;
    TileStage<BN, LDB, BK> sb;
    // descriptors are built from kernel arguments / blockIdx only (wave-uniform: no waterfall loops, guide T20)
    const float* Bp = g.B;
    if (g.tile_group != nullptr) Bp += (long)g.tile_group[blockIdx.x] * g.b_group_stride;
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.A), 0, g.a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Bp), 0, g.b_bytes, 0x00020000);
    sa.prepare(g.lda, g.a_kcontig, m0, g.M, g.m_rows);
    sb.prepare(g.ldb, g.b_kcontig, n0, g.N, nullptr);
    const int nk = (k_end - k_begin + BK - 1) / BK;
    if (nk > 0) {
        sa.template load<VEC>(ra, g.a_bytes, g.lda, g.a_kcontig, g.M, k_begin, k_end, g.k_rows);
        sb.template load<VEC>(rb, g.b_bytes, g.ldb, g.b_kcontig, g.N, k_begin, k_end, g.k_rows);
        sa.store(As[0], g.a_kcontig);
        sb.store(Bs[0], g.b_kcontig);
    }
    __syncthreads();
    const int lt = lane % MT, lk = lane / MT;
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) {
            sa.template load<VEC>(ra, g.a_bytes, g.lda, g.a_kcontig, g.M, k_begin + (kt + 1) * BK, k_end, g.k_rows);
            sb.template load<VEC>(rb, g.b_bytes, g.ldb, g.b_kcontig, g.N, k_begin + (kt + 1) * BK, k_end, g.k_rows);
        }
        const float* as = As[cur];
        const float* bs = Bs[cur];
#pragma unroll
        for (int kk = 0; kk < BK / KSTEP; ++kk) {
            const int kr = kk * KSTEP + lk;
            float af[WM_T], bf[WN_T];
#pragma unroll
            for (int i = 0; i < WM_T; ++i) af[i] = as[kr * LDA + swz(kr, (wm * WM_T + i) * MT + lt)];
#pragma unroll
            for (int j = 0; j < WN_T; ++j) bf[j] = bs[kr * LDB + swz(kr, (wn * WN_T + j) * MT + lt)];
#pragma unroll
            for (int i = 0; i < WM_T; ++i)
#pragma unroll
                for (int j = 0; j < WN_T; ++j) {
                    if constexpr (MT == 16) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[j], af[i], acc[i][j], 0, 0, 0);
                    else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(bf[j], af[i], acc[i][j], 0, 0, 0);
                }
        }
        if (kt + 1 < nk) {
            sa.store(As[cur ^ 1], g.a_kcontig);
            sb.store(Bs[cur ^ 1], g.b_kcontig);
        }
        __syncthreads();
    }

    // epilogue.  D[row = n][col = m]; a lane owns one m (lane % MT) and groups of 4 consecutive n:
    //   MT = 16: n = 4*(lane>>4) + 0..3 (one group);  MT = 32: n = 8*grp + 4*(lane>>5) + 0..3, grp = 0..3
    const bool add_bias = g.bias != nullptr && blockIdx.z == 0;
    constexpr int GROUPS = (MT == 16) ? 1 : 4;
#pragma unroll
    for (int i = 0; i < WM_T; ++i) {
        const int m = m0 + (wm * WM_T + i) * MT + lt;
        if (m >= g.M) continue;
        const int row = g.m_rows ? g.m_rows[m] : m;
        if (row < 0) continue;
#pragma unroll
        for (int j = 0; j < WN_T; ++j) {
#pragma unroll
            for (int grp = 0; grp < GROUPS; ++grp) {
                const int n = n0 + (wn * WN_T + j) * MT + ((MT == 16) ? lk * 4 : 8 * grp + 4 * lk);
                if (n >= g.N) continue;
                float r[4] = {acc[i][j][4 * grp + 0], acc[i][j][4 * grp + 1], acc[i][j][4 * grp + 2], acc[i][j][4 * grp + 3]};
                float* c = g.C + (long)row * g.ldc + n;
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

template <int MT, int WAVES_M, int WAVES_N, int WM_T, int WN_T, int BK>
static void launch(const GemmArgs& g, int splits, bool vec, hipStream_t s) {
    constexpr int BM = WAVES_M * WM_T * MT, BN = WAVES_N * WN_T * MT;
    dim3 grid(cdiv(g.M, BM), cdiv(g.N, BN), splits);
    if (vec) hipLaunchKernelGGL((gemm_f32_kernel<MT, WAVES_M, WAVES_N, WM_T, WN_T, BK, true>), grid, dim3(256), 0, s, g);
    else hipLaunchKernelGGL((gemm_f32_kernel<MT, WAVES_M, WAVES_N, WM_T, WN_T, BK, false>), grid, dim3(256), 0, s, g);
}

// tile configurations {BM, BN, BK, MFMA}; the numbering is part of the tuning entry i3d_gemm_f32_ex
constexpr int N_CFG = 6;
static const int CFG_BM[N_CFG] = {128, 256, 64, 32, 64, 128};
static const int CFG_BN[N_CFG] = {128, 32, 64, 64, 64, 128};
static const int CFG_BK[N_CFG] = {16, 16, 16, 32, 16, 16};

struct Extra {
    const int* m_rows = nullptr;
    const int* k_rows = nullptr;
    const int* tile_group = nullptr;
    long b_group_stride = 0;
    long a_rows_total = -1;   // number of physical rows of A / C when m_rows is used (for the descriptor extent)
    long k_rows_total = -1;   // number of physical rows of the idx-contiguous operands when k_rows is used
};

static int gemm_impl(int trans_a, int trans_b, int M, int N, int K, const float* A, int lda, const float* B, int ldb,
                     float* C, int ldc, const float* bias, int accumulate, int force_cfg, int force_splits,
                     const Extra& ex, void* stream) {
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
    g.m_rows = ex.m_rows; g.k_rows = ex.k_rows; g.tile_group = ex.tile_group; g.b_group_stride = ex.b_group_stride;
    {
        long a_rows = trans_a ? K : M, a_cols = trans_a ? M : K, b_rows = trans_b ? N : K, b_cols = trans_b ? K : N;
        if (!trans_a && ex.m_rows) a_rows = ex.a_rows_total;
        if (trans_a && ex.k_rows) a_rows = ex.k_rows_total;
        if (!trans_b && ex.k_rows) b_rows = ex.k_rows_total;
        const long ab = a_rows > 0 ? ((a_rows - 1) * lda + a_cols) * 4 : 0, bb = b_rows > 0 ? ((b_rows - 1) * ldb + b_cols) * 4 : 0;
        I3D_CHECK_ARG(ab < (1L << 32) - 16 && bb < (1L << 32) - 16, "operand view larger than 4 GiB (32-bit buffer offsets)");
        g.a_bytes = (unsigned)ab;
        g.b_bytes = (unsigned)bb;
    }
    const bool a_al = (((uintptr_t)A & 15) == 0) && (lda % 4 == 0), b_al = (((uintptr_t)B & 15) == 0) && (ldb % 4 == 0);
    g.c_vec = (((uintptr_t)C & 15) == 0) && (ldc % 4 == 0);

    // tile configuration, measured on MI355X at the step's shapes (tools/gemm_bench.py, profiles/r01_gemm_bench_*.log):
    // the batch is small for a 256-CU chip, so many 64x64 tiles beat fewer big ones until there are thousands of tiles.
    int cfg;  // 0: 128x128x16 (16x16x4)  1: 256x32x16  2: 64x64x16 (16x16x4)  3: 32x64x32  4: 64x64x16 (32x32x2)  5: 128x128x16 (32x32x2)
    const long tiles64 = (long)cdiv(M, 64) * cdiv(N, 64);
    if (N <= 32) cfg = 1;
    else if (tiles64 >= 4096) cfg = 5;
    else if (trans_a && tiles64 < 64) cfg = 3;    // tiny weight-gradient outputs
    else cfg = 4;
    if (ex.tile_group != nullptr) cfg = 4;        // the group padding of m_rows is 64 rows
    if (force_cfg >= 0) {
        I3D_CHECK_ARG(force_cfg < N_CFG, "tile_cfg out of range");
        I3D_CHECK_ARG(ex.tile_group == nullptr || CFG_BM[force_cfg] == 64, "grouped GEMM needs 64-row tiles");
        cfg = force_cfg;
    }
    const int bm = CFG_BM[cfg], bn = CFG_BN[cfg], BK = CFG_BK[cfg];
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
        hipError_t e = (ldc == N) ? hipMemsetAsync(C, 0, (size_t)M * N * sizeof(float), s)
                                  : hipMemset2DAsync(C, (size_t)ldc * sizeof(float), 0, (size_t)N * sizeof(float), M, s);
        if (e != hipSuccess) {
            set_error("i3d_gemm_f32: memset failed");
            return I3D_ERR_LAUNCH;
        }
    }
    // fast path: 16-byte loads need aligned pointers / leading dimensions and contiguous extents that are
    // multiples of 4 (K for k-contiguous operands, M or N for the others)
    const bool vec = a_al && b_al && ((g.a_kcontig ? K : M) % 4 == 0) && ((g.b_kcontig ? K : N) % 4 == 0) &&
                     (ex.b_group_stride % 4 == 0);
    switch (cfg) {
        case 0: launch<16, 2, 2, 4, 4, 16>(g, splits, vec, s); break;
        case 1: launch<16, 4, 1, 4, 2, 16>(g, splits, vec, s); break;
        case 2: launch<16, 2, 2, 2, 2, 16>(g, splits, vec, s); break;
        case 3: launch<16, 2, 2, 1, 2, 32>(g, splits, vec, s); break;
        case 4: launch<32, 2, 2, 1, 1, 16>(g, splits, vec, s); break;
        default: launch<32, 2, 2, 2, 2, 16>(g, splits, vec, s); break;
    }
    I3D_CHECK_LAUNCH();
    return I3D_OK;
}

}  // namespace i3d

using namespace i3d;

extern "C" int i3d_gemm_f32(int trans_a, int trans_b, int M, int N, int K, const float* A, int lda, const float* B,
                            int ldb, float* C, int ldc, const float* bias, int accumulate, void* stream) {
    return gemm_impl(trans_a, trans_b, M, N, K, A, lda, B, ldb, C, ldc, bias, accumulate, -1, 0, Extra(), stream);
}

extern "C" int i3d_gemm_f32_ex(int trans_a, int trans_b, int M, int N, int K, const float* A, int lda, const float* B,
                               int ldb, float* C, int ldc, const float* bias, int accumulate, int tile_cfg,
                               int splits, void* stream) {
    return gemm_impl(trans_a, trans_b, M, N, K, A, lda, B, ldb, C, ldc, bias, accumulate, tile_cfg, splits, Extra(),
                     stream);
}

// C[m_rows[m], :] (+)= A[m_rows[m], :] * op(B_g),  g = tile_group[m / 64];  m_rows is padded with -1 to 64 per group
extern "C" int i3d_gemm_f32_grouped(int trans_b, int m_padded, int N, int K, const float* A, int lda, long a_rows_total,
                                    const int* m_rows, const int* tile_group, const float* B, int ldb,
                                    long b_group_stride, float* C, int ldc, int accumulate, void* stream) {
    I3D_CHECK_ARG(m_rows != nullptr && tile_group != nullptr && m_padded % 64 == 0, "grouped GEMM needs 64-padded m_rows");
    Extra ex;
    ex.m_rows = m_rows; ex.tile_group = tile_group; ex.b_group_stride = b_group_stride; ex.a_rows_total = a_rows_total;
    return gemm_impl(0, trans_b, m_padded, N, K, A, lda, B, ldb, C, ldc, nullptr, accumulate, -1, 0, ex, stream);
}

// C[M,N] = sum_{j < n_rows} A[k_rows[j], 0:M]^T * B[k_rows[j], 0:N]   (weight gradient over a subset of rows)
extern "C" int i3d_gemm_f32_rowsubset(int M, int N, int n_rows, const float* A, int lda, const float* B, int ldb,
                                      const int* k_rows, long rows_total, float* C, int ldc, int accumulate,
                                      void* stream) {
    I3D_CHECK_ARG(k_rows != nullptr, "k_rows required");
    Extra ex;
    ex.k_rows = k_rows; ex.k_rows_total = rows_total;
    return gemm_impl(1, 0, M, N, n_rows, A, lda, B, ldb, C, ldc, nullptr, accumulate, -1, 0, ex, stream);
}
