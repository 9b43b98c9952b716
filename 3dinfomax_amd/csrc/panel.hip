// Row-panel GEMM for the chain's products (round 6): C[M, N] (+)= A[M, K] op(W) (+ bias) with N a few hundred - the nn.Linear products of
// a PNA layer (reference models/base_layers.py:101 forward; autograd's data gradient dX = dY W) at hidden 200.
//
// The tiled kernels of gemm.hip cut such a product into 64 x 64 tiles: every 64-row slab of A is loaded - and, in the split form,
// split into its three bf16 images - by every column tile's workgroup (4 x at N = 200, 13 x at N = 800), the weight tile likewise by
// every row tile's workgroup, and N = 200 pads to 256 columns (profiles/r05_split_products.txt: MFMA busy 0.22 of the CU's cycles).
// Here
//   * the WEIGHT is split ONCE per optimisation step by a pack kernel into its three bf16 images hi | mid | lo (x = hi + mid + lo
//     exactly: gemm.hip split_pair), laid out in global memory exactly as the LDS image the MFMA fragments are read from:
//     [column block of 208][K-step of 32][13 column tiles][3 images][64 lanes][8 bf16] - a K-step's slice of a column block is 39
//     pieces of 1 KiB that go global -> LDS by LDS-DMA (global_load_lds_dwordx4, no staging registers, no ds_write), and a lane's B
//     fragment is ONE conflict-free ds_read_b128 at lane * 16;
//   * a workgroup (4 waves) owns 64 rows x one 208-column block (13 tiles of 16: 200 = 12.5 tiles, 4 % padding instead of 22 %); a
//     wave owns 32 rows x 7 (or 6) column tiles; its A fragments never touch the LDS: a lane loads the 8 consecutive k of its row
//     straight from global memory (32 B per lane, 128 B contiguous per row and K-step), splits them ONCE and uses them for all its
//     column tiles;
//   * v_mfma_f32_16x16x32_bf16, the six part products of order <= 2 (hh, hm, mh, hl, lh, mm; each exact in the fp32 accumulator;
//     dropped: <= 3 x 2^-24 |a b| - the same arithmetic as gemm.hip's split form), issued product-major so that consecutive MFMAs
//     never share an accumulator; mfma(B fragment, A fragment): a lane owns 4 consecutive columns of one row -> 16-byte stores.
#include <stdlib.h>

#include <algorithm>
#include <type_traits>

#include "common.h"

namespace i3d {
namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

constexpr int PN = 208, PT = 13, PBK = 32, PBM = 64;
constexpr int PIECE = 1024;                       // one (column tile, image) of a K-step: 64 lanes x 16 bytes
constexpr int STEP_BYTES = PT * 3 * PIECE;        // 39 KiB per K-step and column block

// (a, b) -> packed bf16 pairs hi | mid | lo with a = hi.a + mid.a + lo.a exactly (gemm.hip: split_pair)
__device__ __forceinline__ void split_pair(float a, float b, unsigned& hi, unsigned& mid, unsigned& lo) {
    auto pack = [](f32x2_t v) { return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t)); };
    auto widen = [](unsigned u) { f32x2_t r; r.x = __uint_as_float(u << 16); r.y = __uint_as_float(u & 0xffff0000u); return r; };
    f32x2_t v;
    v.x = a; v.y = b;
    hi = pack(v);
    const f32x2_t r1 = v - widen(hi);
    mid = pack(r1);
    const f32x2_t r2 = r1 - widen(mid);
    lo = pack(r2);
}
__device__ __forceinline__ void split8(const float4 lo4, const float4 hi4, bf16x8& h, bf16x8& m, bf16x8& l) {
    uint4 uh, um, ul;
    split_pair(lo4.x, lo4.y, uh.x, um.x, ul.x);
    split_pair(lo4.z, lo4.w, uh.y, um.y, ul.y);
    split_pair(hi4.x, hi4.y, uh.z, um.z, ul.z);
    split_pair(hi4.z, hi4.w, uh.w, um.w, ul.w);
    h = __builtin_bit_cast(bf16x8, uh);
    m = __builtin_bit_cast(bf16x8, um);
    l = __builtin_bit_cast(bf16x8, ul);
}

// ---- pack: W -> [col block][K-step][13][3][64][8] bf16 -------------------------------------------------------------------------
// trans = 1: B[n][k] = W[n * ldw + k] (a Linear's forward: W stored [out, in]);  0: B[n][k] = W[k * ldw + n] (its data gradient)
struct PackArgs {
    const float* W;
    int ldw, N, K, trans;
    unsigned short* out;
};

constexpr int PACK_MAX = 8;
struct PackMulti {
    PackArgs a[PACK_MAX];
};

// blockIdx.z: which weight (one launch packs every weight of a layer); blocks beyond a weight's own (K-steps, column blocks) leave
__global__ void __launch_bounds__(256) panel_pack_kernel(PackMulti m) {
    const PackArgs a = m.a[blockIdx.z];
    const int KT = (a.K + PBK - 1) / PBK;
    const int cb = blockIdx.y, t = blockIdx.x;
    if (t >= KT || cb * PN >= a.N) return;
    unsigned short* dst = a.out + ((long)(cb * KT + t) * STEP_BYTES) / 2;
    for (int q = threadIdx.x; q < PT * 64; q += 256) {       // one (column tile, lane) per trip: 8 k of one column
        const int j = q / 64, lane = q % 64;
        const int n = cb * PN + j * 16 + (lane & 15), k0 = t * PBK + (lane >> 4) * 8;
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int k = k0 + e;
            v[e] = (n < a.N && k < a.K) ? (a.trans ? a.W[(long)n * a.ldw + k] : a.W[(long)k * a.ldw + n]) : 0.f;
        }
        uint4 uh, um, ul;
        split_pair(v[0], v[1], uh.x, um.x, ul.x);
        split_pair(v[2], v[3], uh.y, um.y, ul.y);
        split_pair(v[4], v[5], uh.z, um.z, ul.z);
        split_pair(v[6], v[7], uh.w, um.w, ul.w);
        uint4* p = reinterpret_cast<uint4*>(dst + ((long)(j * 3) * PIECE) / 2) + lane;
        p[0] = uh;
        p[PIECE / 16] = um;
        p[2 * PIECE / 16] = ul;
    }
}

// sum over the 16 lanes of a DPP row (all lanes get it)
__device__ __forceinline__ float row16_sum(float x) {
    auto dpp = [](float v, auto ctrl) {
        return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), decltype(ctrl)::value, 0xf, 0xf, true));
    };
    x += dpp(x, std::integral_constant<int, 0xB1>{});       // quad_perm [1, 0, 3, 2]
    x += dpp(x, std::integral_constant<int, 0x4E>{});       // quad_perm [2, 3, 0, 1]
    x += dpp(x, std::integral_constant<int, 0x141>{});      // row_half_mirror
    x += dpp(x, std::integral_constant<int, 0x140>{});      // row_mirror
    return x;
}

// ---- the product -------------------------------------------------------------------------------------------------------------------
struct PanelArgs {
    const float* A;
    int lda, M, K, N;
    const unsigned char* Bp;     // packed weight (panel_pack_kernel)
    float* C;
    int ldc;
    const float* bias;           // [N] or null
    int accumulate;              // C += product
    // fused BatchNorm (the forward of a block behind a BatchNorm whose output is never materialised: fused_bn.hip)
    const float* aff;            // AFF: [3 K] mean | scale | shift - A is read as (A[m][k] - mean[k]) * scale[k] + shift[k]
    int act;                     // STATS: activation of the stored value (none / ReLU / LeakyReLU)
    float* stats;                // STATS: [2 ceil(M / 64)][3][N] per 32-row tile and column {sum, M2 about the tile mean, row count} of
                                 // the stored values (bn_finalize_partials_kernel merges the tiles)
};

// RT: 16-row tiles per wave (2: 64 rows per workgroup; 1: 32 rows - twice the workgroups where 64-row slabs leave CUs idle)
template <bool AFF, bool STATS, int RT>
__global__ void __launch_bounds__(256, 2) panel_gemm_kernel(PanelArgs g) {
    I3D_CHAIN_PRIO();
    __shared__ __attribute__((aligned(16))) unsigned char Bs[2][STEP_BYTES];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int rt = wave >> 1, ch = wave & 1;
    const int j0 = ch * 7, nj = ch ? PT - 7 : 7;
    const int cb = blockIdx.y;
    const int m0 = blockIdx.x * (32 * RT) + rt * (16 * RT);
    const int KT = (g.K + PBK - 1) / PBK;
    const unsigned char* bsrc = g.Bp + (long)cb * KT * STEP_BYTES;

    // The B pieces are issued through inline asm: hipcc does not count asm memory operations, so NO compiler-made s_waitcnt vmcnt(0)
    // lands between the prefetch of step t + 1 and the MFMAs of step t (with the builtin every ds_read of the B image was preceded by
    // one: the prefetch was drained before the products began).  The wait is ours: one vmcnt(0) at the top of a step.
    const unsigned bs_base = (unsigned)(unsigned long long)(lds_ptr_t)&Bs[0][0];
    auto stage_b = [&](int t, int buf) {      // 39 pieces of 1 KiB over 4 waves: global -> LDS by LDS-DMA
        const unsigned char* src = bsrc + (long)t * STEP_BYTES + lane * 16;
#pragma unroll
        for (int i = 0; i < 10; ++i) {
            const int p = wave + 4 * i;
            if (p < PT * 3) {
                unsigned keep;
                const unsigned dst = __builtin_amdgcn_readfirstlane(bs_base + (unsigned)(buf * STEP_BYTES + p * PIECE));
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                             : "=&s"(keep)
                             : "v"(src + p * PIECE), "s"(dst)
                             : "memory");
            }
        }
    };
    const int arow = lane & 15, akc = (lane >> 4) * 8;
    // A: ordinary loads (the compiler keeps their values safe: asm-load results may be copied before they land); issued right after
    // the step's barrier and first used BEHIND the step's MFMAs, where the compiler's vmcnt(0) is the wait the next step needs anyway
    struct AStep {
        float4 v[RT][2];
        float4 f[3][2];      // AFF: mean | scale | shift of the lane's 8 k
    };
    auto load_a = [&](int t, AStep& r) {
        const int k = t * PBK + akc;
#pragma unroll
        for (int i = 0; i < RT; ++i) {
            const int row = m0 + i * 16 + arow;
            if (row < g.M && k < g.K) {            // K % 8 == 0: a lane's 8 k are valid or not as a whole
                const float4* p = reinterpret_cast<const float4*>(g.A + (long)row * g.lda + k);
                r.v[i][0] = p[0];
                r.v[i][1] = p[1];
            } else {
                r.v[i][0] = r.v[i][1] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
        if (AFF) {
            const int kc = min(k, g.K - 8);
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const float4* p = reinterpret_cast<const float4*>(g.aff + (long)q * g.K + kc);
                r.f[q][0] = p[0];
                r.f[q][1] = p[1];
            }
        }
    };
    auto split_a = [&](int t, const AStep& r, bf16x8 (&h)[RT], bf16x8 (&m)[RT], bf16x8 (&l)[RT]) {
        const bool kok = t * PBK + akc < g.K;
#pragma unroll
        for (int i = 0; i < RT; ++i) {
            float4 x0 = r.v[i][0], x1 = r.v[i][1];
            if (AFF) {       // gemm.hip store_aff's expression: the same bits; lanes outside the matrix stay exact zeros
                const bool ok = kok && m0 + i * 16 + arow < g.M;
                auto af = [&](float x, float mu, float sc, float sh) { return ok ? (x - mu) * sc + sh : 0.f; };
                x0 = make_float4(af(x0.x, r.f[0][0].x, r.f[1][0].x, r.f[2][0].x), af(x0.y, r.f[0][0].y, r.f[1][0].y, r.f[2][0].y),
                                 af(x0.z, r.f[0][0].z, r.f[1][0].z, r.f[2][0].z), af(x0.w, r.f[0][0].w, r.f[1][0].w, r.f[2][0].w));
                x1 = make_float4(af(x1.x, r.f[0][1].x, r.f[1][1].x, r.f[2][1].x), af(x1.y, r.f[0][1].y, r.f[1][1].y, r.f[2][1].y),
                                 af(x1.z, r.f[0][1].z, r.f[1][1].z, r.f[2][1].z), af(x1.w, r.f[0][1].w, r.f[1][1].w, r.f[2][1].w));
            }
            split8(x0, x1, h[i], m[i], l[i]);
        }
    };

    floatx4 acc[RT][7];
#pragma unroll
    for (int i = 0; i < RT; ++i)
#pragma unroll
        for (int j = 0; j < 7; ++j) acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};

    bf16x8 ah[RT], am[RT], al[RT];
    {
        AStep a0;
        stage_b(0, 0);
        load_a(0, a0);
        split_a(0, a0, ah, am, al);
    }
    for (int t = 0; t < KT; ++t) {
        const int buf = t & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's pieces of B(t) have landed
        __syncthreads();          // every wave's pieces of B(t) are in Bs[buf]; every wave has left step t - 1 (Bs[buf ^ 1] is free)
        AStep an;
        if (t + 1 < KT) {
            stage_b(t + 1, buf ^ 1);
            load_a(t + 1, an);
        }
        const unsigned char* bl = &Bs[buf][0] + lane * 16;
#pragma unroll
        for (int jj = 0; jj < 7; ++jj) {
            if (jj < nj) {
                const int j = j0 + jj;
                const bf16x8 bh = *reinterpret_cast<const bf16x8*>(bl + (j * 3 + 0) * PIECE);
                const bf16x8 bm = *reinterpret_cast<const bf16x8*>(bl + (j * 3 + 1) * PIECE);
                const bf16x8 bo = *reinterpret_cast<const bf16x8*>(bl + (j * 3 + 2) * PIECE);
                // small products first; consecutive MFMAs alternate between the row tiles' accumulators
#pragma unroll
                for (int i = 0; i < RT; ++i) acc[i][jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bo, ah[i], acc[i][jj], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < RT; ++i) acc[i][jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh, al[i], acc[i][jj], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < RT; ++i) acc[i][jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bm, am[i], acc[i][jj], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < RT; ++i) acc[i][jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bm, ah[i], acc[i][jj], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < RT; ++i) acc[i][jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh, am[i], acc[i][jj], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < RT; ++i) acc[i][jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh, ah[i], acc[i][jj], 0, 0, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);       // the split of step t + 1 (and its wait for the A values) stays behind the MFMAs
        if (t + 1 < KT) split_a(t + 1, an, ah, am, al);
    }
    // epilogue: lane owns row (lane & 15) of a row tile, columns (lane >> 4) * 4 .. + 3 of a column tile
    const int row_lo = m0 + (lane & 15);
    bool rok[RT];
#pragma unroll
    for (int i = 0; i < RT; ++i) rok[i] = row_lo + 16 * i < g.M;
#pragma unroll
    for (int jj = 0; jj < 7; ++jj) {
        if (jj >= nj) continue;
        const int col = cb * PN + (j0 + jj) * 16 + (lane >> 4) * 4;
        const bool cok = col < g.N;               // N % 4 == 0
        float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (cok && g.bias != nullptr) b4 = *reinterpret_cast<const float4*>(g.bias + col);
        floatx4 v[RT];
#pragma unroll
        for (int i = 0; i < RT; ++i) {
            v[i] = acc[i][jj];
            v[i][0] += b4.x; v[i][1] += b4.y; v[i][2] += b4.z; v[i][3] += b4.w;
            if (rok[i] && cok) {
                float4* c = reinterpret_cast<float4*>(g.C + (long)(row_lo + 16 * i) * g.ldc + col);
                if (g.accumulate) {
                    const float4 o = *c;
                    v[i][0] += o.x; v[i][1] += o.y; v[i][2] += o.z; v[i][3] += o.w;
                }
                if (STATS) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) v[i][q] = apply_act_c<false>(v[i][q], g.act);
                }
                *c = make_float4(v[i][0], v[i][1], v[i][2], v[i][3]);
            }
        }
        if (STATS) {
            // column statistics of this wave's 16 RT-row tile: lane sums over its rows, then over the 16 lanes that share the columns
            // (xor 1, 2, 4, 8: a fixed tree), the tile mean, M2 about it the same way
            // (the 16 lanes are one DPP row: quad_perm xor 1, xor 2, row_half_mirror, row_mirror - VALU adds, no LDS crossbar;
            // both lanes of a pair add the same two numbers: the same bits in every lane)
            float s[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < RT; ++i) {
#pragma unroll
                for (int q = 0; q < 4; ++q) s[q] += rok[i] ? v[i][q] : 0.f;
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) s[q] = row16_sum(s[q]);
            const int rows_here = min(max(g.M - m0, 0), 16 * RT);
            const float cnt = (float)rows_here, inv_cnt = rows_here > 0 ? 1.f / cnt : 0.f;
            float m2[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float mean_t = s[q] * inv_cnt;
                m2[q] = 0.f;
#pragma unroll
                for (int i = 0; i < RT; ++i) {
                    const float d = v[i][q] - mean_t;
                    m2[q] += rok[i] ? d * d : 0.f;
                }
                m2[q] = row16_sum(m2[q]);
            }
            if ((lane & 15) == 0 && cok) {
                float* o = g.stats + (long)(blockIdx.x * 2 + rt) * 3 * g.N + col;      // (STATS: RT == 2)
                *reinterpret_cast<float4*>(o) = make_float4(s[0], s[1], s[2], s[3]);
                *reinterpret_cast<float4*>(o + g.N) = make_float4(m2[0], m2[1], m2[2], m2[3]);
                *reinterpret_cast<float4*>(o + 2 * g.N) = make_float4(cnt, cnt, cnt, cnt);
            }
        }
    }
}

}  // namespace
}  // namespace i3d

using namespace i3d;

extern "C" long i3d_panel_packed_bytes(int N, int K) {
    return (long)cdiv(N, PN) * cdiv(K, PBK) * STEP_BYTES;
}

extern "C" int i3d_panel_pack_multi(const I3dPanelPack* w, int n, void* stream) {
    I3D_CHECK_ARG(w != nullptr && n >= 1 && n <= PACK_MAX, "1..8 weights per launch");
    PackMulti m = {};
    int kt = 1, nb = 1;
    for (int i = 0; i < n; ++i) {
        I3D_CHECK_ARG(w[i].W != nullptr && w[i].packed != nullptr && w[i].N > 0 && w[i].K > 0 && w[i].ldw >= (w[i].trans ? w[i].K : w[i].N) &&
                          (((uintptr_t)w[i].packed) & 15) == 0, "bad arguments");
        m.a[i] = PackArgs{w[i].W, w[i].ldw, w[i].N, w[i].K, w[i].trans, (unsigned short*)w[i].packed};
        kt = std::max(kt, cdiv(w[i].K, PBK));
        nb = std::max(nb, cdiv(w[i].N, PN));
    }
    hipLaunchKernelGGL(panel_pack_kernel, dim3(kt, nb, n), dim3(256), 0, (hipStream_t)stream, m);
    I3D_CHECK_LAUNCH();
    return I3D_OK;
}

extern "C" int i3d_panel_pack(const float* W, int ldw, int N, int K, int trans, void* packed, void* stream) {
    const I3dPanelPack w{W, ldw, N, K, trans, packed};
    return i3d_panel_pack_multi(&w, 1, stream);
}

static int panel_launch(int M, int N, int K, const float* A, int lda, const void* packed, float* C, int ldc, const float* bias,
                        int accumulate, const float* aff, int act, float* stats, void* stream) {
    I3D_CHECK_ARG(M > 0 && N > 0 && K > 0 && A != nullptr && packed != nullptr && C != nullptr, "bad arguments");
    I3D_CHECK_ARG(K % 8 == 0 && N % 4 == 0 && lda % 4 == 0 && ldc % 4 == 0 && lda >= K && ldc >= N, "K % 8, N % 4, 16-byte rows");
    I3D_CHECK_ARG(((((uintptr_t)A) | ((uintptr_t)C) | ((uintptr_t)packed) | ((uintptr_t)bias) | ((uintptr_t)aff) | ((uintptr_t)stats)) & 15) == 0,
                  "16-byte aligned operands");
    I3D_CHECK_ARG(stats == nullptr || relu_class(act), "epilogue activation: none, ReLU or LeakyReLU");
    PanelArgs g{A, lda, M, K, N, (const unsigned char*)packed, C, ldc, bias, accumulate ? 1 : 0, aff, act, stats};
    // 64-row slabs where they fill the chip, 32-row slabs (same kernel, one row tile per wave) where they would leave CUs idle;
    // the statistics form keeps 64-row workgroups (its tile count is part of the interface: i3d_panel_stats_tiles)
    static const int force_rt = [] { const char* e = getenv("I3D_PANEL_RT"); return e ? atoi(e) : 0; }();
    // (32-row slabs: 37 -> 27 us stand-alone at [N,600]x[600,200], but every workgroup streams the whole packed weight - next to the
    // weight-gradient panels the 64-row form is level or ahead in the step: 1.839 vs 1.831 ms - so only where < 128 slabs exist.
    // Also measured and dropped: the output tile staged through the LDS for whole-row stores - 20.7 vs 20.2 us, step 1.834 vs 1.840;
    // the LDS-DMA pieces of step t + 1 issued between the column tiles' MFMA groups instead of in front of them - 22.1 vs 20.7 us.)
    const bool rt1 = stats == nullptr && (force_rt ? force_rt == 1 : (long)cdiv(M, PBM) * cdiv(N, PN) < 128);
    const dim3 grid(cdiv(M, rt1 ? 32 : PBM), cdiv(N, PN)), block(256);
    hipStream_t s = (hipStream_t)stream;
    if (aff != nullptr && stats != nullptr) hipLaunchKernelGGL((panel_gemm_kernel<true, true, 2>), grid, block, 0, s, g);
    else if (stats != nullptr) hipLaunchKernelGGL((panel_gemm_kernel<false, true, 2>), grid, block, 0, s, g);
    else if (aff != nullptr) {
        if (rt1) hipLaunchKernelGGL((panel_gemm_kernel<true, false, 1>), grid, block, 0, s, g);
        else hipLaunchKernelGGL((panel_gemm_kernel<true, false, 2>), grid, block, 0, s, g);
    } else {
        if (rt1) hipLaunchKernelGGL((panel_gemm_kernel<false, false, 1>), grid, block, 0, s, g);
        else hipLaunchKernelGGL((panel_gemm_kernel<false, false, 2>), grid, block, 0, s, g);
    }
    I3D_CHECK_LAUNCH();
    return I3D_OK;
}

extern "C" int i3d_panel_gemm(int M, int N, int K, const float* A, int lda, const void* packed, float* C, int ldc, const float* bias,
                              int accumulate, void* stream) {
    return panel_launch(M, N, K, A, lda, packed, C, ldc, bias, accumulate, nullptr, I3D_ACT_NONE, nullptr, stream);
}

extern "C" int i3d_panel_stats_tiles(int M) { return 2 * cdiv(M, PBM); }

extern "C" int i3d_panel_gemm_fused(int M, int N, int K, const float* A, int lda, const void* packed, float* C, int ldc, const float* bias,
                                    const float* a_aff, int epi_act, float* stats, void* stream) {
    I3D_CHECK_ARG(stats == nullptr || ((long)N * 4) % 16 == 0, "N % 4");
    return panel_launch(M, N, K, A, lda, packed, C, ldc, bias, 0, a_aff, epi_act, stats, stream);
}
