// The edge stage of the 3D network in one lane per edge (round 2; instruction diet + fused backward pass in round 5).
//
// reference models/net3d.py:57-81, 100-118 for the structure of the pre-training configs (propagation_depth 1, one
// message block, node embedding broadcast from one parameter):
//     f    = fourier(d)                                     commons/utils.py:103-110
//     e0   = post( BN_in( act( W_in f + b_in ) ) )          edge_input block + the outer SiLU (net3d.py:80-81)
//     m    = BN_msg( act( W_s h_src + W_d h_dst + W_e e0 + b_msg ) )      message block on [h_src | h_dst | d] (:113-115)
//     msg  = m * sigmoid( w_g . m + b_g )                   soft edge gate (:117-118)
//     m_sum[v] = mean / sum of msg over the in-edges of v   (:109)
// With h the broadcast of ONE vector, W_s h_src + W_d h_dst + b_msg is one constant vector c: the whole stage is a
// function of the scalar distance of the edge and of the two BatchNorm statistics.  A lane owns an edge and carries the
// 20-wide vectors in registers; what goes through memory is what has to exist as a tensor (the distance embedding the
// forward leaves on the graph, edge-id order) plus ONE saved [E, H] activation and one transient per direction.
//
//   forward   P   parameters packed for scalar loads (below)
//             F1  statistics of act(W_in f + b)                      -> per-tile partials -> finalisation (fused_bn.hip)
//             F2  e0 (stored, edge-id order), x_msg (stored), its statistics -> finalisation
//             F3  m, gate, msg (stored) -> i3d_segment_sum
//   backward  P
//             B1  sums of the message BatchNorm + the gate's parameter gradients        -> R1
//             B2  gradient through the message block (dW_e and dc on the MFMA unit: a wave's 64 edges are the K dimension of
//                 v_mfma_f32_32x32x2_f32, a ones column gives the column sum), on through W_e^T and the post activation
//                 (stored: the only [E, H] tensor the backward pass writes), sums of the input BatchNorm     -> R2
//             B3  gradient through the edge-input block: dW_in | db_in on the MFMA unit  -> R3
// Every pass recomputes the cheap part of the chain (Fourier features, the [H, 2 n_enc + 1] product, e0) from the distance.
// Deterministic: per-lane -> wave tree -> wave order -> block order sums, no atomics.
//
// Round 5: these kernels were instruction-bound, not memory-bound (profiles/r05_n3_*: the pass that reads NO [E, H] tensor took
// 104 us at the QMugs shape; ~2700 instructions per 64 edges in F2, of which 880 LDS reads of weights, 1000 register moves
// feeding packed operands from them and 480 instructions of IEEE division inside 60 SiLUs).  Now: the parameters are packed
// once per direction into a small block in global memory and read through the CONSTANT address space - wave-uniform addresses
// become s_load_dwordx16 and the weights arrive as SGPR pairs that v_pk_fma_f32 takes directly (op_sel splats the lane's
// value over both halves): a 20 x 20 product is 200 packed FMAs and nothing else; no LDS, no moves, no hoisting hack.
// sigmoid = v_rcp_f32(1 + v_exp_f32(-x log2 e)) (1 ulp each; the division was ten instructions).  B2 is the former B2a + B2b
// in one pass: grad_lin never exists as a tensor and e0 is recomputed from the distance instead of gathered from the
// edge-id-ordered d_out (three [E, H] passes less: 2.43 -> 1.3 GB of traffic per backward pass at the QMugs shape).
#include "common.h"
#include "peer.h"

#include <cstdlib>

namespace i3d {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f2 __attribute__((ext_vector_type(2)));
// wave-uniform reads through the constant address space: s_load_dwordx2..x16, the values arrive in SGPRs
typedef const __attribute__((address_space(4))) float* cfp;
typedef const __attribute__((address_space(4))) f2* cf2p;

constexpr int TB = 256;
#ifndef N3_LOOKBACK
#define N3_LOOKBACK 1      // rows of weights a product may have in flight ahead of its FMAs (see `after_all`)
#endif
// The three activations of the stage (edge-input block, the outer one of net3d.py:81, message block) are SiLU in the
// reference's configs (`activation: SiLU` is the constructor default and :81 is hard-coded): compiled in.
constexpr int ACT = I3D_ACT_SILU;
constexpr int odd(int n) { return n | 1; }   // row stride of an MFMA operand tile in LDS: odd -> the per-lane row writes hit
                                             // distinct banks
// >= 3 waves per SIMD (<= 168 registers per lane): the 3D network runs on a stream of its own next to the 2D network - a wave
// that takes the whole register file of its SIMD shuts the other stream out of every CU for the length of the kernel
#define N3_OCC __attribute__((amdgpu_waves_per_eu(3)))
constexpr int MAX_BWD_BLOCKS = 1024; // four blocks per CU; the R kernels add this many partial rows per column

template <int H, int NENC>
struct Dims {
    static constexpr int DIN = NENC > 0 ? 2 * NENC + 1 : 1;
};

// Packed parameter block (floats; every offset even: read as pairs).  Written by n3_pack_kernel once per direction.
template <int H, int DIN>
struct Pk {
    static constexpr int W_IN_T = 0;                 // [DIN][H]     k-major: the pairs of a packed FMA run over the outputs
    static constexpr int B_IN = W_IN_T + DIN * H;    // [H]
    static constexpr int W_E_T = B_IN + H;           // [H k][H o]   forward product lin = c + W_e e0
    static constexpr int W_E = W_E_T + H * H;        // [H o][H k]   backward product W_e^T glin (pairs over k)
    static constexpr int C = W_E + H * H;            // [H]          (W_s + W_d) emb + b_msg
    static constexpr int W_G = C + H;                // [H]
    static constexpr int B_G = W_G + H;              // [1] + pad
    static constexpr int SIZE = B_G + 2;
};
constexpr int PACK_FLOATS = 1280;                    // >= Pk<20, 9>::SIZE = 1042

struct EdgeK {                       // kernel argument (by value)
    int E, N, reduce_mean, rows_per_block;
    int ld_w_in, ld_w_msg;
    float inv_rows;                  // 1 / E
    const float* inv_rows_dev;       // synchronised BatchNorm: 1 / (edges of ALL ranks), on the device; null: inv_rows
    const float* d_raw;
    const int* perm;
    const int* dst_s;
    const int* in_ptr;
    const float* emb;
    const float* W_in;
    const float* b_in;
    const float* W_msg;
    const float* b_msg;
    const float* w_gate;
    const float* b_gate;
    const float* aff_in;             // mean | gamma invstd | beta
    const float* aff_msg;
    const float* invstd_in;
    const float* invstd_msg;
    const float* gsum_in;            // grad_beta | grad_gamma of the input BatchNorm (after R2)   [2H]
    const float* gsum_msg;           // grad_beta | grad_gamma of the message BatchNorm (after R1) [2H]
    float* packed;                   // [PACK_FLOATS] the Pk block
    float* x_msg;
    float* x_center;                 // [H] bf16 storage: x_msg holds bf16(x - x_center) (written by n3_center_kernel, read by everyone)
    float* d_out;
    float* msg;
    const float* grad_m_sum;
    float* grad_ya;
    float* partial;
};

// P: the weights in the layouts of Pk, c = (W_s + W_d) emb + b_msg (every node carries the same embedding, net3d.py:61)
__global__ void __launch_bounds__(256) n3_pack_kernel(EdgeK p, int H, int DIN) {
    float* out = p.packed;
    const int tid = threadIdx.x;
    const int o_bin = DIN * H, o_wet = o_bin + H, o_we = o_wet + H * H, o_c = o_we + H * H, o_wg = o_c + H, o_bg = o_wg + H;
    for (int i = tid; i < DIN * H; i += 256) out[i] = p.W_in[(i % H) * p.ld_w_in + (i / H)];
    for (int i = tid; i < H * H; i += 256) {
        const int k = i / H, o = i - k * H;
        const float v = p.W_msg[(long)o * p.ld_w_msg + 2 * H + k];
        out[o_wet + i] = v;
        out[o_we + o * H + k] = v;
    }
    if (tid < H) {
        out[o_bin + tid] = p.b_in[tid];
        out[o_wg + tid] = p.w_gate[tid];
        float acc = p.b_msg[tid];
        const float* row = p.W_msg + (long)tid * p.ld_w_msg;
        for (int k = 0; k < H; ++k) acc = fmaf(row[k] + row[H + k], p.emb[k], acc);
        out[o_c + tid] = acc;
    }
    if (tid == 0) { out[o_bg] = p.b_gate[0]; out[o_bg + 1] = 0.f; }
}

// The parameter loads are wave-uniform, loop-invariant reads of constant memory: left alone the compiler hoists ALL of them
// (~1500 SGPRs' worth) out of the per-edge loop and spills them to VGPR lanes (v_writelane / v_readlane by the thousand).
// Passing the base pointer through an empty volatile asm once per edge makes its loads belong to that iteration: they are issued
// as s_load_dwordx16 bursts next to their use and the scalar cache serves them.
template <typename T>
__device__ __forceinline__ T per_edge(T ptr) {
    __asm__ volatile("" : "+s"(ptr));
    return ptr;
}
// ... and tied to a value of the lane: the loads through the returned pointer cannot be issued before `dep` exists, so the
// scheduler cannot gather the parameter loads of ALL stages of an edge at its top either (the fused backward pass reads ~1600
// parameter words per edge: gathered, they spill again)
template <typename T>
__device__ __forceinline__ T after(T ptr, float dep) {
    __asm__ volatile("" : "+s"(ptr) : "v"(dep));
    return ptr;
}
// ... tied to EVERY accumulator pair of a product (tied to one, the scheduler runs that accumulator's chain through all rows
// first and parks each row's weights in VGPR lanes for the other nine)
template <int NP, typename T>
__device__ __forceinline__ T after_all(T ptr, const f2* a) {
    static_assert(NP == 10 || NP == 8, "hidden 20 or 16");
    if constexpr (NP == 10)
        __asm__ volatile("" : "+s"(ptr) : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]), "v"(a[8]), "v"(a[9]));
    else
        __asm__ volatile("" : "+s"(ptr) : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]));
    return ptr;
}
__device__ __forceinline__ f2 splat(float x) { return f2{x, x}; }
__device__ __forceinline__ f2 fma2(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ float comp(const f2* v, int c) { return (c & 1) ? v[c >> 1].y : v[c >> 1].x; }
// sigmoid with the hardware's reciprocal and exponential (1 ulp each; x / (1 + e) as an IEEE division is ten instructions,
// and a pass has up to sixty of them per edge)
__device__ __forceinline__ float sigm(float x) { return __builtin_amdgcn_rcpf(1.f + __expf(-x)); }
__device__ __forceinline__ f2 sigm2(f2 x) { return f2{sigm(x.x), sigm(x.y)}; }
__device__ __forceinline__ f2 silu2(f2 x) { return x * sigm2(x); }
__device__ __forceinline__ f2 silu_grad2(f2 x) {          // d/dx x sigmoid(x) = s (1 + x (1 - s))
    const f2 s = sigm2(x);
    return s * (1.f + x * (1.f - s));
}

// reference commons/utils.py:103-110: [sin(d / 2^k)]_k | [cos(d / 2^k)]_k | d.  ONE accurate sincos of the smallest angle
// d / 2^(n-1), the others by angle doubling (sin 2t = 2 sin t cos t, cos 2t = 1 - 2 sin^2 t): each doubling at most doubles
// the absolute error, 2^(n-1) ulp = 5e-7 at n = 4 - the passes that need the features recompute them from the distance
template <int NENC>
__device__ __forceinline__ void fourier(float x, float* f) {
    if constexpr (NENC == 0) {
        f[0] = x;
    } else {
        float sn, cs;
        sincosf(x / (float)(1 << (NENC - 1)), &sn, &cs);
        f[NENC - 1] = sn;
        f[2 * NENC - 1] = cs;
#pragma unroll
        for (int k = NENC - 2; k >= 0; --k) {
            const float s2 = 2.f * sn * cs, c2 = fmaf(-2.f * sn, sn, 1.f);
            sn = s2; cs = c2;
            f[k] = sn;
            f[NENC + k] = cs;
        }
        f[2 * NENC] = x;
    }
}

// The products walk the weights row by row (H / 2 SGPR pairs per row).  Loads of constant memory carry no memory dependence:
// instruction selection floats all 25 s_load_dwordx16 of a product (400 SGPRs) in front of its first FMA and the register
// allocator spills them to VGPR lanes (a scheduling barrier does not hold them: they are placed before it exists).  So row k's
// pointer is tied (`after_all`) to the accumulators as row k - 1 left them: a row of weights is loaded, used by its H / 2
// packed FMAs and dead; the other waves of the SIMD cover the scalar cache's latency.

// a = W_in f + b_in   (per output: b + sum over k in k order)
template <int H, int DIN>
__device__ __forceinline__ void lin_in(cfp P, const float* f, f2* a) {
    const cf2p b = (cf2p)(P + Pk<H, DIN>::B_IN);
#pragma unroll
    for (int o = 0; o < H / 2; ++o) a[o] = b[o];
#pragma unroll
    for (int k = 0; k < DIN; ++k) {
        const cf2p wt = (cf2p)(after_all<H / 2>(P, a) + Pk<H, DIN>::W_IN_T + k * H);
        const f2 fk = splat(f[k]);
#pragma unroll
        for (int o = 0; o < H / 2; ++o) a[o] = fma2(wt[o], fk, a[o]);
    }
}

// y = (x - mean) * (gamma invstd) + beta,  aff = mean | gamma invstd | beta
template <int H>
__device__ __forceinline__ void bn_apply(cfp aff, const f2* x, f2* y) {
    const cf2p mu = (cf2p)aff, sc = (cf2p)(aff + H), sh = (cf2p)(aff + 2 * H);
#pragma unroll
    for (int o = 0; o < H / 2; ++o) y[o] = (x[o] - mu[o]) * sc[o] + sh[o];
}

// lin = c + W_e e0
template <int H, int DIN>
__device__ __forceinline__ void lin_msg(cfp P, const f2* e0, f2* lin) {
    const cf2p c = (cf2p)(P + Pk<H, DIN>::C);
#pragma unroll
    for (int o = 0; o < H / 2; ++o) lin[o] = c[o];
    f2 prev[H / 2];                          // the accumulators as the row before last left them (N3_LOOKBACK 2)
#pragma unroll
    for (int o = 0; o < H / 2; ++o) prev[o] = lin[o];
#pragma unroll
    for (int k = 0; k < H; ++k) {
        const cf2p wt = (cf2p)(after_all<H / 2>(P, N3_LOOKBACK == 2 ? prev : lin) + Pk<H, DIN>::W_E_T + k * H);
        if (N3_LOOKBACK == 2) {
#pragma unroll
            for (int o = 0; o < H / 2; ++o) prev[o] = lin[o];
        }
        const f2 ek = splat(comp(e0, k));
#pragma unroll
        for (int o = 0; o < H / 2; ++o) lin[o] = fma2(wt[o], ek, lin[o]);
    }
}

// ge = W_e^T glin
template <int H, int DIN>
__device__ __forceinline__ void lin_msg_t(cfp P, const f2* glin, f2* ge) {
#pragma unroll
    for (int k = 0; k < H / 2; ++k) ge[k] = splat(0.f);
    f2 prev[H / 2];
#pragma unroll
    for (int k = 0; k < H / 2; ++k) prev[k] = glin[k];
#pragma unroll
    for (int o = 0; o < H; ++o) {
        const cf2p w = (cf2p)(after_all<H / 2>(P, N3_LOOKBACK == 2 ? prev : (o == 0 ? glin : ge)) + Pk<H, DIN>::W_E + o * H);
        if (N3_LOOKBACK == 2) {
#pragma unroll
            for (int k = 0; k < H / 2; ++k) prev[k] = o == 0 ? glin[k] : ge[k];
        }
        const f2 go = splat(comp(glin, o));
#pragma unroll
        for (int k = 0; k < H / 2; ++k) ge[k] = fma2(w[k], go, ge[k]);
    }
}

template <int H>
__device__ __forceinline__ void load_row(const float* p, f2* v) {
#pragma unroll
    for (int c = 0; c < H; c += 4) {
        const float4 t = *reinterpret_cast<const float4*>(p + c);
        v[c / 2] = f2{t.x, t.y};
        v[c / 2 + 1] = f2{t.z, t.w};
    }
}

template <int H>
__device__ __forceinline__ void store_row(float* p, const f2* v) {
#pragma unroll
    for (int c = 0; c < H; c += 4) *reinterpret_cast<float4*>(p + c) = make_float4(v[c / 2].x, v[c / 2].y, v[c / 2 + 1].x, v[c / 2 + 1].y);
}

// The [E, H] ACTIVATIONS that only this stage reads and writes - x_msg (saved) and msg - in the storage type of
// the launch: fp32, or (bf16 matmul mode, I3dNet3dEdgeArgs.store_bf16) bf16 - half the bytes of the passes that are bound by
// them (at the QMugs shape an [E3, 20] fp32 tensor is 313 MB).  Row `row` of a buffer that was sized for fp32; values are
// rounded (RNE) once, where they are stored; statistics are taken before the rounding.  ctr: the centre x_msg is stored about.
template <int H, bool B16>
__device__ __forceinline__ void load_srow(const float* base, long row, f2* v, cfp ctr = nullptr) {
    if constexpr (!B16) {
        load_row<H>(base + row * H, v);
    } else {
        const uint2* q = reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(base) + row * H);
#pragma unroll
        for (int c = 0; c < H; c += 4) {
            const uint2 t = q[c / 4];
            v[c / 2] = f2{__uint_as_float(t.x << 16), __uint_as_float(t.x & 0xffff0000u)};
            v[c / 2 + 1] = f2{__uint_as_float(t.y << 16), __uint_as_float(t.y & 0xffff0000u)};
        }
        if (ctr != nullptr) {
            const cf2p c2 = (cf2p)ctr;
#pragma unroll
            for (int o = 0; o < H / 2; ++o) v[o] += c2[o];
        }
    }
}

__device__ __forceinline__ unsigned bf16_rne(float x) {      // bits of bf16(x), round to nearest even (finite inputs)
    const unsigned u = __float_as_uint(x);
    return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}

template <int H, bool B16>
__device__ __forceinline__ void store_srow(float* base, long row, const f2* v, cfp ctr = nullptr) {
    if constexpr (!B16) {
        store_row<H>(base + row * H, v);
    } else {
        uint2* q = reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(base) + row * H);
        f2 d[H / 2];
        const cf2p c2 = (cf2p)ctr;
#pragma unroll
        for (int o = 0; o < H / 2; ++o) d[o] = ctr != nullptr ? v[o] - c2[o] : v[o];
#pragma unroll
        for (int c = 0; c < H; c += 4)
            q[c / 4] = make_uint2(bf16_rne(d[c / 2].x) | (bf16_rne(d[c / 2].y) << 16),
                                  bf16_rne(d[c / 2 + 1].x) | (bf16_rne(d[c / 2 + 1].y) << 16));
    }
}

// column sums over the block: on return red[w * 2 NP + c] holds wave w's sum of column c of `v` (NP pairs = 2 NP columns; 4 waves)
template <int NP>
__device__ __forceinline__ void block_sum_pairs(f2* v, float* red, int tid) {
#pragma unroll
    for (int c = 0; c < NP; ++c) {
        float x = v[c].x, y = v[c].y;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) { x += __shfl_xor(x, off); y += __shfl_xor(y, off); }
        v[c] = f2{x, y};
    }
    if ((tid & 63) == 0) {
#pragma unroll
        for (int c = 0; c < NP; ++c) { red[(tid >> 6) * 2 * NP + 2 * c] = v[c].x; red[(tid >> 6) * 2 * NP + 2 * c + 1] = v[c].y; }
    }
}

// ---- statistics of a tile from sums about a pivot (a sample of the tile: no cancellation whatever mean / std of the column)
// red: wave w's sums at red[w 2H + c] (shifted sum of column c) and red[w 2H + H + c] (shifted sum of squares)
template <int H>
__device__ __forceinline__ void write_tile_partial(float* partial, int tile, const float* red, const float* pivot, float n,
                                                   int tid) {
    if (tid < H) {
        const float s1 = red[tid] + red[2 * H + tid] + red[4 * H + tid] + red[6 * H + tid];
        const float s2 = red[H + tid] + red[3 * H + tid] + red[5 * H + tid] + red[7 * H + tid];
        float* o = partial + (long)tile * 3 * H;
        o[tid] = fmaf(n, pivot[tid], s1);
        o[H + tid] = fmaxf(s2 - s1 * s1 / n, 0.f);
        o[2 * H + tid] = n;
    }
}

// shifted sums of one row: s[0..H/2) += x - pv,  s[H/2..H) += (x - pv)^2
template <int H>
__device__ __forceinline__ void stat_row(const f2* x, const f2* pv, f2* s) {
#pragma unroll
    for (int o = 0; o < H / 2; ++o) {
        const f2 t = x[o] - pv[o];
        s[o] += t;
        s[H / 2 + o] = fma2(t, t, s[H / 2 + o]);
    }
}

// the pivot of a tile: the row of the tile's first edge (thread 0), broadcast through LDS
template <int H>
__device__ __forceinline__ void share_pivot(float* pivot, const f2* row, f2* pv, int tid) {
    if (tid == 0) {
#pragma unroll
        for (int o = 0; o < H / 2; ++o) { pivot[2 * o] = row[o].x; pivot[2 * o + 1] = row[o].y; }
    }
    __syncthreads();
#pragma unroll
    for (int o = 0; o < H / 2; ++o) pv[o] = f2{pivot[2 * o], pivot[2 * o + 1]};
}

// F1: statistics of xa = act(W_in f + b_in)
template <int H, int NENC, bool B16>
__global__ void __launch_bounds__(TB) N3_OCC n3_stats_in_kernel(EdgeK p) {
    constexpr int DIN = Dims<H, NENC>::DIN;
    __shared__ float pivot[H];
    __shared__ float red[4 * 2 * H];
    const cfp P = (cfp)p.packed;
    const int tid = threadIdx.x;
    const long j0 = (long)blockIdx.x * p.rows_per_block;
    const long jend = min(j0 + p.rows_per_block, (long)p.E);
    f2 xa[H / 2], s[H], pv[H / 2];
#pragma unroll
    for (int c = 0; c < H; ++c) s[c] = splat(0.f);
#pragma unroll
    for (int c = 0; c < H / 2; ++c) xa[c] = splat(0.f);
    auto edge = [&](float d) {
        float f[DIN];
        f2 a[H / 2];
        fourier<NENC>(d, f);
        lin_in<H, DIN>(after(P, f[0]), f, a);
#pragma unroll
        for (int o = 0; o < H / 2; ++o) xa[o] = silu2(a[o]);
    };
    long j = j0 + tid;
    const bool first = j < jend;
    float dn = (j + TB < jend) ? p.d_raw[p.perm[j + TB]] : 0.f;       // the next edge's distance is in flight during this one's arithmetic
    if (first) edge(p.d_raw[p.perm[j]]);
    share_pivot<H>(pivot, xa, pv, tid);
    if (first) stat_row<H>(xa, pv, s);
    for (j += TB; j < jend; j += TB) {
        const float d = dn;
        dn = (j + TB < jend) ? p.d_raw[p.perm[j + TB]] : 0.f;
        edge(d);
        stat_row<H>(xa, pv, s);
    }
    block_sum_pairs<H>(s, red, tid);
    __syncthreads();
    write_tile_partial<H>(p.partial, blockIdx.x, red, pivot, (float)(jend - j0), tid);
}

// e0 = act( BN_in( act( W_in f + b_in ) ) ) of one distance; xa / ya (the BatchNorm's input / output) for the callers that differentiate
template <int H, int NENC>
__device__ __forceinline__ void edge_input(cfp P, cfp aff_in, float d, float* f, f2* a, f2* xa, f2* ya, f2* e0) {
    constexpr int DIN = Dims<H, NENC>::DIN;
    fourier<NENC>(d, f);
    lin_in<H, DIN>(after(P, f[0]), f, a);
#pragma unroll
    for (int o = 0; o < H / 2; ++o) xa[o] = silu2(a[o]);
    bn_apply<H>(after(aff_in, xa[0].x), xa, ya);
#pragma unroll
    for (int o = 0; o < H / 2; ++o) e0[o] = silu2(ya[o]);
}

// bf16 storage: the centre x_msg is stored about - the column means of the message block's activation over the first TB edges.
// A pre-BatchNorm activation must not be rounded as it is: its columns can be nearly constant over the batch (every edge
// feature is a function of ONE scalar distance), the BatchNorm behind then divides a 2^-9 |x| rounding error by a standard
// deviation that is smaller than that (measured with raw storage: 10-96 % error on the stage's parameter gradients at 32
// molecules).  About a centre inside the distribution the rounding error is 2^-9 of the DEVIATION - what the BatchNorm scales.
template <int H, int NENC, bool B16>
__global__ void __launch_bounds__(TB) N3_OCC n3_center_kernel(EdgeK p) {
    constexpr int DIN = Dims<H, NENC>::DIN;
    __shared__ float red[4 * H];
    const cfp P = (cfp)p.packed;
    const int tid = threadIdx.x;
    f2 xm[H / 2];
    const int n = p.E < TB ? p.E : TB;
    if (tid < n) {
        float f[DIN];
        f2 a[H / 2], xa[H / 2], ya[H / 2], e0[H / 2], lin[H / 2];
        edge_input<H, NENC>(P, (cfp)p.aff_in, p.d_raw[p.perm[tid]], f, a, xa, ya, e0);
        lin_msg<H, DIN>(after(P, e0[0].x), e0, lin);
#pragma unroll
        for (int o = 0; o < H / 2; ++o) xm[o] = silu2(lin[o]);
    } else {
#pragma unroll
        for (int o = 0; o < H / 2; ++o) xm[o] = splat(0.f);
    }
    block_sum_pairs<H / 2>(xm, red, tid);
    __syncthreads();
    if (tid < H) p.x_center[tid] = (((red[tid] + red[H + tid]) + red[2 * H + tid]) + red[3 * H + tid]) / (float)n;
}

// F2: e0 -> d_out (edge-id order), x_msg = act(c + W_e e0) (stored, destination-sorted) and its statistics
template <int H, int NENC, bool B16>
__global__ void __launch_bounds__(TB) N3_OCC n3_msg_pre_kernel(EdgeK p) {
    constexpr int DIN = Dims<H, NENC>::DIN;
    __shared__ float pivot[H];
    __shared__ float red[4 * 2 * H];
    const cfp P = (cfp)p.packed;
    const cfp ctr = B16 ? (cfp)p.x_center : nullptr;
    const int tid = threadIdx.x;
    const long j0 = (long)blockIdx.x * p.rows_per_block;
    const long jend = min(j0 + p.rows_per_block, (long)p.E);
    f2 xm[H / 2], s[H], pv[H / 2];
#pragma unroll
    for (int c = 0; c < H; ++c) s[c] = splat(0.f);
#pragma unroll
    for (int c = 0; c < H / 2; ++c) xm[c] = splat(0.f);
    auto edge = [&](long jj, int eid, float d) {
        float f[DIN];
        f2 a[H / 2], xa[H / 2], ya[H / 2], e0[H / 2], lin[H / 2];
        edge_input<H, NENC>(P, (cfp)p.aff_in, d, f, a, xa, ya, e0);
        store_row<H>(p.d_out + (long)eid * H, e0);
        lin_msg<H, DIN>(after(P, e0[0].x), e0, lin);
#pragma unroll
        for (int o = 0; o < H / 2; ++o) xm[o] = silu2(lin[o]);
        store_srow<H, B16>(p.x_msg, jj, xm, B16 ? after(ctr, xm[0].x) : nullptr);
    };
    long j = j0 + tid;
    const bool first = j < jend;
    int en = 0;
    float dn = 0.f;
    if (j + TB < jend) { en = p.perm[j + TB]; dn = p.d_raw[en]; }
    if (first) {
        const int eid = p.perm[j];
        edge(j, eid, p.d_raw[eid]);
    }
    share_pivot<H>(pivot, xm, pv, tid);
    if (first) stat_row<H>(xm, pv, s);
    for (j += TB; j < jend; j += TB) {
        const int eid = en;
        const float d = dn;
        if (j + TB < jend) { en = p.perm[j + TB]; dn = p.d_raw[en]; }
        edge(j, eid, d);
        stat_row<H>(xm, pv, s);
    }
    block_sum_pairs<H>(s, red, tid);
    __syncthreads();
    write_tile_partial<H>(p.partial, blockIdx.x, red, pivot, (float)(jend - j0), tid);
}

// the gate: w = sigmoid(w_g . m + b_g)
template <int H, int DIN>
__device__ __forceinline__ float gate_of(cfp P, const f2* m) {
    const cf2p wg = (cf2p)(P + Pk<H, DIN>::W_G);
    f2 acc = f2{P[Pk<H, DIN>::B_G], 0.f};
#pragma unroll
    for (int o = 0; o < H / 2; ++o) acc = fma2(m[o], wg[o], acc);
    return sigm(acc.x + acc.y);
}

// F3: msg = m * gate, m = BN_msg(x_msg)
template <int H, int NENC, bool B16>
__global__ void __launch_bounds__(TB) N3_OCC n3_gate_kernel(EdgeK p) {
    constexpr int DIN = Dims<H, NENC>::DIN;
    const cfp P = (cfp)p.packed;
    const long j = (long)blockIdx.x * TB + threadIdx.x;
    if (j >= p.E) return;
    f2 xm[H / 2], m[H / 2];
    load_srow<H, B16>(p.x_msg, j, xm, B16 ? (cfp)p.x_center : nullptr);
    bn_apply<H>((cfp)p.aff_msg, xm, m);
    const float g = gate_of<H, DIN>(P, m);
#pragma unroll
    for (int o = 0; o < H / 2; ++o) m[o] *= g;
    store_srow<H, B16>(p.msg, j, m);
}

// gradient reaching m of edge j:  gm = gmsg * g + gg * w_g,  gg = (gmsg . m) g (1 - g),  gmsg = grad_m_sum[dst] (/ deg)
template <int H, int DIN>
__device__ __forceinline__ void grad_m(cfp P, const EdgeK& p, long j, const f2* m, f2* gm, float& gg) {
    const int v = p.dst_s[j];
    float sc = 1.f;
    if (p.reduce_mean) sc = 1.f / (float)(p.in_ptr[v + 1] - p.in_ptr[v]);
    f2 gmsg[H / 2];
    load_row<H>(p.grad_m_sum + (long)v * H, gmsg);
    const float g = gate_of<H, DIN>(P, m);
    f2 dot = splat(0.f);
#pragma unroll
    for (int o = 0; o < H / 2; ++o) { gmsg[o] *= sc; dot = fma2(gmsg[o], m[o], dot); }
    gg = (dot.x + dot.y) * g * (1.f - g);
    const cf2p wg = (cf2p)(P + Pk<H, DIN>::W_G);
#pragma unroll
    for (int o = 0; o < H / 2; ++o) gm[o] = gmsg[o] * g + gg * wg[o];
}

// B1: partial[block] = sum gm | sum gm xhat_m | sum gg m | sum gg
template <int H, int NENC, bool B16>
__global__ void __launch_bounds__(TB) N3_OCC n3_bwd_sums_kernel(EdgeK p) {
    constexpr int DIN = Dims<H, NENC>::DIN;
    constexpr int NC = 3 * H + 1, NPR = 3 * H / 2 + 1;      // columns; pairs (the last pair: sum gg | 0)
    __shared__ float red[4 * 2 * NPR];
    const cfp P = (cfp)p.packed;
    const cfp ctr = B16 ? (cfp)p.x_center : nullptr;
    const int tid = threadIdx.x;
    const long j0 = (long)blockIdx.x * p.rows_per_block;
    const long jend = min(j0 + p.rows_per_block, (long)p.E);
    f2 s[NPR];
#pragma unroll
    for (int c = 0; c < NPR; ++c) s[c] = splat(0.f);
    for (long j = j0 + tid; j < jend; j += TB) {
        f2 xm[H / 2], m[H / 2], gm[H / 2];
        float gg;
        load_srow<H, B16>(p.x_msg, j, xm, B16 ? per_edge(ctr) : nullptr);
        const cfp aff = after((cfp)p.aff_msg, xm[0].x);
        bn_apply<H>(aff, xm, m);
        grad_m<H, DIN>(after(P, m[0].x), p, j, m, gm, gg);
        const cf2p mu = (cf2p)after(aff, gm[0].x), istd = (cf2p)after((cfp)p.invstd_msg, gm[0].x);
#pragma unroll
        for (int o = 0; o < H / 2; ++o) {
            const f2 xh = (xm[o] - mu[o]) * istd[o];
            s[o] += gm[o];
            s[H / 2 + o] = fma2(gm[o], xh, s[H / 2 + o]);
            s[H + o] = fma2(splat(gg), m[o], s[H + o]);
        }
        s[3 * H / 2].x += gg;
    }
    block_sum_pairs<NPR>(s, red, tid);
    __syncthreads();
    if (tid < NC) p.partial[(long)blockIdx.x * NC + tid] = red[tid] + red[2 * NPR + tid] + red[4 * NPR + tid] + red[6 * NPR + tid];
}

// D[i][j] += sum over the wave's 64 edges of A[edge][i] * B[edge][j]: v_mfma_f32_32x32x2_f32, the edge pair (2 s, 2 s + 1)
// is the K dimension of step s.  tile_a / tile_b: this wave's [64][odd(NA)] / [64][odd(NB)] tiles; the lanes of the
// padding columns (>= NA / NB) feed zeros.  a: NA values as pairs; b: NB values as floats.
template <int NA, int NB>
__device__ __forceinline__ void outer_accumulate(float* tile_a, float* tile_b, const f2* a, const float* b, bool valid, int lane,
                                                 f32x16& acc) {
    constexpr int LDA = odd(NA), LDB = odd(NB);
    // The tiles are the WAVE's own: a wave's LDS operations execute in issue order, so its writes below are complete before its
    // reads, and those before the next trip's writes - no workgroup barrier (round 2 had two per trip: four waves in lock step,
    // nobody's VALU work under anybody's MFMAs).  The wave barriers only pin the order for the compiler.
#pragma unroll
    for (int c = 0; c < NA; ++c) tile_a[lane * LDA + c] = valid ? comp(a, c) : 0.f;
#pragma unroll
    for (int c = 0; c < NB; ++c) tile_b[lane * LDB + c] = valid ? b[c] : 0.f;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const int row = lane >> 5, col = lane & 31;
    const int ca = col < NA ? col : 0, cb = col < NB ? col : 0;
#pragma unroll 8
    for (int s = 0; s < 32; ++s) {
        float av = tile_a[(2 * s + row) * LDA + ca];
        float bv = tile_b[(2 * s + row) * LDB + cb];
        av = col < NA ? av : 0.f;
        bv = col < NB ? bv : 0.f;
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc, 0, 0, 0);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <int NA, int NB>
constexpr int tile_floats() { return 4 * 64 * (odd(NA) + odd(NB)) > 4096 ? 4 * 64 * (odd(NA) + odd(NB)) : 4096; }

// the four waves' accumulators -> partial[NA * NB] of the block.  scratch: >= 4 * 1024 floats of LDS
template <int NA, int NB>
__device__ __forceinline__ void write_outer(const f32x16& acc, float* scratch, float* out, int tid) {
    const int lane = tid & 63, wv = tid >> 6;
    __syncthreads();          // every wave is done with its operand tiles (the scratch overlays them)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int i = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        scratch[wv * 1024 + i * 32 + (lane & 31)] = acc[r];
    }
    __syncthreads();
    for (int t = tid; t < NA * NB; t += TB) {
        const int i = t / NB, jx = t - i * NB;
        const int o = i * 32 + jx;
        out[t] = (scratch[o] + scratch[1024 + o]) + (scratch[2048 + o] + scratch[3072 + o]);
    }
}

// B2: gradient through the message block and on through W_e^T and the post activation of the edge-input block:
//     glin = BN_msg'(gm) act'(lin)            (registers only)
//     partial[block][H][H + 1] = sum glin (x) [e0 | 1]         (dW_e | dc, MFMA)
//     gya  = (W_e^T glin) act'(ya)            (stored)
//     partial[block][H (H + 1) ...] = sum gya [H] | sum gya xhat_a [H]
// e0, ya, xa are recomputed from the distance (a few hundred instructions against a gathered [E, H] read of d_out).
template <int H, int NENC, bool B16, int OCC>
__global__ void __launch_bounds__(TB) __attribute__((amdgpu_waves_per_eu(OCC))) n3_bwd_msg_kernel(EdgeK p) {
    constexpr int DIN = Dims<H, NENC>::DIN;
    constexpr int NB = H + 1, NP = H * NB + 2 * H;
    __shared__ float tiles[tile_floats<H, NB>()];      // also the accumulator exchange (4 * 1024 floats) at the end
    __shared__ float red[4 * 2 * H];
    const cfp P = (cfp)p.packed;
    const cfp ctr = B16 ? (cfp)p.x_center : nullptr;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const float inv_rows = p.inv_rows_dev != nullptr ? p.inv_rows_dev[0] : p.inv_rows;
    float* tile_a = tiles + wv * 64 * (odd(H) + odd(NB));
    float* tile_b = tile_a + 64 * odd(H);
    const long j0 = (long)blockIdx.x * p.rows_per_block;
    const long jend = min(j0 + p.rows_per_block, (long)p.E);
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    f2 s[H];
#pragma unroll
    for (int c = 0; c < H; ++c) s[c] = splat(0.f);
    float dn = (j0 + tid < jend) ? p.d_raw[p.perm[j0 + tid]] : 0.f;
    for (long base = j0; base < jend; base += TB) {
        const long j = base + tid;
        const bool valid = j < jend;
        const float d = dn;
        dn = (j + TB < jend) ? p.d_raw[p.perm[j + TB]] : 0.f;
        f2 glin[H / 2];
        float e0x[NB];
#pragma unroll
        for (int c = 0; c < H / 2; ++c) glin[c] = splat(0.f);
#pragma unroll
        for (int c = 0; c < H; ++c) e0x[c] = 0.f;
        e0x[H] = 1.f;
        if (valid) {
            float f[DIN];
            f2 a[H / 2], xa[H / 2], ya[H / 2], e0[H / 2];
            edge_input<H, NENC>(P, (cfp)p.aff_in, d, f, a, xa, ya, e0);
            f2 lin[H / 2];
            lin_msg<H, DIN>(after(P, e0[0].x), e0, lin);
#pragma unroll
            for (int c = 0; c < H; ++c) e0x[c] = comp(e0, c);
            {
                f2 xm[H / 2], m[H / 2], gm[H / 2];
                float gg;
                load_srow<H, B16>(p.x_msg, j, xm, B16 ? after(ctr, lin[0].x) : nullptr);
                const cfp aff_m = after((cfp)p.aff_msg, xm[0].x);
                bn_apply<H>(aff_m, xm, m);
                grad_m<H, DIN>(after(P, m[0].x), p, j, m, gm, gg);
                const cfp aff_m2 = after(aff_m, gm[0].x), gsm = after((cfp)p.gsum_msg, gm[0].x);
                const cf2p mu_m = (cf2p)aff_m2, sc_m = (cf2p)(aff_m2 + H), istd_m = (cf2p)after((cfp)p.invstd_msg, gm[0].x);
                const cf2p gs_m = (cf2p)gsm, gs_m2 = (cf2p)(gsm + H);
#pragma unroll
                for (int o = 0; o < H / 2; ++o) {
                    const f2 xh = (xm[o] - mu_m[o]) * istd_m[o];
                    const f2 gx = sc_m[o] * (gm[o] - gs_m[o] * inv_rows - xh * (gs_m2[o] * inv_rows));
                    glin[o] = gx * silu_grad2(lin[o]);
                }
            }
            f2 gya[H / 2];
            lin_msg_t<H, DIN>(after(P, glin[0].x), glin, gya);
            const cf2p mu_a = (cf2p)after((cfp)p.aff_in, gya[0].x), istd_a = (cf2p)after((cfp)p.invstd_in, gya[0].x);
#pragma unroll
            for (int o = 0; o < H / 2; ++o) {
                gya[o] *= silu_grad2(ya[o]);
                const f2 xh = (xa[o] - mu_a[o]) * istd_a[o];
                s[o] += gya[o];
                s[H / 2 + o] = fma2(gya[o], xh, s[H / 2 + o]);
            }
            store_row<H>(p.grad_ya + j * H, gya);
        }
        outer_accumulate<H, NB>(tile_a, tile_b, glin, e0x, valid, lane, acc);
    }
    block_sum_pairs<H>(s, red, tid);
    write_outer<H, NB>(acc, tiles, p.partial + (long)blockIdx.x * NP, tid);      // (its barrier also publishes `red`)
    if (tid < 2 * H)
        p.partial[(long)blockIdx.x * NP + H * NB + tid] = red[tid] + red[2 * H + tid] + red[4 * H + tid] + red[6 * H + tid];
}

// B3: gradient through the edge-input block: partial[block] = [H][DIN + 1] (dW_in | db_in)
template <int H, int NENC, bool B16>
__global__ void __launch_bounds__(TB) N3_OCC n3_bwd_in_kernel(EdgeK p) {
    constexpr int DIN = Dims<H, NENC>::DIN;
    constexpr int NB = DIN + 1, NP = H * NB;
    __shared__ float tiles[tile_floats<H, NB>()];
    const cfp P = (cfp)p.packed;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const float inv_rows = p.inv_rows_dev != nullptr ? p.inv_rows_dev[0] : p.inv_rows;
    float* tile_a = tiles + wv * 64 * (odd(H) + odd(NB));
    float* tile_b = tile_a + 64 * odd(H);
    const long j0 = (long)blockIdx.x * p.rows_per_block;
    const long jend = min(j0 + p.rows_per_block, (long)p.E);
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    float dn = (j0 + tid < jend) ? p.d_raw[p.perm[j0 + tid]] : 0.f;
    for (long base = j0; base < jend; base += TB) {
        const long j = base + tid;
        const bool valid = j < jend;
        const float d = dn;
        dn = (j + TB < jend) ? p.d_raw[p.perm[j + TB]] : 0.f;
        f2 ga[H / 2];
        float fx[NB];
#pragma unroll
        for (int c = 0; c < H / 2; ++c) ga[c] = splat(0.f);
#pragma unroll
        for (int c = 0; c < DIN; ++c) fx[c] = 0.f;
        fx[DIN] = 1.f;
        if (valid) {
            f2 a[H / 2], gya[H / 2];
            load_row<H>(p.grad_ya + j * H, gya);
            fourier<NENC>(d, fx);
            lin_in<H, DIN>(after(P, fx[0]), fx, a);
            const cfp aff_a = after((cfp)p.aff_in, a[0].x), gsa = after((cfp)p.gsum_in, a[0].x);
            const cf2p mu_a = (cf2p)aff_a, sc_a = (cf2p)(aff_a + H), istd_a = (cf2p)after((cfp)p.invstd_in, a[0].x);
            const cf2p gs_a = (cf2p)gsa, gs_a2 = (cf2p)(gsa + H);
#pragma unroll
            for (int o = 0; o < H / 2; ++o) {
                const f2 xa = silu2(a[o]);
                const f2 xh = (xa - mu_a[o]) * istd_a[o];
                const f2 gx = sc_a[o] * (gya[o] - gs_a[o] * inv_rows - xh * (gs_a2[o] * inv_rows));
                ga[o] = gx * silu_grad2(a[o]);
            }
        }
        outer_accumulate<H, NB>(tile_a, tile_b, ga, fx, valid, lane, acc);
    }
    write_outer<H, NB>(acc, tiles, p.partial + (long)blockIdx.x * NP, tid);
}

// ---- R kernels: the partial rows of a pass (block order) -> the parameter gradients.  One workgroup per 32 columns: its 1024
// threads form 32 row groups x 32 columns, group g adds rows g, g + 32, ... (two alternating accumulators, eight loads in
// flight), the groups are added in group order: a fixed order whatever the grid.  Every column's owner writes what the column
// means; the few outputs that need SEVERAL columns (d emb from all of dc) are formed by workgroup 0, which adds those columns
// once more (the same order: the same bits).  Round 4's one-workgroup form needed 67 us for 0.9 MB at the QMugs shape.
struct ReduceK {
    int H, DIN, n_rows, ld_w_in, ld_w_msg;
    const float* partial;
    const float* emb;
    const float* W_msg;
    float* gsum;                      // [2H] grad_beta | grad_gamma handed to the next pass
    float* grad_gamma;
    float* grad_beta;
    float* grad_w_gate;
    float* grad_b_gate;
    float* grad_W_msg;
    float* grad_b_msg;
    float* grad_emb;
    float* grad_W_in;
    float* grad_b_in;
};

// sum over the rows of column `col` (col < 0: nothing) for the thread's row group -> the column total in tot[c0] (c0 = tid & 31)
__device__ __forceinline__ void reduce_col(const float* partial, int n_rows, int n_cols, int col, float* scratch, float* tot, int tid) {
    const int c0 = tid & 31, grp = tid >> 5;
    float a0 = 0.f, a1 = 0.f;
    if (col >= 0) {
        const float* q = partial + col;
        int r = grp;
        for (; r + 15 * 32 < n_rows; r += 16 * 32) {      // sixteen rows in flight (a row is a page apart from the next)
            float v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) v[u] = q[(long)(r + u * 32) * n_cols];
#pragma unroll
            for (int u = 0; u < 16; u += 2) { a0 += v[u]; a1 += v[u + 1]; }
        }
        for (; r + 32 < n_rows; r += 2 * 32) {
            a0 += q[(long)r * n_cols];
            a1 += q[(long)(r + 32) * n_cols];
        }
        if (r < n_rows) a0 += q[(long)r * n_cols];
    }
    scratch[tid] = a0 + a1;
    __syncthreads();
    if (grp == 0) {
        float t = 0.f;
        for (int g = 0; g < 32; ++g) t += scratch[g * 32 + c0];
        tot[c0] = t;
    }
    __syncthreads();
}

__global__ void __launch_bounds__(1024) n3_reduce_sums_kernel(ReduceK q) {      // R1: 3H + 1 columns
    __shared__ float tot[32];
    __shared__ float scratch[1024];
    const int H = q.H, tid = threadIdx.x, n_cols = 3 * H + 1;
    const int col = blockIdx.x * 32 + (tid & 31);
    reduce_col(q.partial, q.n_rows, n_cols, col < n_cols ? col : -1, scratch, tot, tid);
    if (tid < 32 && col < n_cols) {
        const float t = tot[tid];
        if (col < H) { q.grad_beta[col] = t; q.gsum[col] = t; }
        else if (col < 2 * H) { q.grad_gamma[col - H] = t; q.gsum[col] = t; }
        else if (col < 3 * H) q.grad_w_gate[col - 2 * H] = t;
        else q.grad_b_gate[0] = t;
    }
}

__global__ void __launch_bounds__(1024) n3_reduce_msg_kernel(ReduceK q) {       // R2: H (H + 1) + 2H columns
    __shared__ float tot[32];
    __shared__ float scratch[1024];
    const int H = q.H, NB = H + 1, tid = threadIdx.x, n_cols = H * NB + 2 * H;
    const int col = blockIdx.x * 32 + (tid & 31);
    reduce_col(q.partial, q.n_rows, n_cols, col < n_cols ? col : -1, scratch, tot, tid);
    // message weights [H, 3H] = [W_s | W_d | W_e]: dW_s = dW_d = dc (x) emb, dW_e from the MFMA accumulators, db = dc
    if (tid < 32 && col < n_cols) {
        const float t = tot[tid];
        if (col < H * NB) {
            const int o = col / NB, k = col - o * NB;
            float* row = q.grad_W_msg + (long)o * q.ld_w_msg;
            if (k < H) {
                row[2 * H + k] = t;
            } else {                          // dc[o]
                q.grad_b_msg[o] = t;
                for (int kk = 0; kk < H; ++kk) {
                    const float ge = t * q.emb[kk];
                    row[kk] = ge;
                    row[H + kk] = ge;
                }
            }
        } else {
            const int c = col - H * NB;       // grad_beta [H] | grad_gamma [H] of the input BatchNorm
            q.gsum[c] = t;
            if (c < H) q.grad_beta[c] = t; else q.grad_gamma[c - H] = t;
        }
    }
    if (blockIdx.x == 0) {                    // d emb += (W_s + W_d)^T dc: every dc, added once more in the same order
        __syncthreads();
        const int c0 = tid & 31;
        reduce_col(q.partial, q.n_rows, n_cols, c0 < H ? c0 * NB + H : -1, scratch, tot, tid);
        if (tid < H) {
            float acc = 0.f;
            for (int o = 0; o < H; ++o) {
                const float* row = q.W_msg + (long)o * q.ld_w_msg;
                acc = fmaf(row[tid] + row[H + tid], tot[o], acc);
            }
            q.grad_emb[tid] += acc;
        }
    }
}

__global__ void __launch_bounds__(1024) n3_reduce_in_kernel(ReduceK q) {        // R3: H (DIN + 1) columns
    __shared__ float tot[32];
    __shared__ float scratch[1024];
    const int H = q.H, NB = q.DIN + 1, tid = threadIdx.x, n_cols = H * NB;
    const int col = blockIdx.x * 32 + (tid & 31);
    reduce_col(q.partial, q.n_rows, n_cols, col < n_cols ? col : -1, scratch, tot, tid);
    if (tid < 32 && col < n_cols) {
        const int o = col / NB, k = col - o * NB;
        if (k < q.DIN) q.grad_W_in[(long)o * q.ld_w_in + k] = tot[tid];
        else q.grad_b_in[o] = tot[tid];
    }
}

// ---- host side
struct Plan {
    int fwd_rows_per_tile, fwd_tiles;
    int bwd_rows_per_block, bwd_blocks;
};

Plan plan_for(long E) {
    Plan pl;
    const long chunks = (E + TB - 1) / TB;
    const long per = (chunks + 1023) / 1024;              // <= 1024 statistics tiles
    pl.fwd_rows_per_tile = (int)(per * TB);
    pl.fwd_tiles = (int)((E + pl.fwd_rows_per_tile - 1) / pl.fwd_rows_per_tile);
    long perb = (chunks + MAX_BWD_BLOCKS - 1) / MAX_BWD_BLOCKS;
    if (perb < 2 && chunks >= 512) perb = 2;             // (a block's fixed cost - accumulator exchange, partial row - over >= 512 edges)
    pl.bwd_rows_per_block = (int)(perb * TB);
    pl.bwd_blocks = (int)((E + pl.bwd_rows_per_block - 1) / pl.bwd_rows_per_block);
    return pl;
}

bool supported(int hidden, int n_enc) { return (hidden == 20 && (n_enc == 4 || n_enc == 0)) || (hidden == 16 && n_enc == 2); }

EdgeK kernel_args(const I3dNet3dEdgeArgs* a) {
    EdgeK p{};
    p.E = a->num_edges; p.N = a->num_nodes; p.reduce_mean = a->reduce_mean;
    p.ld_w_in = a->ld_w_in; p.ld_w_msg = a->ld_w_msg;
    p.inv_rows = 1.f / (float)a->num_edges;
    p.d_raw = a->d_raw; p.perm = a->perm; p.dst_s = a->dst_s; p.in_ptr = a->in_ptr; p.emb = a->emb;
    p.W_in = a->W_in; p.b_in = a->b_in; p.W_msg = a->W_msg; p.b_msg = a->b_msg; p.w_gate = a->w_gate; p.b_gate = a->b_gate;
    p.aff_in = a->aff_in; p.aff_msg = a->aff_msg; p.invstd_in = a->tail_in.invstd; p.invstd_msg = a->tail_msg.invstd;
    p.x_msg = a->x_msg; p.d_out = a->d_out; p.msg = a->msg;
    p.x_center = a->store_bf16 ? a->x_center : nullptr;
    p.grad_m_sum = a->grad_m_sum; p.grad_ya = a->grad_ya;
    return p;
}

int check_common(const I3dNet3dEdgeArgs* a) {
    I3D_CHECK_ARG(a != nullptr, "null argument struct");
    I3D_CHECK_ARG(supported(a->hidden, a->n_enc), "hidden / fourier_encodings combination not built (i3d_net3d_edge_supported)");
    I3D_CHECK_ARG(a->num_edges > 0 && a->num_nodes > 0, "empty graph");
    I3D_CHECK_ARG(a->tail_msg.post_act == I3D_ACT_NONE, "the message block has no post activation");
    I3D_CHECK_ARG(a->tail_in.act == ACT && a->tail_in.post_act == ACT && a->tail_msg.act == ACT, "built for SiLU activations");
    I3D_CHECK_ARG(a->d_raw && a->perm && a->dst_s && a->in_ptr && a->emb && a->W_in && a->b_in && a->W_msg && a->b_msg &&
                      a->w_gate && a->b_gate, "null input");
    I3D_CHECK_ARG(a->aff_in && a->aff_msg && a->x_msg && a->d_out && a->tail_in.mean && a->tail_in.invstd &&
                      a->tail_msg.mean && a->tail_msg.invstd, "null saved buffer");
    I3D_CHECK_ARG(((((uintptr_t)a->x_msg | (uintptr_t)a->d_out) & 15) == 0), "saved buffers must be 16-byte aligned");
    return I3D_OK;
}

// synchronised BatchNorm (comm.hip): the backward sums of a BatchNorm of this stage - [sum dy | sum dy xhat] over THIS rank's
// edges, as the R kernels leave them - become the sums over all ranks (fp64 all-reduce on the caller's stream), and the row
// count the data gradient divides by becomes the edges of all ranks; grad_gamma / grad_beta keep this rank's share (the
// gradient all-reduce adds the ranks up)
__global__ void n3_sums_to_f64_kernel(const float* __restrict__ gsum, int n, double rows, double* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (double)gsum[i];
    if (i == n) out[n] = rows;
}
__global__ void n3_sums_from_f64_kernel(const double* __restrict__ in, int n, float* __restrict__ gsum, float* __restrict__ inv_rows) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) gsum[i] = (float)in[i];
    if (i == n) inv_rows[0] = (float)(1.0 / in[n]);
}

int sync_backward_sums(float* gsum, int n, long local_rows, float* inv_rows_dev, void* stream) {
    const I3dCollectives* coll = collectives();
    if (coll == nullptr) return I3D_OK;
    if (PeerCtx* pc = peer_active(stream))       // peer-write exchange (peer.h): fp32 in, fp64 sums over the ranks, fp32 + 1 / rows out: one launch
        return peer_sum_f32(pc, gsum, n, 1, (double)local_rows, nullptr, gsum, inv_rows_dev, stream);
    I3D_CHECK_ARG(coll->scratch_bytes >= (long)(n + 1) * 8, "collective scratch too small");
    double* s64 = (double*)coll->scratch;
    hipLaunchKernelGGL(n3_sums_to_f64_kernel, dim3(1), dim3(128), 0, (hipStream_t)stream, gsum, n, (double)local_rows, s64);
    I3D_CHECK_LAUNCH();
    const int rc = coll->all_reduce_f64(coll->user, s64, n + 1, stream);
    if (rc != I3D_OK) return rc;
    hipLaunchKernelGGL(n3_sums_from_f64_kernel, dim3(1), dim3(128), 0, (hipStream_t)stream, s64, n, gsum, inv_rows_dev);
    I3D_CHECK_LAUNCH();
    return I3D_OK;
}

#define N3_DISPATCH_T(KERNEL, B16V, grid, block, stream, ...)                                                           \
    do {                                                                                                               \
        if (a->hidden == 20 && a->n_enc == 4) hipLaunchKernelGGL((KERNEL<20, 4, B16V>), grid, block, 0, stream, __VA_ARGS__); \
        else if (a->hidden == 20 && a->n_enc == 0) hipLaunchKernelGGL((KERNEL<20, 0, B16V>), grid, block, 0, stream, __VA_ARGS__); \
        else hipLaunchKernelGGL((KERNEL<16, 2, B16V>), grid, block, 0, stream, __VA_ARGS__);                            \
    } while (0)
#define N3_DISPATCH_O(KERNEL, O, grid, block, stream, ...)                                                              \
    do {                                                                                                               \
        if (a->hidden == 20 && a->n_enc == 4) {                                                                        \
            if (a->store_bf16) hipLaunchKernelGGL((KERNEL<20, 4, true, O>), grid, block, 0, stream, __VA_ARGS__);      \
            else hipLaunchKernelGGL((KERNEL<20, 4, false, O>), grid, block, 0, stream, __VA_ARGS__);                   \
        } else if (a->hidden == 20 && a->n_enc == 0) {                                                                 \
            if (a->store_bf16) hipLaunchKernelGGL((KERNEL<20, 0, true, O>), grid, block, 0, stream, __VA_ARGS__);      \
            else hipLaunchKernelGGL((KERNEL<20, 0, false, O>), grid, block, 0, stream, __VA_ARGS__);                   \
        } else {                                                                                                       \
            if (a->store_bf16) hipLaunchKernelGGL((KERNEL<16, 2, true, O>), grid, block, 0, stream, __VA_ARGS__);      \
            else hipLaunchKernelGGL((KERNEL<16, 2, false, O>), grid, block, 0, stream, __VA_ARGS__);                   \
        }                                                                                                              \
    } while (0)
#define N3_DISPATCH(KERNEL, grid, block, stream, ...)                                                                  \
    do {                                                                                                               \
        if (a->store_bf16) N3_DISPATCH_T(KERNEL, true, grid, block, stream, __VA_ARGS__);                              \
        else N3_DISPATCH_T(KERNEL, false, grid, block, stream, __VA_ARGS__);                                           \
    } while (0)

}  // namespace
}  // namespace i3d

using namespace i3d;

extern "C" int i3d_net3d_edge_supported(int hidden, int n_enc) { return supported(hidden, n_enc) ? 1 : 0; }

// forward scratch: [tiles][3 hidden] statistics partials | the packed parameter block
static long stats_partial_floats(int num_edges, int hidden) { return ((long)plan_for(num_edges).fwd_tiles * 3 * hidden + 3) & ~3L; }

extern "C" long i3d_net3d_edge_stats_floats(int num_edges, int hidden) {
    if (num_edges <= 0 || hidden <= 0) return 0;
    return stats_partial_floats(num_edges, hidden) + PACK_FLOATS;
}

extern "C" long i3d_net3d_edge_bwd_floats(int num_edges, int hidden, int n_enc) {
    if (num_edges <= 0 || hidden <= 0) return 0;
    const int din = n_enc > 0 ? 2 * n_enc + 1 : 1;
    const long per = (long)hidden * (hidden + 1) + 2 * hidden;          // the widest pass (B2); B1: 3H+1, B3: H (DIN+1)
    const long per3 = (long)hidden * (din + 1);
    return (long)plan_for(num_edges).bwd_blocks * (per > per3 ? per : per3) + 4 * hidden + 4 + PACK_FLOATS;      // ... | gsum [4H] | 1 / rows of all ranks | packed parameters
}

extern "C" int i3d_net3d_edge_fwd(const I3dNet3dEdgeArgs* a, void* stream_) {
    if (int rc = check_common(a)) return rc;
    I3D_CHECK_ARG(a->stats && a->msg && a->m_sum, "null forward buffer");
    hipStream_t stream = (hipStream_t)stream_;
    const Plan pl = plan_for(a->num_edges);
    EdgeK p = kernel_args(a);
    p.rows_per_block = pl.fwd_rows_per_tile;
    p.partial = a->stats;
    p.packed = a->stats + stats_partial_floats(a->num_edges, a->hidden);
    const int H = a->hidden;
    hipLaunchKernelGGL(n3_pack_kernel, dim3(1), dim3(256), 0, stream, p, H, a->n_enc > 0 ? 2 * a->n_enc + 1 : 1);
    I3D_CHECK_LAUNCH();
    N3_DISPATCH(n3_stats_in_kernel, dim3(pl.fwd_tiles), dim3(TB), stream, p);
    I3D_CHECK_LAUNCH();
    const I3dBnTail& t1 = a->tail_in;
    if (int rc = i3d_bn_finalize_partials(a->stats, pl.fwd_tiles, H, t1.eps, t1.momentum, t1.gamma, t1.beta, t1.mean, t1.invstd,
                                          t1.running_mean, t1.running_var, t1.num_batches_tracked, a->aff_in, stream_))
        return rc;
    if (a->store_bf16) {
        I3D_CHECK_ARG(a->x_center != nullptr, "bf16 storage needs x_center");
        N3_DISPATCH(n3_center_kernel, dim3(1), dim3(TB), stream, p);
        I3D_CHECK_LAUNCH();
    }
    N3_DISPATCH(n3_msg_pre_kernel, dim3(pl.fwd_tiles), dim3(TB), stream, p);
    I3D_CHECK_LAUNCH();
    const I3dBnTail& t2 = a->tail_msg;
    if (int rc = i3d_bn_finalize_partials(a->stats, pl.fwd_tiles, H, t2.eps, t2.momentum, t2.gamma, t2.beta, t2.mean, t2.invstd,
                                          t2.running_mean, t2.running_var, t2.num_batches_tracked, a->aff_msg, stream_))
        return rc;
    N3_DISPATCH(n3_gate_kernel, dim3(cdiv(a->num_edges, TB)), dim3(TB), stream, p);
    I3D_CHECK_LAUNCH();
    if (a->store_bf16) return i3d_segment_sum_bf16(a->msg, H, a->in_ptr, nullptr, a->num_nodes, H, a->reduce_mean, a->m_sum, H, stream_);
    return i3d_segment_sum(a->msg, H, a->in_ptr, nullptr, a->num_nodes, H, a->reduce_mean, a->m_sum, H, stream_);
}

extern "C" int i3d_net3d_edge_bwd(const I3dNet3dEdgeArgs* a, void* stream_) {
    if (int rc = check_common(a)) return rc;
    I3D_CHECK_ARG(a->grad_m_sum && a->grad_ya && a->partial, "null backward buffer");      // (grad_lin: no longer used)
    I3D_CHECK_ARG(a->grad_W_in && a->grad_b_in && a->grad_gamma_in && a->grad_beta_in && a->grad_W_msg && a->grad_b_msg &&
                      a->grad_gamma_msg && a->grad_beta_msg && a->grad_w_gate && a->grad_b_gate && a->grad_emb,
                  "null gradient buffer");
    hipStream_t stream = (hipStream_t)stream_;
    const Plan pl = plan_for(a->num_edges);
    const int H = a->hidden, DIN = a->n_enc > 0 ? 2 * a->n_enc + 1 : 1;
    const long per = (long)H * (H + 1) + 2 * H, per3 = (long)H * (DIN + 1);
    float* gsum = a->partial + (long)pl.bwd_blocks * (per > per3 ? per : per3);      // [2H] message | [2H] input
    EdgeK p = kernel_args(a);
    p.rows_per_block = pl.bwd_rows_per_block;
    p.partial = a->partial;
    p.packed = gsum + 4 * H + 4;
    hipLaunchKernelGGL(n3_pack_kernel, dim3(1), dim3(256), 0, stream, p, H, DIN);
    I3D_CHECK_LAUNCH();
    ReduceK q{};
    q.H = H; q.DIN = DIN; q.n_rows = pl.bwd_blocks; q.ld_w_in = a->ld_w_in; q.ld_w_msg = a->ld_w_msg;
    q.partial = a->partial; q.emb = a->emb; q.W_msg = a->W_msg;
    q.grad_w_gate = a->grad_w_gate; q.grad_b_gate = a->grad_b_gate; q.grad_W_msg = a->grad_W_msg; q.grad_b_msg = a->grad_b_msg;
    q.grad_emb = a->grad_emb; q.grad_W_in = a->grad_W_in; q.grad_b_in = a->grad_b_in;

    N3_DISPATCH(n3_bwd_sums_kernel, dim3(pl.bwd_blocks), dim3(TB), stream, p);
    I3D_CHECK_LAUNCH();
    q.gsum = gsum; q.grad_gamma = a->grad_gamma_msg; q.grad_beta = a->grad_beta_msg;
    hipLaunchKernelGGL(n3_reduce_sums_kernel, dim3(cdiv(3 * H + 1, 32)), dim3(1024), 0, stream, q);
    I3D_CHECK_LAUNCH();
    const bool synced = collectives() != nullptr;
    float* inv_rows_dev = gsum + 4 * H;
    I3D_CHECK_ARG(!synced || 2 * H < 128, "hidden too wide for the sum exchange");
    if (synced) {
        if (int rc = sync_backward_sums(gsum, 2 * H, a->num_edges, inv_rows_dev, stream_)) return rc;
        p.inv_rows_dev = inv_rows_dev;
    }
    p.gsum_msg = gsum;
    // (two waves per SIMD: 256 registers without spills; three - 168 registers, ~40 spilled - measured slower)
    N3_DISPATCH_O(n3_bwd_msg_kernel, 2, dim3(pl.bwd_blocks), dim3(TB), stream, p);
    I3D_CHECK_LAUNCH();
    q.gsum = gsum + 2 * H; q.grad_gamma = a->grad_gamma_in; q.grad_beta = a->grad_beta_in;
    hipLaunchKernelGGL(n3_reduce_msg_kernel, dim3(cdiv((int)per, 32)), dim3(1024), 0, stream, q);
    I3D_CHECK_LAUNCH();
    if (synced) {
        if (int rc = sync_backward_sums(gsum + 2 * H, 2 * H, a->num_edges, inv_rows_dev, stream_)) return rc;
    }
    p.gsum_in = gsum + 2 * H;
    N3_DISPATCH(n3_bwd_in_kernel, dim3(pl.bwd_blocks), dim3(TB), stream, p);
    I3D_CHECK_LAUNCH();
    hipLaunchKernelGGL(n3_reduce_in_kernel, dim3(cdiv((int)per3, 32)), dim3(1024), 0, stream, q);
    I3D_CHECK_LAUNCH();
    return I3D_OK;
}
