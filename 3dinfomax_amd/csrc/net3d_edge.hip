// The edge stage of the 3D network in one lane per edge (round 2).
//
// reference models/net3d.py:57-81, 100-118 for the structure of the pre-training configs (propagation_depth 1, one
// message block, node embedding broadcast from one parameter):
//     f    = fourier(d)                                     commons/utils.py:103-110
//     e0   = post( BN_in( act( W_in f + b_in ) ) )          edge_input block + the outer SiLU (net3d.py:80-81)
//     m    = BN_msg( act( W_s h_src + W_d h_dst + W_e e0 + b_msg ) )      message block on [h_src | h_dst | d] (:113-115)
//     msg  = m * sigmoid( w_g . m + b_g )                   soft edge gate (:117-118)
//     m_sum[v] = mean / sum of msg over the in-edges of v   (:109)
// With h the broadcast of ONE vector, W_s h_src + W_d h_dst + b_msg is one constant vector c: the whole stage is a
// function of the scalar distance of the edge and of the two BatchNorm statistics.  The per-block path (net3d_native.py)
// runs it as ~20 launches per direction over [E, 20] tensors (GEMM tiles 20 columns wide, a statistics pass and an apply
// pass per BatchNorm).  Here a lane owns an edge and carries the 20-wide vectors in registers; the weights sit in LDS and
// are read as broadcasts; what goes through memory is what has to exist as a tensor (the distance embedding the forward
// leaves on the graph, edge-id order) plus ONE saved [E, H] activation and one transient per direction.
//
//   forward   F1  statistics of act(W_in f + b)                      -> per-tile partials -> finalisation (fused_bn.hip)
//             F2  e0 (stored, edge-id order), x_msg (stored), its statistics -> finalisation
//             F3  m, gate, msg (stored) -> i3d_segment_sum
//   backward  B1  sums of the message BatchNorm + the gate's parameter gradients        -> R1
//             B2a gradient through the message block (stored): dW_e and dc on the MFMA unit (a wave's 64 edges are the K
//                 dimension of v_mfma_f32_32x32x2_f32, a ones column gives the column sum)
//             B2b grad of e0 -> through the post activation (stored), sums of the input BatchNorm     -> R2
//             B3  gradient through the edge-input block: dW_in | db_in on the MFMA unit  -> R3
// Every pass recomputes the cheap part of the chain (Fourier features, the [H, 2 n_enc + 1] product) from the distance.
// Deterministic: per-lane -> wave tree -> wave order -> block order sums, no atomics.
#include "common.h"
#include "peer.h"

namespace i3d {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int TB = 256;
// The three activations of the stage (edge-input block, the outer one of net3d.py:81, message block) are SiLU in the
// reference's configs (`activation: SiLU` is the constructor default and :81 is hard-coded): compiled in.  A runtime code
// per element turns the unrolled 20-wide loops into ~1000 branches and 500 live registers.
constexpr int ACT = I3D_ACT_SILU;
constexpr int odd(int n) { return n | 1; }   // row stride of an MFMA operand tile in LDS: odd -> the per-lane row writes hit
                                             // distinct banks
// The weights are loop-invariant LDS reads: without this the compiler hoists all ~600 of them out of the per-edge loops
// into registers and spills.  A compiler-level memory barrier at the top of an iteration keeps them as LDS broadcasts.
#define N3_NO_HOIST() __asm__ volatile("" ::: "memory")
// <= 168 registers per lane (3 waves per SIMD): the 3D network runs on a stream of its own next to the 2D network - a wave
// that takes the whole register file of its SIMD (the unrolled 20-wide code schedules to 470 registers when allowed to)
// shuts the other stream out of every CU for the length of the kernel
#define N3_OCC __attribute__((amdgpu_waves_per_eu(3, 3)))
constexpr int MAX_BWD_BLOCKS = 512;  // two blocks per CU; the R kernels add this many partial rows per column

template <int H, int NENC>
struct Dims {
    static constexpr int DIN = NENC > 0 ? 2 * NENC + 1 : 1;
};

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
    float* x_msg;
    float* x_center;                 // [H] bf16 storage: x_msg holds bf16(x - x_center) (written by n3_center_kernel, read by everyone)
    float* d_out;
    float* msg;
    const float* grad_m_sum;
    float* grad_ya;
    float* grad_lin;
    float* partial;
};

template <int H, int DIN>
struct Wts {                         // LDS copy of the parameters of the stage
    float w_in[H * DIN];
    float b_in[H];
    float w_e[H * H];
    float w_et[H * H];               // transposed: the backward product reads contiguous rows too
    float c[H];
    float aff_in[3 * H];
    float aff_msg[3 * H];
    float istd_in[H];
    float istd_msg[H];
    float w_g[H];
    float gs_in[2 * H];
    float gs_msg[2 * H];
    float ctr[H];
    float b_g;
};

template <int H, int DIN>
__device__ __forceinline__ void load_weights(Wts<H, DIN>& w, const EdgeK& p, int tid) {
    for (int i = tid; i < H * DIN; i += TB) w.w_in[i] = p.W_in[(i / DIN) * p.ld_w_in + (i % DIN)];
    for (int i = tid; i < H * H; i += TB) {
        const float v = p.W_msg[(i / H) * p.ld_w_msg + 2 * H + (i % H)];
        w.w_e[i] = v;
        w.w_et[(i % H) * H + (i / H)] = v;
    }
    for (int i = tid; i < 3 * H; i += TB) {
        w.aff_in[i] = p.aff_in[i];
        w.aff_msg[i] = p.aff_msg[i];
    }
    for (int i = tid; i < 2 * H; i += TB) {
        w.gs_in[i] = p.gsum_in ? p.gsum_in[i] : 0.f;
        w.gs_msg[i] = p.gsum_msg ? p.gsum_msg[i] : 0.f;
    }
    if (tid < H) {
        w.b_in[tid] = p.b_in[tid];
        w.istd_in[tid] = p.invstd_in[tid];
        w.istd_msg[tid] = p.invstd_msg[tid];
        w.w_g[tid] = p.w_gate[tid];
        w.ctr[tid] = p.x_center != nullptr ? p.x_center[tid] : 0.f;
        // c = (W_s + W_d) emb + b_msg: every node carries the same embedding (reference net3d.py:61)
        float acc = p.b_msg[tid];
        const float* row = p.W_msg + (long)tid * p.ld_w_msg;
        for (int k = 0; k < H; ++k) acc = fmaf(row[k] + row[H + k], p.emb[k], acc);
        w.c[tid] = acc;
    }
    if (tid == 0) w.b_g = p.b_gate[0];
}

// reference commons/utils.py:103-110: [sin(d / 2^k)]_k | [cos(d / 2^k)]_k | d.  ONE accurate sincos of the smallest angle
// d / 2^(n-1), the others by angle doubling (sin 2t = 2 sin t cos t, cos 2t = 1 - 2 sin^2 t): each doubling at most doubles
// the absolute error, 2^(n-1) ulp = 5e-7 at n = 4 - the four passes that need the features recompute them from the
// distance, and eight range-reduced sinf / cosf per edge and pass were a third of their instruction count
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

// a = W_in f + b_in, xa = act(a)
template <int H, int DIN>
__device__ __forceinline__ void lin_in(const Wts<H, DIN>& w, const float* f, int act, float* a, float* xa) {
#pragma unroll
    for (int o = 0; o < H; ++o) {
        N3_NO_HOIST();
        float acc = w.b_in[o];
#pragma unroll
        for (int k = 0; k < DIN; ++k) acc = fmaf(w.w_in[o * DIN + k], f[k], acc);
        a[o] = acc;
        xa[o] = apply_act(acc, act);
    }
}

template <int H>
__device__ __forceinline__ void bn_apply(const float* aff, const float* x, float* y) {
#pragma unroll
    for (int c = 0; c < H; ++c) y[c] = (x[c] - aff[c]) * aff[H + c] + aff[2 * H + c];
}

// lin = c + W_e e0
template <int H, int DIN>
__device__ __forceinline__ void lin_msg(const Wts<H, DIN>& w, const float* e0, float* lin) {
#pragma unroll
    for (int o = 0; o < H; ++o) {
        N3_NO_HOIST();
        float acc = w.c[o];
#pragma unroll
        for (int k = 0; k < H; ++k) acc = fmaf(w.w_e[o * H + k], e0[k], acc);
        lin[o] = acc;
    }
}

template <int H>
__device__ __forceinline__ void load_row(const float* p, float* v) {
#pragma unroll
    for (int c = 0; c < H; c += 4) {
        const float4 t = *reinterpret_cast<const float4*>(p + c);
        v[c] = t.x; v[c + 1] = t.y; v[c + 2] = t.z; v[c + 3] = t.w;
    }
}

template <int H>
__device__ __forceinline__ void store_row(float* p, const float* v) {
#pragma unroll
    for (int c = 0; c < H; c += 4) *reinterpret_cast<float4*>(p + c) = make_float4(v[c], v[c + 1], v[c + 2], v[c + 3]);
}

// The [E, H] ACTIVATIONS that only this stage reads and writes - x_msg (saved) and msg - in the storage type of
// the launch: fp32, or (bf16 matmul mode, EdgeK.store16) bf16 - half the bytes of the passes that are bound by them (at the
// QMugs shape an [E3, 20] fp32 tensor is 313 MB and the stage moves eleven of them per step).  Row `row` of a buffer that was
// sized for fp32; values are rounded (RNE) once, where they are stored; statistics are taken before the rounding.
template <int H, bool B16>
__device__ __forceinline__ void load_srow(const float* base, long row, float* v, const float* ctr = nullptr) {
    if constexpr (!B16) {
        load_row<H>(base + row * H, v);
    } else {
        const uint2* q = reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(base) + row * H);
#pragma unroll
        for (int c = 0; c < H; c += 4) {
            const uint2 t = q[c / 4];
            v[c] = __uint_as_float(t.x << 16); v[c + 1] = __uint_as_float(t.x & 0xffff0000u);
            v[c + 2] = __uint_as_float(t.y << 16); v[c + 3] = __uint_as_float(t.y & 0xffff0000u);
        }
        if (ctr != nullptr) {
#pragma unroll
            for (int c = 0; c < H; ++c) v[c] += ctr[c];
        }
    }
}

__device__ __forceinline__ unsigned bf16_rne(float x) {      // bits of bf16(x), round to nearest even (finite inputs)
    const unsigned u = __float_as_uint(x);
    return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}

template <int H, bool B16>
__device__ __forceinline__ void store_srow(float* base, long row, const float* v, const float* ctr = nullptr) {
    if constexpr (!B16) {
        store_row<H>(base + row * H, v);
    } else {
        uint2* q = reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(base) + row * H);
        float d[H];
#pragma unroll
        for (int c = 0; c < H; ++c) d[c] = ctr != nullptr ? v[c] - ctr[c] : v[c];
#pragma unroll
        for (int c = 0; c < H; c += 4)
            q[c / 4] = make_uint2(bf16_rne(d[c]) | (bf16_rne(d[c + 1]) << 16), bf16_rne(d[c + 2]) | (bf16_rne(d[c + 3]) << 16));
    }
}

// column sums over the block: on return red[w * NC + c] holds wave w's sum of column c (4 waves)
template <int NC>
__device__ __forceinline__ void block_sum_cols(float* v, float* red, int tid) {
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        float x = v[c];
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) x += __shfl_xor(x, off);
        v[c] = x;
    }
    if ((tid & 63) == 0) {
#pragma unroll
        for (int c = 0; c < NC; ++c) red[(tid >> 6) * NC + c] = v[c];
    }
    __syncthreads();
}

// ---- statistics of a tile from sums about a pivot (a sample of the tile: no cancellation whatever mean / std of the column)
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

// F1: statistics of xa = act(W_in f + b_in)
template <int H, int NENC, bool B16>
__global__ void __launch_bounds__(TB) N3_OCC n3_stats_in_kernel(EdgeK p) {
    constexpr int DIN = Dims<H, NENC>::DIN;
    __shared__ Wts<H, DIN> w;
    __shared__ float pivot[H];
    __shared__ float red[4 * 2 * H];
    const int tid = threadIdx.x;
    load_weights(w, p, tid);
    __syncthreads();
    const long j0 = (long)blockIdx.x * p.rows_per_block;
    const long jend = min(j0 + p.rows_per_block, (long)p.E);
    float f[DIN], a[H], xa[H], s[2 * H];
#pragma unroll
    for (int c = 0; c < 2 * H; ++c) s[c] = 0.f;
    long j = j0 + tid;
    const bool first = j < jend;
    if (first) {
        fourier<NENC>(p.d_raw[p.perm[j]], f);
        lin_in<H, DIN>(w, f, ACT, a, xa);
    }
    if (tid == 0) {
#pragma unroll
        for (int c = 0; c < H; ++c) pivot[c] = xa[c];
    }
    __syncthreads();
    float pv[H];
#pragma unroll
    for (int c = 0; c < H; ++c) pv[c] = pivot[c];
    if (first) {
#pragma unroll
        for (int c = 0; c < H; ++c) { const float t = xa[c] - pv[c]; s[c] += t; s[H + c] = fmaf(t, t, s[H + c]); }
    }
    for (j += TB; j < jend; j += TB) {
        N3_NO_HOIST();
        fourier<NENC>(p.d_raw[p.perm[j]], f);
        lin_in<H, DIN>(w, f, ACT, a, xa);
#pragma unroll
        for (int c = 0; c < H; ++c) { const float t = xa[c] - pv[c]; s[c] += t; s[H + c] = fmaf(t, t, s[H + c]); }
    }
    block_sum_cols<2 * H>(s, red, tid);
    write_tile_partial<H>(p.partial, blockIdx.x, red, pivot, (float)(jend - j0), tid);
}

// F2: e0 -> d_out (edge-id order), x_msg = act(c + W_e e0) (stored, destination-sorted) and its statistics
// bf16 storage: the centre x_msg is stored about - the column means of the message block's activation over the first TB edges.
// A pre-BatchNorm activation must not be rounded as it is: its columns can be nearly constant over the batch (every edge
// feature is a function of ONE scalar distance), the BatchNorm behind then divides a 2^-9 |x| rounding error by a standard
// deviation that is smaller than that (measured with raw storage: 10-96 % error on the stage's parameter gradients at 32
// molecules).  About a centre inside the distribution the rounding error is 2^-9 of the DEVIATION - what the BatchNorm scales.
template <int H, int NENC, bool B16>
__global__ void __launch_bounds__(TB) N3_OCC n3_center_kernel(EdgeK p) {
    constexpr int DIN = Dims<H, NENC>::DIN;
    __shared__ Wts<H, DIN> w;
    __shared__ float red[4 * H];
    const int tid = threadIdx.x;
    load_weights(w, p, tid);
    __syncthreads();
    float f[DIN], a[H], xa[H], e0[H], xm[H];
    const int n = p.E < TB ? p.E : TB;
    if (tid < n) {
        N3_NO_HOIST();
        fourier<NENC>(p.d_raw[p.perm[tid]], f);
        lin_in<H, DIN>(w, f, ACT, a, xa);
        bn_apply<H>(w.aff_in, xa, e0);
#pragma unroll
        for (int c = 0; c < H; ++c) e0[c] = apply_act(e0[c], ACT);
        lin_msg<H, DIN>(w, e0, xm);
#pragma unroll
        for (int c = 0; c < H; ++c) xm[c] = apply_act(xm[c], ACT);
    } else {
#pragma unroll
        for (int c = 0; c < H; ++c) xm[c] = 0.f;
    }
    block_sum_cols<H>(xm, red, tid);
    if (tid < H) p.x_center[tid] = (((red[tid] + red[H + tid]) + red[2 * H + tid]) + red[3 * H + tid]) / (float)n;
}

template <int H, int NENC, bool B16>
__global__ void __launch_bounds__(TB) N3_OCC n3_msg_pre_kernel(EdgeK p) {
    constexpr int DIN = Dims<H, NENC>::DIN;
    __shared__ Wts<H, DIN> w;
    __shared__ float pivot[H];
    __shared__ float red[4 * 2 * H];
    const int tid = threadIdx.x;
    load_weights(w, p, tid);
    __syncthreads();
    const long j0 = (long)blockIdx.x * p.rows_per_block;
    const long jend = min(j0 + p.rows_per_block, (long)p.E);
    float f[DIN], a[H], xa[H], e0[H], xm[H], s[2 * H];
#pragma unroll
    for (int c = 0; c < 2 * H; ++c) s[c] = 0.f;
    auto edge = [&](long jj) {
        N3_NO_HOIST();
        const int eid = p.perm[jj];
        fourier<NENC>(p.d_raw[eid], f);
        lin_in<H, DIN>(w, f, ACT, a, xa);
        bn_apply<H>(w.aff_in, xa, e0);
#pragma unroll
        for (int c = 0; c < H; ++c) e0[c] = apply_act(e0[c], ACT);
        store_row<H>(p.d_out + (long)eid * H, e0);
        lin_msg<H, DIN>(w, e0, xm);
#pragma unroll
        for (int c = 0; c < H; ++c) xm[c] = apply_act(xm[c], ACT);
        store_srow<H, B16>(p.x_msg, jj, xm, w.ctr);
    };
    long j = j0 + tid;
    const bool first = j < jend;
    if (first) edge(j);
    if (tid == 0) {
#pragma unroll
        for (int c = 0; c < H; ++c) pivot[c] = xm[c];
    }
    __syncthreads();
    float pv[H];
#pragma unroll
    for (int c = 0; c < H; ++c) pv[c] = pivot[c];
    if (first) {
#pragma unroll
        for (int c = 0; c < H; ++c) { const float t = xm[c] - pv[c]; s[c] += t; s[H + c] = fmaf(t, t, s[H + c]); }
    }
    for (j += TB; j < jend; j += TB) {
        edge(j);
#pragma unroll
        for (int c = 0; c < H; ++c) { const float t = xm[c] - pv[c]; s[c] += t; s[H + c] = fmaf(t, t, s[H + c]); }
    }
    block_sum_cols<2 * H>(s, red, tid);
    write_tile_partial<H>(p.partial, blockIdx.x, red, pivot, (float)(jend - j0), tid);
}

// the gate: w = sigmoid(w_g . m + b_g) (same expression as edge.hip soft_edge_fwd_kernel)
template <int H, int DIN>
__device__ __forceinline__ float gate_of(const Wts<H, DIN>& w, const float* m) {
    float dot = w.b_g;
#pragma unroll
    for (int c = 0; c < H; ++c) dot += m[c] * w.w_g[c];
    return 1.f / (1.f + expf(-dot));
}

// F3: msg = m * gate, m = BN_msg(x_msg)
template <int H, int NENC, bool B16>
__global__ void __launch_bounds__(TB) N3_OCC n3_gate_kernel(EdgeK p) {
    constexpr int DIN = Dims<H, NENC>::DIN;
    __shared__ Wts<H, DIN> w;
    const int tid = threadIdx.x;
    load_weights(w, p, tid);
    __syncthreads();
    const long j = (long)blockIdx.x * TB + tid;
    if (j >= p.E) return;
    float xm[H], m[H];
    load_srow<H, B16>(p.x_msg, j, xm, w.ctr);
    bn_apply<H>(w.aff_msg, xm, m);
    const float g = gate_of<H, DIN>(w, m);
#pragma unroll
    for (int c = 0; c < H; ++c) m[c] *= g;
    store_srow<H, B16>(p.msg, j, m);
}

// gradient reaching m of edge j:  gm = gmsg * g + gg * w_g,  gg = (gmsg . m) g (1 - g),  gmsg = grad_m_sum[dst] (/ deg)
template <int H, int DIN>
__device__ __forceinline__ void grad_m(const Wts<H, DIN>& w, const EdgeK& p, long j, const float* m, float* gm, float& gg) {
    const int v = p.dst_s[j];
    float sc = 1.f;
    if (p.reduce_mean) sc = 1.f / (float)(p.in_ptr[v + 1] - p.in_ptr[v]);
    float gmsg[H];
    load_row<H>(p.grad_m_sum + (long)v * H, gmsg);
    const float g = gate_of<H, DIN>(w, m);
    float dot = 0.f;
#pragma unroll
    for (int c = 0; c < H; ++c) { gmsg[c] *= sc; dot += gmsg[c] * m[c]; }
    gg = dot * g * (1.f - g);
#pragma unroll
    for (int c = 0; c < H; ++c) gm[c] = gmsg[c] * g + gg * w.w_g[c];
}

// B1: partial[block] = sum gm | sum gm xhat_m | sum gg m | sum gg
template <int H, int NENC, bool B16>
__global__ void __launch_bounds__(TB) N3_OCC n3_bwd_sums_kernel(EdgeK p) {
    constexpr int DIN = Dims<H, NENC>::DIN;
    constexpr int NC = 3 * H + 1;
    __shared__ Wts<H, DIN> w;
    __shared__ float red[4 * NC];
    const int tid = threadIdx.x;
    load_weights(w, p, tid);
    __syncthreads();
    const long j0 = (long)blockIdx.x * p.rows_per_block;
    const long jend = min(j0 + p.rows_per_block, (long)p.E);
    float s[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) s[c] = 0.f;
    for (long j = j0 + tid; j < jend; j += TB) {
        N3_NO_HOIST();
        float xm[H], m[H], gm[H], gg;
        load_srow<H, B16>(p.x_msg, j, xm, w.ctr);
        bn_apply<H>(w.aff_msg, xm, m);
        grad_m<H, DIN>(w, p, j, m, gm, gg);
#pragma unroll
        for (int c = 0; c < H; ++c) {
            const float xh = (xm[c] - w.aff_msg[c]) * w.istd_msg[c];
            s[c] += gm[c];
            s[H + c] = fmaf(gm[c], xh, s[H + c]);
            s[2 * H + c] = fmaf(gg, m[c], s[2 * H + c]);
        }
        s[3 * H] += gg;
    }
    block_sum_cols<NC>(s, red, tid);
    if (tid < NC) p.partial[(long)blockIdx.x * NC + tid] = red[tid] + red[NC + tid] + red[2 * NC + tid] + red[3 * NC + tid];
}

// D[i][j] += sum over the wave's 64 edges of A[edge][i] * B[edge][j]: v_mfma_f32_32x32x2_f32, the edge pair (2 s, 2 s + 1)
// is the K dimension of step s.  tile_a / tile_b: this wave's [64][odd(NA)] / [64][odd(NB)] tiles; the lanes of the
// padding columns (>= NA / NB) feed zeros.
template <int NA, int NB>
__device__ __forceinline__ void outer_accumulate(float* tile_a, float* tile_b, const float* a, const float* b, bool valid, int lane,
                                                 f32x16& acc) {
    constexpr int LDA = odd(NA), LDB = odd(NB);
#pragma unroll
    for (int c = 0; c < NA; ++c) tile_a[lane * LDA + c] = valid ? a[c] : 0.f;
#pragma unroll
    for (int c = 0; c < NB; ++c) tile_b[lane * LDB + c] = valid ? b[c] : 0.f;
    __syncthreads();
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
    __syncthreads();
}

template <int NA, int NB>
constexpr int tile_floats() { return 4 * 64 * (odd(NA) + odd(NB)) > 4096 ? 4 * 64 * (odd(NA) + odd(NB)) : 4096; }

// the four waves' accumulators -> partial[NA * NB] of the block.  scratch: >= 4 * 1024 floats of LDS
template <int NA, int NB>
__device__ __forceinline__ void write_outer(const f32x16& acc, float* scratch, float* out, int tid) {
    const int lane = tid & 63, wv = tid >> 6;
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

// B2a: gradient through the message block: glin (stored), partial[block] = [H][H + 1] (dW_e | dc)
template <int H, int NENC, bool B16>
__global__ void __launch_bounds__(TB) N3_OCC n3_bwd_msg_kernel(EdgeK p) {
    constexpr int DIN = Dims<H, NENC>::DIN;
    constexpr int NB = H + 1, NP = H * NB + 2 * H;
    __shared__ Wts<H, DIN> w;
    __shared__ float tiles[tile_floats<H, NB>()];      // also the accumulator exchange (4 * 1024 floats) at the end
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    load_weights(w, p, tid);
    const float inv_rows = p.inv_rows_dev != nullptr ? p.inv_rows_dev[0] : p.inv_rows;
    __syncthreads();
    float* tile_a = tiles + wv * 64 * (odd(H) + odd(NB));
    float* tile_b = tile_a + 64 * odd(H);
    const long j0 = (long)blockIdx.x * p.rows_per_block;
    const long jend = min(j0 + p.rows_per_block, (long)p.E);
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (long base = j0; base < jend; base += TB) {
        N3_NO_HOIST();
        const long j = base + tid;
        const bool valid = j < jend;
        float glin[H], e0x[NB];
#pragma unroll
        for (int c = 0; c < H; ++c) { glin[c] = 0.f; e0x[c] = 0.f; }
        e0x[H] = 1.f;
        if (valid) {
            float xm[H], m[H], gm[H], gg, lin[H];
            load_srow<H, B16>(p.x_msg, j, xm, w.ctr);
            load_row<H>(p.d_out + (long)p.perm[j] * H, e0x);
            bn_apply<H>(w.aff_msg, xm, m);
            grad_m<H, DIN>(w, p, j, m, gm, gg);
            lin_msg<H, DIN>(w, e0x, lin);
#pragma unroll
            for (int c = 0; c < H; ++c) {
                const float xh = (xm[c] - w.aff_msg[c]) * w.istd_msg[c];
                const float gx = w.aff_msg[H + c] * (gm[c] - w.gs_msg[c] * inv_rows - xh * (w.gs_msg[H + c] * inv_rows));
                glin[c] = gx * act_grad(lin[c], ACT);
            }
            store_row<H>(p.grad_lin + j * H, glin);
        }
        outer_accumulate<H, NB>(tile_a, tile_b, glin, e0x, valid, lane, acc);
    }
    write_outer<H, NB>(acc, tiles, p.partial + (long)blockIdx.x * NP, tid);
}

// B2b: grad of e0 = W_e^T glin, through the post activation (needs ya = BN_in(xa), recomputed from the distance): gya
// (stored), partial[block][H (H + 1) ...] = sum gya [H] | sum gya xhat_a [H]
template <int H, int NENC, bool B16>
__global__ void __launch_bounds__(TB) N3_OCC n3_bwd_post_kernel(EdgeK p) {
    constexpr int DIN = Dims<H, NENC>::DIN;
    constexpr int NB = H + 1, NP = H * NB + 2 * H;
    __shared__ Wts<H, DIN> w;
    __shared__ float red[4 * 2 * H];
    const int tid = threadIdx.x;
    load_weights(w, p, tid);
    __syncthreads();
    const long j0 = (long)blockIdx.x * p.rows_per_block;
    const long jend = min(j0 + p.rows_per_block, (long)p.E);
    float s[2 * H];
#pragma unroll
    for (int c = 0; c < 2 * H; ++c) s[c] = 0.f;
    for (long j = j0 + tid; j < jend; j += TB) {
        N3_NO_HOIST();
        float f[DIN], a[H], xa[H], ya[H], glin[H], gya[H];
        fourier<NENC>(p.d_raw[p.perm[j]], f);
        lin_in<H, DIN>(w, f, ACT, a, xa);
        bn_apply<H>(w.aff_in, xa, ya);
        load_row<H>(p.grad_lin + j * H, glin);
#pragma unroll
        for (int k = 0; k < H; ++k) {
            N3_NO_HOIST();
            float ge = 0.f;
#pragma unroll
            for (int o = 0; o < H; ++o) ge = fmaf(w.w_et[k * H + o], glin[o], ge);
            gya[k] = ge * act_grad(ya[k], ACT);
            const float xh = (xa[k] - w.aff_in[k]) * w.istd_in[k];
            s[k] += gya[k];
            s[H + k] = fmaf(gya[k], xh, s[H + k]);
        }
        store_row<H>(p.grad_ya + j * H, gya);
    }
    block_sum_cols<2 * H>(s, red, tid);
    if (tid < 2 * H)
        p.partial[(long)blockIdx.x * NP + H * NB + tid] = red[tid] + red[2 * H + tid] + red[4 * H + tid] + red[6 * H + tid];
}

// B3: gradient through the edge-input block: partial[block] = [H][DIN + 1] (dW_in | db_in)
template <int H, int NENC, bool B16>
__global__ void __launch_bounds__(TB) N3_OCC n3_bwd_in_kernel(EdgeK p) {
    constexpr int DIN = Dims<H, NENC>::DIN;
    constexpr int NB = DIN + 1, NP = H * NB;
    __shared__ Wts<H, DIN> w;
    __shared__ float tiles[tile_floats<H, NB>()];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    load_weights(w, p, tid);
    const float inv_rows = p.inv_rows_dev != nullptr ? p.inv_rows_dev[0] : p.inv_rows;
    __syncthreads();
    float* tile_a = tiles + wv * 64 * (odd(H) + odd(NB));
    float* tile_b = tile_a + 64 * odd(H);
    const long j0 = (long)blockIdx.x * p.rows_per_block;
    const long jend = min(j0 + p.rows_per_block, (long)p.E);
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (long base = j0; base < jend; base += TB) {
        N3_NO_HOIST();
        const long j = base + tid;
        const bool valid = j < jend;
        float ga[H], fx[NB];
#pragma unroll
        for (int c = 0; c < H; ++c) ga[c] = 0.f;
#pragma unroll
        for (int c = 0; c < DIN; ++c) fx[c] = 0.f;
        fx[DIN] = 1.f;
        if (valid) {
            float a[H], xa[H], gya[H];
            fourier<NENC>(p.d_raw[p.perm[j]], fx);
            lin_in<H, DIN>(w, fx, ACT, a, xa);
            load_row<H>(p.grad_ya + j * H, gya);
#pragma unroll
            for (int c = 0; c < H; ++c) {
                const float xh = (xa[c] - w.aff_in[c]) * w.istd_in[c];
                const float gx = w.aff_in[H + c] * (gya[c] - w.gs_in[c] * inv_rows - xh * (w.gs_in[H + c] * inv_rows));
                ga[c] = gx * act_grad(a[c], ACT);
            }
        }
        outer_accumulate<H, NB>(tile_a, tile_b, ga, fx, valid, lane, acc);
    }
    write_outer<H, NB>(acc, tiles, p.partial + (long)blockIdx.x * NP, tid);
}

// ---- R kernels: one block adds the partial rows of the pass (block order) and writes the parameter gradients
// tot[c] = sum over the rows of partial[row][c] (fixed order).  The block's threads form `groups` row groups (a power of
// two) x `cw` columns; group g adds rows g, g + groups, ...; the groups are added in group order.  scratch: nthreads floats.
__device__ __forceinline__ void reduce_rows(const float* partial, int n_rows, int n_cols, float* tot, float* scratch, int tid,
                                            int nthreads) {
    int cw = 32;
    while (cw < n_cols && cw < nthreads) cw <<= 1;
    const int groups = nthreads / cw;
    const int c0 = tid % cw, grp = tid / cw;
    for (int cb = 0; cb < n_cols; cb += cw) {
        const int c = cb + c0;
        float a0 = 0.f, a1 = 0.f;
        if (c < n_cols) {
            int r = grp;
            for (; r + groups < n_rows; r += 2 * groups) {
                a0 += partial[(long)r * n_cols + c];
                a1 += partial[(long)(r + groups) * n_cols + c];
            }
            if (r < n_rows) a0 += partial[(long)r * n_cols + c];
        }
        scratch[tid] = a0 + a1;
        __syncthreads();
        if (grp == 0 && c < n_cols) {
            float t = 0.f;
            for (int q = 0; q < groups; ++q) t += scratch[q * cw + c0];
            tot[c] = t;
        }
        __syncthreads();
    }
}

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

__global__ void __launch_bounds__(1024) n3_reduce_sums_kernel(ReduceK q) {      // R1
    __shared__ float tot[3 * 32 + 1];
    __shared__ float scratch[1024];
    const int H = q.H, tid = threadIdx.x;
    reduce_rows(q.partial, q.n_rows, 3 * H + 1, tot, scratch, tid, 1024);
    if (tid < H) {
        q.grad_beta[tid] = tot[tid];
        q.grad_gamma[tid] = tot[H + tid];
        q.gsum[tid] = tot[tid];
        q.gsum[H + tid] = tot[H + tid];
        q.grad_w_gate[tid] = tot[2 * H + tid];
    }
    if (tid == 0) q.grad_b_gate[0] = tot[3 * H];
}

__global__ void __launch_bounds__(1024) n3_reduce_msg_kernel(ReduceK q) {       // R2
    __shared__ float tot[32 * 33 + 64];
    __shared__ float scratch[1024];
    const int H = q.H, NB = H + 1, tid = threadIdx.x;
    reduce_rows(q.partial, q.n_rows, H * NB + 2 * H, tot, scratch, tid, 1024);
    // message weights [H, 3H] = [W_s | W_d | W_e]: dW_s = dW_d = dc (x) emb, dW_e from the MFMA accumulators, db = dc
    for (int t = tid; t < H * H; t += 1024) {
        const int o = t / H, k = t - o * H;
        const float gc = tot[o * NB + H];
        const float ge = gc * q.emb[k];
        float* row = q.grad_W_msg + (long)o * q.ld_w_msg;
        row[k] = ge;
        row[H + k] = ge;
        row[2 * H + k] = tot[o * NB + k];
    }
    if (tid < H) {
        q.grad_b_msg[tid] = tot[tid * NB + H];
        // d emb += (W_s + W_d)^T dc
        float acc = 0.f;
        for (int o = 0; o < H; ++o) {
            const float* row = q.W_msg + (long)o * q.ld_w_msg;
            acc = fmaf(row[tid] + row[H + tid], tot[o * NB + H], acc);
        }
        q.grad_emb[tid] += acc;
        const float gb = tot[H * NB + tid], gg = tot[H * NB + H + tid];
        q.grad_beta[tid] = gb;
        q.grad_gamma[tid] = gg;
        q.gsum[tid] = gb;
        q.gsum[H + tid] = gg;
    }
}

__global__ void __launch_bounds__(1024) n3_reduce_in_kernel(ReduceK q) {        // R3
    __shared__ float tot[32 * 33];
    __shared__ float scratch[1024];
    const int H = q.H, NB = q.DIN + 1, tid = threadIdx.x;
    reduce_rows(q.partial, q.n_rows, H * NB, tot, scratch, tid, 1024);
    for (int t = tid; t < H * q.DIN; t += 1024) {
        const int o = t / q.DIN, k = t - o * q.DIN;
        q.grad_W_in[(long)o * q.ld_w_in + k] = tot[o * NB + k];
    }
    if (tid < H) q.grad_b_in[tid] = tot[tid * NB + q.DIN];
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
    const long perb = (chunks + MAX_BWD_BLOCKS - 1) / MAX_BWD_BLOCKS;
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
    p.grad_m_sum = a->grad_m_sum; p.grad_ya = a->grad_ya; p.grad_lin = a->grad_lin;
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
#define N3_DISPATCH(KERNEL, grid, block, stream, ...)                                                                  \
    do {                                                                                                               \
        if (a->store_bf16) N3_DISPATCH_T(KERNEL, true, grid, block, stream, __VA_ARGS__);                              \
        else N3_DISPATCH_T(KERNEL, false, grid, block, stream, __VA_ARGS__);                                           \
    } while (0)

}  // namespace
}  // namespace i3d

using namespace i3d;

extern "C" int i3d_net3d_edge_supported(int hidden, int n_enc) { return supported(hidden, n_enc) ? 1 : 0; }

extern "C" long i3d_net3d_edge_stats_floats(int num_edges, int hidden) {
    if (num_edges <= 0 || hidden <= 0) return 0;
    return (long)plan_for(num_edges).fwd_tiles * 3 * hidden;
}

extern "C" long i3d_net3d_edge_bwd_floats(int num_edges, int hidden, int n_enc) {
    if (num_edges <= 0 || hidden <= 0) return 0;
    const int din = n_enc > 0 ? 2 * n_enc + 1 : 1;
    const long per = (long)hidden * (hidden + 1) + 2 * hidden;          // the widest pass (B2); B1: 3H+1, B3: H (DIN+1)
    const long per3 = (long)hidden * (din + 1);
    return (long)plan_for(num_edges).bwd_blocks * (per > per3 ? per : per3) + 4 * hidden + 4;      // ... | gsum [4H] | 1 / rows of all ranks
}

extern "C" int i3d_net3d_edge_fwd(const I3dNet3dEdgeArgs* a, void* stream_) {
    if (int rc = check_common(a)) return rc;
    I3D_CHECK_ARG(a->stats && a->msg && a->m_sum, "null forward buffer");
    hipStream_t stream = (hipStream_t)stream_;
    const Plan pl = plan_for(a->num_edges);
    EdgeK p = kernel_args(a);
    p.rows_per_block = pl.fwd_rows_per_tile;
    p.partial = a->stats;
    const int H = a->hidden;
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
    I3D_CHECK_ARG(a->grad_m_sum && a->grad_ya && a->grad_lin && a->partial, "null backward buffer");
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
    ReduceK q{};
    q.H = H; q.DIN = DIN; q.n_rows = pl.bwd_blocks; q.ld_w_in = a->ld_w_in; q.ld_w_msg = a->ld_w_msg;
    q.partial = a->partial; q.emb = a->emb; q.W_msg = a->W_msg;
    q.grad_w_gate = a->grad_w_gate; q.grad_b_gate = a->grad_b_gate; q.grad_W_msg = a->grad_W_msg; q.grad_b_msg = a->grad_b_msg;
    q.grad_emb = a->grad_emb; q.grad_W_in = a->grad_W_in; q.grad_b_in = a->grad_b_in;

    N3_DISPATCH(n3_bwd_sums_kernel, dim3(pl.bwd_blocks), dim3(TB), stream, p);
    I3D_CHECK_LAUNCH();
    q.gsum = gsum; q.grad_gamma = a->grad_gamma_msg; q.grad_beta = a->grad_beta_msg;
    hipLaunchKernelGGL(n3_reduce_sums_kernel, dim3(1), dim3(1024), 0, stream, q);
    I3D_CHECK_LAUNCH();
    const bool synced = collectives() != nullptr;
    float* inv_rows_dev = gsum + 4 * H;
    I3D_CHECK_ARG(!synced || 2 * H < 128, "hidden too wide for the sum exchange");
    if (synced) {
        if (int rc = sync_backward_sums(gsum, 2 * H, a->num_edges, inv_rows_dev, stream_)) return rc;
        p.inv_rows_dev = inv_rows_dev;
    }
    p.gsum_msg = gsum;
    N3_DISPATCH(n3_bwd_msg_kernel, dim3(pl.bwd_blocks), dim3(TB), stream, p);
    I3D_CHECK_LAUNCH();
    N3_DISPATCH(n3_bwd_post_kernel, dim3(pl.bwd_blocks), dim3(TB), stream, p);
    I3D_CHECK_LAUNCH();
    q.gsum = gsum + 2 * H; q.grad_gamma = a->grad_gamma_in; q.grad_beta = a->grad_beta_in;
    hipLaunchKernelGGL(n3_reduce_msg_kernel, dim3(1), dim3(1024), 0, stream, q);
    I3D_CHECK_LAUNCH();
    if (synced) {
        if (int rc = sync_backward_sums(gsum + 2 * H, 2 * H, a->num_edges, inv_rows_dev, stream_)) return rc;
    }
    p.gsum_in = gsum + 2 * H;
    N3_DISPATCH(n3_bwd_in_kernel, dim3(pl.bwd_blocks), dim3(TB), stream, p);
    I3D_CHECK_LAUNCH();
    hipLaunchKernelGGL(n3_reduce_in_kernel, dim3(1), dim3(1024), 0, stream, q);
    I3D_CHECK_LAUNCH();
    return I3D_OK;
}
