// Column statistics, BatchNorm1d (train / eval, forward / backward) and the activation fused in front of it.
//
// Replaces nn.BatchNorm1d + the activation of FCLayer (reference models/base_layers.py:100-111): the
// reference's order is Linear -> activation -> BatchNorm, batch statistics over ALL rows of the batch
// (edges for pretrans, nodes for posttrans, graphs for the head), momentum m, eps 1e-5, unbiased
// running_var.  All column reductions are two-stage and deterministic: stage 1 reduces a row chunk per
// workgroup (fp32, shifted by row 0 so E[x^2]-E[x]^2 does not cancel), stage 2 sums the chunk partials in
// fp64 and finalises.  HBM-bound: one read (+ one write when an activation is fused) per pass.
#include <stdlib.h>

#include <atomic>

#include "common.h"
#include "peer.h"

namespace i3d {

constexpr int MAX_PARTIAL_BLOCKS = 256;      // capacity of the partial buffer / of the in-launch finalisation (raise for A/B runs)
// row chunks used: 256 = one workgroup per CU.  Measured at
// batch 512 (tools/ab_partial_blocks.sh, profiles/r02_ab_partial_blocks.txt): 512 / 1024 chunks make every reduction
// SLOWER (apply+colsum 22.4 -> 24.2 -> 27.2 us with the in-launch finalisation, 12.1 -> 10.5 -> 13.1 us without) - the
// per-workgroup epilogue (LDS combine, uncached partial stores) and the longer finalisation outweigh the extra waves.
static int partial_blocks() { return MAX_PARTIAL_BLOCKS; }
constexpr int RU = 4;                     // rows per thread in flight in the streaming kernels   // ~ one partial block per CU; stage 2 reduces them with 8 lanes per column

struct Chunking {
    int tpr;       // threads per row (column vectors handled in parallel)
    int rl;        // row lanes per block
    int rpb;       // rows per block
    int nblk;      // blocks along rows
    int ncolblk;   // blocks along columns
    int cv;        // column vectors per row
    int V;         // floats per column vector
};

static Chunking make_chunking(int rows, int feat) {
    Chunking c;
    c.V = (feat % 4 == 0) ? 4 : 1;
    c.cv = feat / c.V;
    c.tpr = c.cv < 256 ? c.cv : 256;
    c.rl = 256 / c.tpr;
    c.ncolblk = cdiv(c.cv, c.tpr);
    int rpb = cdiv(rows, partial_blocks());
    int min_rpb = c.rl * 4;
    if (rpb < min_rpb) rpb = min_rpb;
    c.rpb = rpb;
    c.nblk = rows > 0 ? cdiv(rows, rpb) : 0;
    return c;
}

enum { MODE_STATS = 0, MODE_BN_BWD = 1, MODE_COLSUM = 2 };

// The pivot of the shifted sums of column c: a value near the column's data that EVERY workgroup and the finalisation derive from
// row 0 alike - also when the activation is applied in place and row 0 may already hold act(pre) when a workgroup reads it.
// ReLU / LeakyReLU (the in-place activations): max(v, 0) - the same number from pre and from act(pre) (act(act(v)) would not be,
// for LeakyReLU: round 4, a wrong mean); identical to relu(v), the pivot used so far.  The others are never applied in place (their
// derivative needs the pre-activation).
template <bool GA = true>      // (GA false: the kernel was launched for none / ReLU / LeakyReLU only, common.h)
__device__ __forceinline__ float stats_pivot(float v, int act) {
    if (act == I3D_ACT_RELU || act == I3D_ACT_LEAKY_RELU) return fmaxf(v, 0.f);
    return GA ? apply_act(v, act) : v;
}

constexpr int FIN_COLS = 8, FIN_LANES = 32;    // stage 2: 8 columns x 32 partial-lanes per workgroup (25 of them at F=200)
constexpr int WS_HEADER = 128;                 // bytes in front of the partial rows: the arrival / done counters

// Stage 2 inside the stage-1 launch.  Every workgroup publishes its partial row (agent-scope stores), waits for its
// own stores (vmcnt(0)), then takes an arrival ticket.  The LAST R arrivers (R = number of 8-column groups, <= grid
// size) wait until all partial rows are published - they are the last to arrive, the wait is short, and the grid
// (<= 256 x ncolblk workgroups) is always fully resident, so it cannot deadlock - and each reduces its column groups in
// a fixed order (deterministic, as the separate finalise kernels are).  Saves one launch per reduction, 53 per
// training step (see fused_final() for the measured trade-off).  The counters live at the head of the workspace, start at zero and are reset by
// the last reducer, so the workspace must be zero-initialised once and must not be shared by concurrent streams.
struct Final {
    int kind;              // 0: batch statistics (mean / invstd / running stats / fp64 sums)   1: pair of column sums
    int feat, rows, act;
    float eps, momentum;
    const float* pre_row0; // statistics: row 0 of the pre-activation (the shift of the shifted sums)
    float* mean;
    float* invstd;
    float* running_mean;
    float* running_var;
    float* out1;           // pair: fp32 results (may be null)
    float* out2;
    double* sums_out;      // fp64 results for the synchronised-BN all-reduce (may be null)
    long long* batches_tracked;   // BatchNorm1d.num_batches_tracked (int64), incremented with the running statistics; may be null
    unsigned* counters;    // null: stage 2 is a separate launch (I3D_FUSED_FINAL=0)
    // synchronised BatchNorm through the peer-write exchange (peer.h), inside the in-launch finalisation: the reducer of a
    // column block writes its fp64 {s1, s2, rows} into every rank's mailbox, waits for every rank's and finishes from the
    // sums over the ranks (rank order) - statistics: mean / invstd / running statistics of the global batch; pair: out1 /
    // out2 keep this rank's share, g1 / g2 / ginv receive the global sums as fp32 and 1 / global rows
    int peer_on;
    float* g1;
    float* g2;
    float* ginv;
    PeerDev peer;
};

__device__ __forceinline__ float ld_agent(const float* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_agent(float* p, float v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <bool GA>
__device__ __forceinline__ void finalize_column(const Final& f, int c, double s1, double s2) {
    if (f.kind == 0) {
        const double shift = (double)stats_pivot<GA>(f.pre_row0[c], f.act);
        const double n = (double)f.rows;
        if (f.sums_out != nullptr) {   // synchronised BN: hand un-shifted fp64 sums to the all-reduce
            f.sums_out[c] = s1 + n * shift;
            f.sums_out[f.feat + c] = s2 + 2.0 * shift * s1 + n * shift * shift;
            if (c == 0) f.sums_out[2 * f.feat] = n;
            return;
        }
        const double m = s1 / n;
        double var = s2 / n - m * m;
        if (var < 0.0) var = 0.0;
        f.mean[c] = (float)(shift + m);
        f.invstd[c] = (float)(1.0 / sqrt(var + (double)f.eps));
        if (f.running_mean != nullptr) {
            const double unbiased = f.rows > 1 ? var * n / (n - 1.0) : var;
            f.running_mean[c] = (float)((1.0 - f.momentum) * (double)f.running_mean[c] + f.momentum * (shift + m));
            f.running_var[c] = (float)((1.0 - f.momentum) * (double)f.running_var[c] + f.momentum * unbiased);
        }
        if (c == 0 && f.batches_tracked != nullptr) *f.batches_tracked += 1;
    } else {
        if (f.sums_out != nullptr) {
            f.sums_out[c] = s1;
            f.sums_out[f.feat + c] = s2;
        }
        if (f.out1 != nullptr) f.out1[c] = (float)s1;
        if (f.out2 != nullptr) f.out2[c] = (float)s2;
    }
}

// the part of finalize_column that follows the exchange: T1, T2, N = sums over the ranks of what finalize_column_local put
__device__ __forceinline__ void finalize_column_global(const Final& f, int c, double T1, double T2, double N) {
    if (f.kind == 0) {          // == stats_from_sums_kernel
        const double m = T1 / N;
        double var = T2 / N - m * m;
        if (var < 0.0) var = 0.0;
        f.mean[c] = (float)m;
        f.invstd[c] = (float)(1.0 / sqrt(var + (double)f.eps));
        if (f.running_mean != nullptr) {
            const double unbiased = N > 1.0 ? var * N / (N - 1.0) : var;
            f.running_mean[c] = (float)((1.0 - f.momentum) * (double)f.running_mean[c] + f.momentum * m);
            f.running_var[c] = (float)((1.0 - f.momentum) * (double)f.running_var[c] + f.momentum * unbiased);
        }
        if (c == 0 && f.batches_tracked != nullptr) *f.batches_tracked += 1;
    } else {                    // == sums_to_float_kernel
        f.g1[c] = (float)T1;
        f.g2[c] = (float)T2;
        if (c == 0) f.ginv[0] = (float)(1.0 / N);
    }
}

// this rank's contribution to the exchange (un-shifted sums for the statistics); the pair's local results are stored here
template <bool GA>
__device__ __forceinline__ void finalize_column_local(const Final& f, int c, double s1, double s2) {
    const double n = (double)f.rows;
    double a = s1, b = s2;
    if (f.kind == 0) {
        const double shift = (double)stats_pivot<GA>(f.pre_row0[c], f.act);
        a = s1 + n * shift;
        b = s2 + 2.0 * shift * s1 + n * shift * shift;
    } else {
        if (f.out1 != nullptr) f.out1[c] = (float)s1;
        if (f.out2 != nullptr) f.out2[c] = (float)s2;
    }
    for (int p = 0; p < f.peer.world; ++p) {
        peer_put_f64(f.peer, p, c, a);
        peer_put_f64(f.peer, p, f.feat + c, b);
        peer_put_f64(f.peer, p, 2 * f.feat + c, n);
    }
}

// called by ALL threads of every workgroup after the partial row of the workgroup has been stored with st_agent
template <bool GA>
__device__ __forceinline__ void arrive_and_finalize(const Final& f, const float* partial, int nblk) {
    __shared__ unsigned s_ticket;
    __shared__ double s_red[2][FIN_LANES][FIN_COLS];
    __shared__ double s_peer[3][PEER_MAX_WORLD][FIN_COLS];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's partial stores have left (agent-scope, write-through)
    __syncthreads();
    const unsigned total = gridDim.x * gridDim.y;
    if (threadIdx.x == 0)
        s_ticket = __hip_atomic_fetch_add(&f.counters[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const unsigned ticket = s_ticket;
    const unsigned ncb = (unsigned)((f.feat + FIN_COLS - 1) / FIN_COLS);
    const unsigned R = ncb < total ? ncb : total;
    if (ticket < total - R) return;
    if (threadIdx.x == 0) {        // bounded: a poisoned counter must not hang the device (results are then wrong, not stuck)
        int spins = 0;
        while (__hip_atomic_load(&f.counters[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < total && ++spins < (1 << 24))
            __builtin_amdgcn_s_sleep(2);
    }
    __syncthreads();
    const int cx = threadIdx.x & (FIN_COLS - 1), ly = threadIdx.x / FIN_COLS;
    for (unsigned cb = ticket - (total - R); cb < ncb; cb += R) {
        const int c = (int)cb * FIN_COLS + cx;
        double a1 = 0.0, a2 = 0.0;
        if (c < f.feat) {
            // all loads of the lane first (<= MAX_PARTIAL_BLOCKS / FIN_LANES = 8 partial rows, 16 uncached loads in
            // flight), then the sums: a load-add loop pays the ~1.5 us agent-scope latency once per iteration
            constexpr int NB = MAX_PARTIAL_BLOCKS / FIN_LANES;
            float v1[NB], v2[NB];
#pragma unroll
            for (int k = 0; k < NB; ++k) {
                const int b = min(ly + k * FIN_LANES, nblk - 1);
                v1[k] = ld_agent(partial + (long)b * 2 * f.feat + c);
                v2[k] = ld_agent(partial + (long)b * 2 * f.feat + f.feat + c);
            }
#pragma unroll
            for (int k = 0; k < NB; ++k) {
                if (ly + k * FIN_LANES < nblk) { a1 += (double)v1[k]; a2 += (double)v2[k]; }
            }
        }
        s_red[0][ly][cx] = a1;
        s_red[1][ly][cx] = a2;
        __syncthreads();
        if (ly == 0 && c < f.feat) {
            double s1 = 0.0, s2 = 0.0;
#pragma unroll
            for (int k = 0; k < FIN_LANES; ++k) { s1 += s_red[0][k][cx]; s2 += s_red[1][k][cx]; }
            if (f.peer_on) finalize_column_local<GA>(f, c, s1, s2);
            else finalize_column<GA>(f, c, s1, s2);
        }
        if (f.peer_on) {        // (uniform) the column block's exchange: lane `ly` < world fetches rank ly's triple of its column
            if (ly < f.peer.world && c < f.feat) {
                double t[3];
                peer_get3_f64(f.peer, ly, c, f.feat + c, 2 * f.feat + c, t);
                s_peer[0][ly][cx] = t[0]; s_peer[1][ly][cx] = t[1]; s_peer[2][ly][cx] = t[2];
            }
            __syncthreads();
            if (ly == 0 && c < f.feat) {
                double T1 = 0.0, T2 = 0.0, N = 0.0;
                for (int q = 0; q < f.peer.world; ++q) {      // rank order: the same bits on every rank
                    T1 += s_peer[0][q][cx];
                    T2 += s_peer[1][q][cx];
                    N += s_peer[2][q][cx];
                }
                finalize_column_global(f, c, T1, T2, N);
            }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const unsigned d = __hip_atomic_fetch_add(&f.counters[1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (d == R - 1) {          // every reducer is done reading: re-arm for the next launch
            __hip_atomic_store(&f.counters[1], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&f.counters[0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

struct ReduceArgs {
    const float* a;        // STATS: pre          BN_BWD: grad_y      COLSUM: x
    const float* b;        // STATS: -            BN_BWD: x           COLSUM: row weights (or null)
    float* out;            // STATS: x = act(pre) (or null)
    const float* mean;     // BN_BWD
    const float* invstd;   // BN_BWD
    const float* gamma;    // BN_BWD with post_act
    const float* beta;     // BN_BWD with post_act
    int rows, feat, act, post_act;
    float* partial;        // [nblk][2][feat]
    int b_bf16;            // MODE_BN_BWD: b (the BatchNorm input x) is stored as bf16 (V = 4)
    int lda;               // MODE_COLSUM: row pitch of a in floats (0: feat) - a column block of a wider matrix
};

// 4 consecutive values of a row operand that is stored as fp32 or (x_bf16: the bf16 mode's message storage) as bf16
__device__ __forceinline__ float4 load4_maybe_bf16(const float* base, long off, int is_bf16) {
    if (is_bf16) {
        const uint2 t = *reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(base) + off);
        return make_float4(__uint_as_float(t.x << 16), __uint_as_float(t.x & 0xffff0000u), __uint_as_float(t.y << 16),
                           __uint_as_float(t.y & 0xffff0000u));
    }
    return *reinterpret_cast<const float4*>(base + off);
}

template <int MODE, int V, bool GA>      // GA: see apply_act_c (common.h)
__global__ void __launch_bounds__(256) colreduce_partial_kernel(ReduceArgs g, Chunking ch, Final fin) {
    I3D_CHAIN_PRIO();
    __shared__ float sm[2][256 * 4];
    const int t = threadIdx.x;
    const int cl = t % ch.tpr, rlane = t / ch.tpr;
    const int cvi = blockIdx.y * ch.tpr + cl;
    const bool active = rlane < ch.rl && cvi < ch.cv;
    const int F = g.feat;
    const int c0 = cvi * V;
    float a1[V], a2[V], shift[V], mu[V], is[V], ga[V], be[V];
#pragma unroll
    for (int i = 0; i < V; ++i) { a1[i] = 0.f; a2[i] = 0.f; shift[i] = 0.f; mu[i] = 0.f; is[i] = 1.f; ga[i] = 1.f; be[i] = 0.f; }
    if (active) {
        if (MODE == MODE_STATS) {
#pragma unroll
            for (int i = 0; i < V; ++i) shift[i] = stats_pivot<GA>(g.a[c0 + i], g.act);   // row 0
        }
        if (MODE == MODE_BN_BWD) {
#pragma unroll
            for (int i = 0; i < V; ++i) {
                mu[i] = g.mean[c0 + i];
                is[i] = g.invstd[c0 + i];
                if (g.post_act != I3D_ACT_NONE) { ga[i] = g.gamma[c0 + i]; be[i] = g.beta[c0 + i]; }
            }
        }
        const int r_begin = blockIdx.x * ch.rpb;
        const int r_end = min(g.rows, r_begin + ch.rpb);
        // RU rows per thread in flight (clamped row index: the loads are unconditional, the accumulation is masked and
        // keeps the row order r, r + rl, r + 2 rl, ...): a one-row loop left one 16-byte load in flight per lane
        for (int r0 = r_begin + rlane; r0 < r_end; r0 += ch.rl * RU) {
            float xs[RU][V], ys[RU][V], ws[RU];
#pragma unroll
            for (int u = 0; u < RU; ++u) {
                const int r = min(r0 + u * ch.rl, r_end - 1);
                const long off = (long)r * F + c0;
                const long off_a = (MODE == MODE_COLSUM && g.lda > 0) ? (long)r * g.lda + c0 : off;
                if (V == 4) {
                    float4 xx = *reinterpret_cast<const float4*>(g.a + off_a);
                    xs[u][0] = xx.x; xs[u][1 % V] = xx.y; xs[u][2 % V] = xx.z; xs[u][3 % V] = xx.w;
                } else {
                    xs[u][0] = g.a[off_a];
                }
                if (MODE == MODE_BN_BWD) {
                    if (V == 4) {
                        float4 yy = load4_maybe_bf16(g.b, off, g.b_bf16);
                        ys[u][0] = yy.x; ys[u][1 % V] = yy.y; ys[u][2 % V] = yy.z; ys[u][3 % V] = yy.w;
                    } else {
                        ys[u][0] = g.b[off];
                    }
                }
                if (MODE == MODE_COLSUM) ws[u] = g.b != nullptr ? g.b[r] : 1.f;
            }
#pragma unroll
            for (int u = 0; u < RU; ++u) {
                const int r = r0 + u * ch.rl;
                if (r >= r_end) continue;
                const long off = (long)r * F + c0;
                float* x = xs[u];
                if (MODE == MODE_STATS) {
#pragma unroll
                    for (int i = 0; i < V; ++i) {
                        x[i] = apply_act_c<GA>(x[i], g.act);
                        float d = x[i] - shift[i];
                        a1[i] += d;
                        a2[i] += d * d;
                    }
                    if (g.out != nullptr && g.act != I3D_ACT_NONE) {
                        if (V == 4) *reinterpret_cast<float4*>(g.out + off) = make_float4(x[0], x[1 % V], x[2 % V], x[3 % V]);
                        else g.out[off] = x[0];
                    }
                } else if (MODE == MODE_BN_BWD) {
#pragma unroll
                    for (int i = 0; i < V; ++i) {
                        float xh = (ys[u][i] - mu[i]) * is[i];
                        float dy = x[i];
                        if (g.post_act != I3D_ACT_NONE) dy *= act_grad_c<GA>(xh * ga[i] + be[i], g.post_act);
                        a1[i] += dy;
                        a2[i] += dy * xh;
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < V; ++i) a1[i] += x[i] * ws[u];
                }
            }
        }
    }
    // reduce over the row lanes of the block
#pragma unroll
    for (int i = 0; i < V; ++i) { sm[0][t * V + i] = a1[i]; sm[1][t * V + i] = a2[i]; }
    __syncthreads();
    if (rlane == 0 && cvi < ch.cv) {
        float s1[V], s2[V];
#pragma unroll
        for (int i = 0; i < V; ++i) { s1[i] = 0.f; s2[i] = 0.f; }
        for (int k = 0; k < ch.rl; ++k) {
            const int tt = k * ch.tpr + cl;
#pragma unroll
            for (int i = 0; i < V; ++i) { s1[i] += sm[0][tt * V + i]; s2[i] += sm[1][tt * V + i]; }
        }
        float* p = g.partial + (long)blockIdx.x * 2 * F + c0;
#pragma unroll
        for (int i = 0; i < V; ++i) { st_agent(p + i, s1[i]); st_agent(p + F + i, s2[i]); }
    }
    if (fin.counters != nullptr) arrive_and_finalize<GA>(fin, g.partial, gridDim.x);
}

// ---- stage 2 ----------------------------------------------------------------------------------------
// 256 threads = 8 columns x 32 partial-lanes: reads of the chunk partials, fp64 accumulation, one LDS
// hop.  (A one-thread-per-column loop over the partials was latency-bound: 130-150 us per call, 40 % of the step.)

__device__ __forceinline__ bool reduce_partials(const float* __restrict__ partial, int nblk, int feat, int& c,
                                                double& s1, double& s2) {
    __shared__ double sm[2][FIN_LANES][FIN_COLS];
    const int cx = threadIdx.x & (FIN_COLS - 1), ly = threadIdx.x / FIN_COLS;
    c = blockIdx.x * FIN_COLS + cx;
    double a1 = 0.0, a2 = 0.0;
    if (c < feat) {
        for (int b = ly; b < nblk; b += FIN_LANES) {
            a1 += (double)partial[(long)b * 2 * feat + c];
            a2 += (double)partial[(long)b * 2 * feat + feat + c];
        }
    }
    sm[0][ly][cx] = a1;
    sm[1][ly][cx] = a2;
    __syncthreads();
    if (ly != 0 || c >= feat) return false;
    s1 = 0.0; s2 = 0.0;
#pragma unroll
    for (int k = 0; k < FIN_LANES; ++k) { s1 += sm[0][k][cx]; s2 += sm[1][k][cx]; }
    return true;
}

__global__ void stats_final_kernel(const float* __restrict__ partial, int nblk, const float* __restrict__ pre_row0,
                                   int act, int rows, int feat, float eps, float momentum, float* mean, float* invstd,
                                   float* running_mean, float* running_var, double* sums_out,
                                   long long* batches_tracked) {
    I3D_CHAIN_PRIO();
    int c;
    double s1, s2;
    if (!reduce_partials(partial, nblk, feat, c, s1, s2)) return;
    double shift = (double)stats_pivot(pre_row0[c], act);
    double n = (double)rows;
    if (sums_out != nullptr) {   // synchronised BN: hand un-shifted fp64 sums to the all-reduce
        sums_out[c] = s1 + n * shift;
        sums_out[feat + c] = s2 + 2.0 * shift * s1 + n * shift * shift;
        if (c == 0) sums_out[2 * feat] = n;
        return;
    }
    double m = s1 / n;
    double var = s2 / n - m * m;
    if (var < 0.0) var = 0.0;
    mean[c] = (float)(shift + m);
    invstd[c] = (float)(1.0 / sqrt(var + (double)eps));
    if (running_mean != nullptr) {
        double unbiased = rows > 1 ? var * n / (n - 1.0) : var;
        running_mean[c] = (float)((1.0 - momentum) * (double)running_mean[c] + momentum * (shift + m));
        running_var[c] = (float)((1.0 - momentum) * (double)running_var[c] + momentum * unbiased);
    }
    if (c == 0 && batches_tracked != nullptr) *batches_tracked += 1;
}

__global__ void stats_from_sums_kernel(const double* __restrict__ sums, int feat, float eps, float momentum, float* mean,
                                       float* invstd, float* running_mean, float* running_var,
                                       long long* batches_tracked = nullptr) {
    I3D_CHAIN_PRIO();
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= feat) return;
    if (c == 0 && batches_tracked != nullptr) *batches_tracked += 1;
    double n = sums[2 * feat];
    double m = sums[c] / n;
    double var = sums[feat + c] / n - m * m;
    if (var < 0.0) var = 0.0;
    mean[c] = (float)m;
    invstd[c] = (float)(1.0 / sqrt(var + (double)eps));
    if (running_mean != nullptr) {
        double unbiased = n > 1.0 ? var * n / (n - 1.0) : var;
        running_mean[c] = (float)((1.0 - momentum) * (double)running_mean[c] + momentum * m);
        running_var[c] = (float)((1.0 - momentum) * (double)running_var[c] + momentum * unbiased);
    }
}

// sums of the two partial columns -> out1[feat], out2[feat] (fp32) or fp64 sums_out[2*feat]
__global__ void pair_final_kernel(const float* __restrict__ partial, int nblk, int feat, float* out1, float* out2,
                                  double* sums_out) {
    I3D_CHAIN_PRIO();
    int c;
    double s1, s2;
    if (!reduce_partials(partial, nblk, feat, c, s1, s2)) return;
    if (sums_out != nullptr) {
        sums_out[c] = s1;
        sums_out[feat + c] = s2;
    }
    if (out1 != nullptr) out1[c] = (float)s1;
    if (out2 != nullptr) out2[c] = (float)s2;
}

__global__ void invstd_from_var_kernel(const float* __restrict__ rv, int feat, float eps, float* __restrict__ o) {
    I3D_CHAIN_PRIO();
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < feat) o[c] = 1.f / sqrtf(rv[c] + eps);
}

// fp64 all-reduced sums {sum dy, sum dy*xhat, count} -> fp32 scratch {.., .., 1/count}
__global__ void sums_to_float_kernel(const double* __restrict__ sums, int feat, float* out1, float* out2) {
    I3D_CHAIN_PRIO();
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= feat) return;
    out1[c] = (float)sums[c];
    out2[c] = (float)sums[feat + c];
    if (c == 0) out2[feat] = (float)(1.0 / sums[2 * feat]);
}

// ---- elementwise passes ------------------------------------------------------------------------------
struct ApplyArgs {
    const float* x;
    const float* mean;
    const float* invstd;      // train: invstd      eval: running_var (invstd computed on the fly)
    const float* gamma;
    const float* beta;
    const float* residual;
    float* y;
    long items;               // rows * feat / V
    int feat, post_act, eval_mode;
    float eps;
};

// Elementwise passes: a thread owns one column vector (its per-column parameters live in registers) and RU rows per
// trip, rows of a block are consecutive: no index division and no parameter gathers in the loop, RU 16-byte loads per
// operand in flight.  (The first version walked a flat grid-stride index and re-read 4 parameters per element: ~1.7 TB/s.)
struct RowTiling {
    int tpr, rl, cv, ncolblk;   // threads per row, row lanes per block, column vectors per row, blocks along columns
};

static RowTiling make_row_tiling(int feat, int V) {
    RowTiling t;
    t.cv = feat / V;
    t.tpr = t.cv < 256 ? t.cv : 256;
    t.rl = 256 / t.tpr;
    t.ncolblk = cdiv(t.cv, t.tpr);
    return t;
}

template <int V, bool GA>
__global__ void __launch_bounds__(256) bn_apply_kernel(ApplyArgs g, RowTiling rt, int rows) {
    I3D_CHAIN_PRIO();
    const int cl = threadIdx.x % rt.tpr, rlane = threadIdx.x / rt.tpr;
    const int cvi = blockIdx.y * rt.tpr + cl;
    if (rlane >= rt.rl || cvi >= rt.cv) return;
    const int c0 = cvi * V, F = g.feat;
    float sc[V], sh[V];
#pragma unroll
    for (int i = 0; i < V; ++i) {
        const float is = g.eval_mode ? 1.f / sqrtf(g.invstd[c0 + i] + g.eps) : g.invstd[c0 + i];
        sc[i] = is;
        sh[i] = g.mean[c0 + i];
    }
    float ga[V], be[V];
#pragma unroll
    for (int i = 0; i < V; ++i) { ga[i] = g.gamma[c0 + i]; be[i] = g.beta[c0 + i]; }
    const int r0 = blockIdx.x * rt.rl * RU + rlane;
    float x[RU][V], r[RU][V];
#pragma unroll
    for (int u = 0; u < RU; ++u) {
        const int row = min(r0 + u * rt.rl, rows - 1);
        const long off = (long)row * F + c0;
        if (V == 4) {
            float4 xx = *reinterpret_cast<const float4*>(g.x + off);
            x[u][0] = xx.x; x[u][1 % V] = xx.y; x[u][2 % V] = xx.z; x[u][3 % V] = xx.w;
            if (g.residual) {
                float4 rr = *reinterpret_cast<const float4*>(g.residual + off);
                r[u][0] = rr.x; r[u][1 % V] = rr.y; r[u][2 % V] = rr.z; r[u][3 % V] = rr.w;
            }
        } else {
            x[u][0] = g.x[off];
            if (g.residual) r[u][0] = g.residual[off];
        }
    }
#pragma unroll
    for (int u = 0; u < RU; ++u) {
        const int row = r0 + u * rt.rl;
        if (row >= rows) continue;
        const long off = (long)row * F + c0;
#pragma unroll
        for (int i = 0; i < V; ++i) {
            float v = (x[u][i] - sh[i]) * sc[i] * ga[i] + be[i];
            v = apply_act_c<GA>(v, g.post_act);
            if (g.residual) v += r[u][i];
            x[u][i] = v;
        }
        if (V == 4) *reinterpret_cast<float4*>(g.y + off) = make_float4(x[u][0], x[u][1 % V], x[u][2 % V], x[u][3 % V]);
        else g.y[off] = x[u][0];
    }
}

struct BwdApplyArgs {
    const float* grad_y;
    const float* x;
    const float* pre;
    const float* mean;
    const float* invstd;      // eval: running_var
    const float* gamma;
    const float* beta;
    const float* sum_dy;      // [feat]  (train only)
    const float* sum_dy_xhat; // [feat]
    float* grad_pre;
    int ld_out;               // row pitch of grad_pre in floats (>= feat): the result may be a column block of a wider matrix
    float* zero_out;          // [feat] or null: filled with zeros by the first row block (see i3d_bn_bwd: exact-zero bias gradient)
    const float* inv_n_ptr;   // device 1/N (synchronised BN) or null -> inv_n
    long items;
    int feat, act, post_act, eval_mode;
    float inv_n, eps;
    int x_bf16;               // x is stored as bf16 (feat % 4 == 0)
};

template <int V, bool GA>
__global__ void __launch_bounds__(256) bn_bwd_apply_kernel(BwdApplyArgs g, RowTiling rt, int rows) {
    I3D_CHAIN_PRIO();
    const int cl = threadIdx.x % rt.tpr, rlane = threadIdx.x / rt.tpr;
    const int cvi = blockIdx.y * rt.tpr + cl;
    if (rlane >= rt.rl || cvi >= rt.cv) return;
    const int c0 = cvi * V, F = g.feat;
    if (g.zero_out != nullptr && blockIdx.x == 0 && rlane == 0) {
#pragma unroll
        for (int i = 0; i < V; ++i) g.zero_out[c0 + i] = 0.f;
    }
    const float inv_n = g.inv_n_ptr ? g.inv_n_ptr[0] : g.inv_n;
    float mu[V], is[V], ga[V], be[V], k1[V], k2[V];
#pragma unroll
    for (int i = 0; i < V; ++i) {
        const int c = c0 + i;
        is[i] = g.eval_mode ? 1.f / sqrtf(g.invstd[c] + g.eps) : g.invstd[c];
        mu[i] = g.mean[c];
        ga[i] = g.gamma[c];
        be[i] = g.beta[c];
        k1[i] = g.eval_mode ? 0.f : g.sum_dy[c] * inv_n;
        k2[i] = g.eval_mode ? 0.f : g.sum_dy_xhat[c] * inv_n;
    }
    const int r0 = blockIdx.x * rt.rl * RU + rlane;
    float dy[RU][V], x[RU][V], p[RU][V];
#pragma unroll
    for (int u = 0; u < RU; ++u) {
        const int row = min(r0 + u * rt.rl, rows - 1);
        const long off = (long)row * F + c0;
        if (V == 4) {
            float4 a = *reinterpret_cast<const float4*>(g.grad_y + off);
            float4 b = load4_maybe_bf16(g.x, off, g.x_bf16);
            dy[u][0] = a.x; dy[u][1 % V] = a.y; dy[u][2 % V] = a.z; dy[u][3 % V] = a.w;
            x[u][0] = b.x; x[u][1 % V] = b.y; x[u][2 % V] = b.z; x[u][3 % V] = b.w;
            if (g.pre) {
                float4 c = *reinterpret_cast<const float4*>(g.pre + off);
                p[u][0] = c.x; p[u][1 % V] = c.y; p[u][2 % V] = c.z; p[u][3 % V] = c.w;
            }
        } else {
            dy[u][0] = g.grad_y[off];
            x[u][0] = g.x[off];
            if (g.pre) p[u][0] = g.pre[off];
        }
    }
#pragma unroll
    for (int u = 0; u < RU; ++u) {
        const int row = r0 + u * rt.rl;
        if (row >= rows) continue;
        const long off = (long)row * F + c0;
#pragma unroll
        for (int i = 0; i < V; ++i) {
            float xh = (x[u][i] - mu[i]) * is[i];
            float d = dy[u][i];
            if (g.post_act != I3D_ACT_NONE) d *= act_grad_c<GA>(xh * ga[i] + be[i], g.post_act);
            float gx;
            if (g.eval_mode) gx = d * ga[i] * is[i];
            else gx = ga[i] * is[i] * (d - k1[i] - xh * k2[i]);
            if (g.act != I3D_ACT_NONE) gx *= act_grad_c<GA>(g.pre ? p[u][i] : x[u][i], g.act);   // relu'(pre) == relu'(x)
            dy[u][i] = gx;
        }
        const long oo = (long)row * g.ld_out + c0;
        if (V == 4) *reinterpret_cast<float4*>(g.grad_pre + oo) = make_float4(dy[u][0], dy[u][1 % V], dy[u][2 % V], dy[u][3 % V]);
        else g.grad_pre[oo] = dy[u][0];
    }
}

// bn_bwd_apply + column sums of the result (the bias gradient of the Linear in front: dL/db = sum_rows grad_pre): the
// row-chunk layout of the reduction kernels, one partial row per block, finalised by pair_final_kernel.
template <int V, bool GA>
__global__ void __launch_bounds__(256) bn_bwd_apply_colsum_kernel(BwdApplyArgs g, Chunking ch, int rows, float* partial,
                                                                  Final fin) {
    I3D_CHAIN_PRIO();
    __shared__ float sm[256 * 4];
    const int t = threadIdx.x;
    const int cl = t % ch.tpr, rlane = t / ch.tpr;
    const int cvi = blockIdx.y * ch.tpr + cl;
    const bool active = rlane < ch.rl && cvi < ch.cv;
    const int c0 = cvi * V, F = g.feat;
    float a1[V];
#pragma unroll
    for (int i = 0; i < V; ++i) a1[i] = 0.f;
    if (active) {
        const float inv_n = g.inv_n_ptr ? g.inv_n_ptr[0] : g.inv_n;
        float mu[V], is[V], ga[V], be[V], k1[V], k2[V];
#pragma unroll
        for (int i = 0; i < V; ++i) {
            const int c = c0 + i;
            is[i] = g.invstd[c];
            mu[i] = g.mean[c];
            ga[i] = g.gamma[c];
            be[i] = g.beta[c];
            k1[i] = g.sum_dy[c] * inv_n;
            k2[i] = g.sum_dy_xhat[c] * inv_n;
        }
        const int r_begin = blockIdx.x * ch.rpb;
        const int r_end = min(rows, r_begin + ch.rpb);
        for (int r0 = r_begin + rlane; r0 < r_end; r0 += ch.rl * RU) {
            float dy[RU][V], x[RU][V], p[RU][V];
#pragma unroll
            for (int u = 0; u < RU; ++u) {
                const int row = min(r0 + u * ch.rl, r_end - 1);
                const long off = (long)row * F + c0;
                if (V == 4) {
                    float4 a = *reinterpret_cast<const float4*>(g.grad_y + off);
                    float4 b = load4_maybe_bf16(g.x, off, g.x_bf16);
                    dy[u][0] = a.x; dy[u][1 % V] = a.y; dy[u][2 % V] = a.z; dy[u][3 % V] = a.w;
                    x[u][0] = b.x; x[u][1 % V] = b.y; x[u][2 % V] = b.z; x[u][3 % V] = b.w;
                    if (g.pre) {
                        float4 c = *reinterpret_cast<const float4*>(g.pre + off);
                        p[u][0] = c.x; p[u][1 % V] = c.y; p[u][2 % V] = c.z; p[u][3 % V] = c.w;
                    }
                } else {
                    dy[u][0] = g.grad_y[off];
                    x[u][0] = g.x[off];
                    if (g.pre) p[u][0] = g.pre[off];
                }
            }
#pragma unroll
            for (int u = 0; u < RU; ++u) {
                const int row = r0 + u * ch.rl;
                if (row >= r_end) continue;
                const long off = (long)row * F + c0;
#pragma unroll
                for (int i = 0; i < V; ++i) {
                    float xh = (x[u][i] - mu[i]) * is[i];
                    float d = dy[u][i];
                    if (g.post_act != I3D_ACT_NONE) d *= act_grad_c<GA>(xh * ga[i] + be[i], g.post_act);
                    float gx = ga[i] * is[i] * (d - k1[i] - xh * k2[i]);
                    if (g.act != I3D_ACT_NONE) gx *= act_grad_c<GA>(g.pre ? p[u][i] : x[u][i], g.act);
                    dy[u][i] = gx;
                    a1[i] += gx;
                }
                const long oo = (long)row * g.ld_out + c0;
                if (V == 4) *reinterpret_cast<float4*>(g.grad_pre + oo) = make_float4(dy[u][0], dy[u][1 % V], dy[u][2 % V], dy[u][3 % V]);
                else g.grad_pre[oo] = dy[u][0];
            }
        }
    }
#pragma unroll
    for (int i = 0; i < V; ++i) sm[t * V + i] = a1[i];
    __syncthreads();
    if (rlane == 0 && cvi < ch.cv) {
        float s1[V];
#pragma unroll
        for (int i = 0; i < V; ++i) s1[i] = 0.f;
        for (int k = 0; k < ch.rl; ++k) {
            const int tt = k * ch.tpr + cl;
#pragma unroll
            for (int i = 0; i < V; ++i) s1[i] += sm[tt * V + i];
        }
        float* q = partial + (long)blockIdx.x * 2 * F + c0;
#pragma unroll
        for (int i = 0; i < V; ++i) { st_agent(q + i, s1[i]); st_agent(q + F + i, 0.f); }
    }
    if (fin.counters != nullptr) arrive_and_finalize<GA>(fin, partial, gridDim.x);
}

// ---- BatchNorm backward in ONE launch (round 6) ----------------------------------------------------------------------------
// The reduction (sum dy, sum dy xhat over the rows) and the data gradient that needs those sums were two launches, each reading
// dy and x: colreduce_partial_kernel<MODE_BN_BWD> 15.6 us + bn_bwd_apply_kernel 14.1 us + a launch boundary at batch 512, on the
// dependent chain, three BatchNorms per layer (profiles/r05_step_kernel_trace_v4.txt; counters: profiles/r06_bn_pmc.txt: one wave
// per SIMD, waiting for memory 82 % of its cycles, then a ~5 us serial finalisation tail).  Here a workgroup of 512 threads owns a
// row chunk and KEEPS it in registers (RPT rows of 4 columns per thread, all loads issued up front), and the two hand-offs between
// the workgroups go through SELF-VALIDATING 8-byte words {value bits, launch tag} (one agent-scope store each; the peer exchange's
// protocol, peer.h) instead of counters:
//   1. every workgroup stores its partial row as tagged words;
//   2. workgroup b < ceil(feat / 8) is the reducer of columns [8 b, 8 b + 8): it polls the 256 x 16 words of its columns until every
//      one carries this launch's tag, adds them in fp64 in a fixed order, stores the two sums per column as tagged words (and as
//      plain floats: grad_beta / grad_gamma);
//   3. every workgroup polls the 2 feat final words (first wave, through the LDS) and writes the data gradient from its registers.
// No ticket, no arrival counter, no generation word: two data hand-offs (~1.5-2.5 us each) are the whole serial tail.  (The first
// version of this kernel went through the ticket / last-arrivers finalisation above plus a generation word: 32-35 us per launch -
// as long as the two launches it replaced; profiles/r06_ab_onelaunch.txt.)  dy and x are read once, not twice.  Same expressions as
// the two-pass kernels; the row sums are taken in another (fixed) order.  The tag is a process-wide launch counter (never 0: the
// workspace starts zeroed), so words of earlier launches - of any shape, on any stream's workspace - never validate.
//
// Residency: every workgroup must be resident for the polls to end.  The grid is <= 256 workgroups of 8 waves with <= 128 VGPRs
// (amdgpu_waves_per_eu) and < 32 KB of LDS: two fit on a CU, so two such launches on two streams always fit the 256 CUs together
// - the 2D chain and the 3D network's stream are the only concurrent callers of a process - and other kernels in flight end on
// their own.  Larger tensors (rows > 256 * rl * 8) take the two-pass path, and so does synchronised BatchNorm (its exchange lives in
// the finalisation above).  Several PROCESSES on one GPU (the one-GPU data-parallel tests) have no residency guarantee:
// i3d_set_bn_bwd_one_launch(0).  Every poll is bounded (a lost workgroup gives wrong numbers, not a hung device).
struct FusedBwdGeom {
    int rpb;      // rows per workgroup
    int tpr;      // threads per row = feat / 4
    int rl;       // row lanes = 512 / tpr
    unsigned tag; // this launch's tag
};
constexpr int FUSED_THREADS = 512, FUSED_MAX_RPT = 8, FUSED_LDS = 2048;      // rl * feat <= 2048 floats per sum
constexpr int FUSED_MAX_BLOCKS = 256;

__device__ __forceinline__ void st_tagged(unsigned long long* p, float v, unsigned tag) {
    __hip_atomic_store(p, ((unsigned long long)tag << 32) | (unsigned long long)__float_as_uint(v), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned long long ld_tagged(const unsigned long long* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <int RPT, bool GA>
__global__ void __launch_bounds__(FUSED_THREADS) __attribute__((amdgpu_waves_per_eu(4, 4)))
bn_bwd_fused_kernel(BwdApplyArgs g, int rows, FusedBwdGeom ge, unsigned long long* tpart /* [nblk][2][feat] */,
                    unsigned long long* tfinal /* [2][feat] */, float* out_beta, float* out_gamma) {
    I3D_CHAIN_PRIO();
    __shared__ float sm[2 * FUSED_LDS];     // [2][rl][feat]; later: the 2 feat final sums
    __shared__ double s_red[2][64][FIN_COLS];
    const int t = threadIdx.x;
    const int cl = t % ge.tpr, rlane = t / ge.tpr;
    const bool active = rlane < ge.rl;
    const int F = g.feat, c0 = cl * 4;
    const unsigned tag = ge.tag;
    const int nblk = gridDim.x;
    if (g.zero_out != nullptr && blockIdx.x == 0 && rlane == 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) g.zero_out[c0 + i] = 0.f;
    }
    float mu[4], is[4], ga[4], be[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { mu[i] = 0.f; is[i] = 1.f; ga[i] = 1.f; be[i] = 0.f; }
    const int r_begin = blockIdx.x * ge.rpb;
    const int r_end = min(rows, r_begin + ge.rpb);
    float dy[RPT][4], x[RPT][4];
    float a1[4] = {0.f, 0.f, 0.f, 0.f}, a2[4] = {0.f, 0.f, 0.f, 0.f};
    if (active) {
#pragma unroll
        for (int u = 0; u < RPT; ++u) {      // all loads first (clamped row: unconditional)
            const int r = min(r_begin + rlane + u * ge.rl, r_end - 1);
            const long off = (long)r * F + c0;
            const float4 a = *reinterpret_cast<const float4*>(g.grad_y + off);
            const float4 b = load4_maybe_bf16(g.x, off, g.x_bf16);
            dy[u][0] = a.x; dy[u][1] = a.y; dy[u][2] = a.z; dy[u][3] = a.w;
            x[u][0] = b.x; x[u][1] = b.y; x[u][2] = b.z; x[u][3] = b.w;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) { mu[i] = g.mean[c0 + i]; is[i] = g.invstd[c0 + i]; ga[i] = g.gamma[c0 + i]; be[i] = g.beta[c0 + i]; }
#pragma unroll
        for (int u = 0; u < RPT; ++u) {      // rows r, r + rl, r + 2 rl, ... in this order
            if (r_begin + rlane + u * ge.rl >= r_end) continue;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float xh = (x[u][i] - mu[i]) * is[i];
                float d = dy[u][i];
                if (g.post_act != I3D_ACT_NONE) d *= act_grad_c<GA>(xh * ga[i] + be[i], g.post_act);
                dy[u][i] = d;
                a1[i] += d;
                a2[i] += d * xh;
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            sm[rlane * F + c0 + i] = a1[i];
            sm[FUSED_LDS + rlane * F + c0 + i] = a2[i];
        }
    }
    __syncthreads();
    // 1. the workgroup's partial row: one thread per (sum, column) adds the row lanes in order and publishes a tagged word
    for (int j = t; j < 2 * F; j += FUSED_THREADS) {
        const int which = j >= F ? 1 : 0, c = j - which * F;
        float s = 0.f;
        for (int k = 0; k < ge.rl; ++k) s += sm[which * FUSED_LDS + k * F + c];
        st_tagged(tpart + (long)blockIdx.x * 2 * F + j, s, tag);
    }
    // 2. the reducers: workgroup b, b + nblk, ... reduce the column blocks [8 b, 8 b + 8) (fp64, partial rows in the fixed order
    //    lane ly takes rows ly, ly + 64, ...; then the 64 lanes in order)
    const int ncb = (F + FIN_COLS - 1) / FIN_COLS;
    for (int cb = blockIdx.x; cb < ncb; cb += nblk) {
        const int cx = t & (FIN_COLS - 1), ly = t / FIN_COLS;       // 8 columns x 64 lanes
        const int c = cb * FIN_COLS + cx;
        constexpr int NB = FUSED_MAX_BLOCKS / 64;
        double p1 = 0.0, p2 = 0.0;
        if (c < F) {
            unsigned long long v1[NB], v2[NB];
            for (int spins = 0; spins < (1 << 22); ++spins) {
                bool ok = true;
#pragma unroll
                for (int k = 0; k < NB; ++k) {
                    const int b = min(ly + k * 64, nblk - 1);
                    v1[k] = ld_tagged(tpart + (long)b * 2 * F + c);
                    v2[k] = ld_tagged(tpart + (long)b * 2 * F + F + c);
                }
#pragma unroll
                for (int k = 0; k < NB; ++k) ok = ok && (unsigned)(v1[k] >> 32) == tag && (unsigned)(v2[k] >> 32) == tag;
                if (ok) break;
                __builtin_amdgcn_s_sleep(1);
            }
#pragma unroll
            for (int k = 0; k < NB; ++k) {
                if (ly + k * 64 < nblk) {
                    p1 += (double)__uint_as_float((unsigned)v1[k]);
                    p2 += (double)__uint_as_float((unsigned)v2[k]);
                }
            }
        }
        __syncthreads();            // (s_red of a previous column block has been read)
        s_red[0][ly][cx] = p1;
        s_red[1][ly][cx] = p2;
        __syncthreads();
        if (t < 2 * FIN_COLS) {
            const int which = t / FIN_COLS, cc = cb * FIN_COLS + (t & (FIN_COLS - 1));
            if (cc < F) {
                double sum = 0.0;
#pragma unroll
                for (int k = 0; k < 64; ++k) sum += s_red[which][k][t & (FIN_COLS - 1)];
                const float fs = (float)sum;
                st_tagged(tfinal + which * F + cc, fs, tag);
                (which ? out_gamma : out_beta)[cc] = fs;      // grad_gamma = sum dy xhat, grad_beta = sum dy
            }
        }
    }
    // 3. everybody: the 2 feat final sums, polled by the first wave, through the LDS (sm is free: all of it was read before the
    //    publishing loop's stores, and the barriers above order that for the reducers; a barrier here for the others)
    __syncthreads();
    if (t < 64) {
        for (int j = t; j < 2 * F; j += 64) {
            unsigned long long v = 0;
            for (int spins = 0; spins < (1 << 22); ++spins) {
                v = ld_tagged(tfinal + j);
                if ((unsigned)(v >> 32) == tag) break;
                __builtin_amdgcn_s_sleep(1);
            }
            sm[j] = __uint_as_float((unsigned)v);
        }
    }
    __syncthreads();
    if (!active) return;
    const float inv_n = g.inv_n;
    float k1[4], k2[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { k1[i] = sm[c0 + i] * inv_n; k2[i] = sm[F + c0 + i] * inv_n; }
#pragma unroll
    for (int u = 0; u < RPT; ++u) {
        const int r = r_begin + rlane + u * ge.rl;
        if (r >= r_end) continue;
        float o[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float xh = (x[u][i] - mu[i]) * is[i];
            float gx = ga[i] * is[i] * (dy[u][i] - k1[i] - xh * k2[i]);
            if (g.act != I3D_ACT_NONE) gx *= act_grad_c<GA>(x[u][i], g.act);     // relu'(pre) == relu'(x)
            o[i] = gx;
        }
        *reinterpret_cast<float4*>(g.grad_pre + (long)r * g.ld_out + c0) = make_float4(o[0], o[1], o[2], o[3]);
    }
}

// BatchNorm backward of the edge block fused with what follows it on the chain (round 4): the data gradient
//     g_e = gamma invstd (dy_e - mean(dy) - xhat_e mean(dy xhat)) act'(x_e)
// of every edge row (bn_bwd_apply_kernel's expression) is formed ON THE FLY by the two segmented sums that consume it -
//     dP[src][v] = sum over the out-edges of v,  dP[dst][v] = sum over the in-edges of v          (edge.hip: segment_sum_kernel<4, true>)
// - one lane per (node, 4 features, direction), rows in the order the segmented-sum kernel adds them: the same bits as the two
// launches it replaces (apply + column sums of the result: 21-24 us, pair of segmented sums: 8.5 us at batch 512).  The lanes of
// the in-edge direction also store g (the weight gradients of the block read it); the bias gradient - the column sum of g - is the
// column sum of dP[dst] and is taken on the weight-gradient stream (i3d_colsum_strided), off the chain.
struct EdgeSums {
    const int* in_ptr;      // [N + 1] rows of the in-edges of node v: [in_ptr[v], in_ptr[v + 1]) (destination-sorted edge order)
    const int* out_ptr;     // [N + 1] / out_epos: row of every out-edge, grouped by source node
    const int* out_epos;
    int num_nodes;
    float* out_src;         // dP[src] [N, feat], row pitch ldo
    float* out_dst;         // dP[dst]
    int ldo;
};

__global__ void __launch_bounds__(256) bn_bwd_apply_edge_sums_kernel(BwdApplyArgs g, EdgeSums e) {
    I3D_CHAIN_PRIO();
    constexpr int MAX_F = 512;
    __shared__ __attribute__((aligned(16))) float col[5 * MAX_F];      // mean | invstd | gamma invstd | k1 | k2
    const int F = g.feat, FV = F / 4;
    {
        const float inv_n = g.inv_n_ptr ? g.inv_n_ptr[0] : g.inv_n;
        for (int c = threadIdx.x; c < F; c += blockDim.x) {
            const float is = g.invstd[c];
            col[c] = g.mean[c];
            col[F + c] = is;
            col[2 * F + c] = g.gamma[c] * is;
            col[3 * F + c] = g.sum_dy[c] * inv_n;
            col[4 * F + c] = g.sum_dy_xhat[c] * inv_n;
        }
    }
    __syncthreads();
    long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long half = (long)e.num_nodes * FV;
    if (t >= 2 * half) return;
    const bool in_dir = t >= half;          // second half of the grid: in-edges (contiguous rows), stores g
    if (in_dir) t -= half;
    const int v = (int)(t / FV), c = (int)(t - (long)v * FV) * 4;
    const int* ptr = in_dir ? e.in_ptr : e.out_ptr;
    const int beg = ptr[v], end = ptr[v + 1];
    const float4 mu = *reinterpret_cast<const float4*>(col + c), is = *reinterpret_cast<const float4*>(col + F + c),
                 gi = *reinterpret_cast<const float4*>(col + 2 * F + c), k1 = *reinterpret_cast<const float4*>(col + 3 * F + c),
                 k2 = *reinterpret_cast<const float4*>(col + 4 * F + c);
    const int act = g.act;
    auto grad_of = [&](const float4 dy, const float4 x) -> float4 {
        float4 r;
        {   const float xh = (x.x - mu.x) * is.x; r.x = gi.x * (dy.x - k1.x - xh * k2.x); }
        {   const float xh = (x.y - mu.y) * is.y; r.y = gi.y * (dy.y - k1.y - xh * k2.y); }
        {   const float xh = (x.z - mu.z) * is.z; r.z = gi.z * (dy.z - k1.z - xh * k2.z); }
        {   const float xh = (x.w - mu.w) * is.w; r.w = gi.w * (dy.w - k1.w - xh * k2.w); }
        if (act != I3D_ACT_NONE) {
            r.x *= act_grad_c<false>(x.x, act); r.y *= act_grad_c<false>(x.y, act);
            r.z *= act_grad_c<false>(x.z, act); r.w *= act_grad_c<false>(x.w, act);
        }
        return r;
    };
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    // four rows per trip, loaded unconditionally (clamped index), the additions in the order j = beg .. end - 1: segment_sum_kernel
    for (int j = beg; j < end; j += 4) {
        long rows[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int jj = min(j + k, end - 1);
            rows[k] = in_dir ? jj : e.out_epos[jj];
        }
        float4 dy[4], xx[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            dy[k] = *reinterpret_cast<const float4*>(g.grad_y + rows[k] * F + c);
            xx[k] = *reinterpret_cast<const float4*>(g.x + rows[k] * F + c);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (j + k < end) {
                const float4 r = grad_of(dy[k], xx[k]);
                if (in_dir) *reinterpret_cast<float4*>(g.grad_pre + rows[k] * g.ld_out + c) = r;
                acc.x += r.x; acc.y += r.y; acc.z += r.z; acc.w += r.w;
            }
        }
    }
    float* o = (in_dir ? e.out_dst : e.out_src) + (long)v * e.ldo + c;
    *reinterpret_cast<float4*>(o) = acc;
}

__global__ void __launch_bounds__(256) act_fwd_kernel(const float* __restrict__ x, long n, int act, float* __restrict__ y) {
    I3D_CHAIN_PRIO();
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (long)gridDim.x * blockDim.x)
        y[t] = apply_act_any(x[t], act);
}

__global__ void __launch_bounds__(256)
act_bwd_kernel(const float* __restrict__ gy, const float* __restrict__ x, long n, int act, float* __restrict__ gx) {
    I3D_CHAIN_PRIO();
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (long)gridDim.x * blockDim.x)
        gx[t] = gy[t] * act_grad_any(x[t], act);
}

__global__ void __launch_bounds__(256) add_inplace_kernel(float* __restrict__ dst, const float* __restrict__ src, long n) {
    I3D_CHAIN_PRIO();
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (long)gridDim.x * blockDim.x) dst[t] += src[t];
}

__global__ void __launch_bounds__(256) add_kernel(const float* __restrict__ a, const float* __restrict__ b, long n,
                                                  float* __restrict__ out) {
    I3D_CHAIN_PRIO();
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (long)gridDim.x * blockDim.x) out[t] = a[t] + b[t];
}

// out = a * b (dropout: b is the mask already scaled by 1 / (1 - p), reference models/base_layers.py:104-105; out may alias a)
__global__ void __launch_bounds__(256) mul_kernel(const float* __restrict__ a, const float* __restrict__ b, long n, float* out) {
    I3D_CHAIN_PRIO();
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (long)gridDim.x * blockDim.x) out[t] = a[t] * b[t];
}

__global__ void __launch_bounds__(256) broadcast_row_kernel(const float* __restrict__ row, long rows, int feat,
                                                            float* __restrict__ out) {
    I3D_CHAIN_PRIO();
    const long n = rows * feat;
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (long)gridDim.x * blockDim.x) out[t] = row[t % feat];
}

static int grid_for(long items) {
    long b = (items + 255) / 256;
    if (b > 256 * 16) b = 256 * 16;
    if (b < 1) b = 1;
    return (int)b;
}

// On by default (I3D_FUSED_FINAL=0: separate stage-2 launches).  Measured on MI355X the in-launch stage 2 costs on the
// GPU what the extra kernel costs (ticket + poll + uncached partial reads ~ 5 us vs a 4.5 us kernel + boundary) and saves
// the host 53 launches per step (~0.18 ms): -1.5 % step time at batch 512 / depth 4, at depth 7 and at batch 2048
// (min and median of 4-6 interleaved runs per variant on one box, tools/ab.sh); bit-identical results.
static bool fused_final() {
    static const bool on = [] { const char* e = getenv("I3D_FUSED_FINAL"); return e == nullptr || e[0] != '0'; }();
    return on;
}

// the bias gradient of a Linear directly in front of a BatchNorm (no activation) is written as the exact value, zero (see bn_bwd_impl)
static bool exact_zero_bias_grad() { return true; }

static float* partial_of(void* workspace) { return (float*)((char*)workspace + WS_HEADER); }

// [3 feat] doubles per rank in a mailbox slot, one flag per column block
static bool peer_fits(int feat) { return 6L * feat <= PEER_PAYLOAD_WORDS; }

static Final stats_final_desc(void* workspace, const float* pre, int act, int rows, int feat, float eps, float momentum,
                              float* mean, float* invstd, float* running_mean, float* running_var, double* sums_out,
                              long long* batches_tracked = nullptr) {
    Final f = {};
    f.batches_tracked = batches_tracked;
    f.kind = 0; f.feat = feat; f.rows = rows; f.act = act; f.eps = eps; f.momentum = momentum; f.pre_row0 = pre;
    f.mean = mean; f.invstd = invstd; f.running_mean = running_mean; f.running_var = running_var; f.sums_out = sums_out;
    f.counters = fused_final() ? (unsigned*)workspace : nullptr;
    return f;
}

static Final pair_final_desc(void* workspace, int feat, float* out1, float* out2, double* sums_out) {
    Final f = {};
    f.kind = 1; f.feat = feat; f.out1 = out1; f.out2 = out2; f.sums_out = sums_out;
    f.counters = fused_final() ? (unsigned*)workspace : nullptr;
    return f;
}

// the one-launch BatchNorm backward (bn_bwd_fused_kernel): process-wide switch, on by default (I3D_BN_BWD_ONE_LAUNCH=0 / the setter)
static int g_one_launch = [] { const char* e = getenv("I3D_BN_BWD_ONE_LAUNCH"); return (e == nullptr || e[0] != '0') ? 1 : 0; }();
static std::atomic<unsigned> g_fused_tag{0};

// the tagged words of the one-launch backward live behind the two-pass kernels' part of the workspace (i3d_colreduce_workspace_bytes)
static long two_pass_workspace_bytes(int feat) {
    return WS_HEADER + (long)MAX_PARTIAL_BLOCKS * 2 * feat * sizeof(float) + 4 * (long)feat * sizeof(double) + 64;
}

// launches bn_bwd_fused_kernel when the call qualifies (-> true)
static bool bn_bwd_one_launch(const BwdApplyArgs& b, int rows, int feat, void* workspace, float* grad_beta, float* grad_gamma, hipStream_t s) {
    if (!g_one_launch || !fused_final() || feat % 4 != 0 || feat > 2048 || b.pre != nullptr || b.grad_pre == nullptr) return false;
    if (b.ld_out % 4 != 0 || ((((uintptr_t)b.grad_y) | ((uintptr_t)b.grad_pre)) & 15) != 0 || (((uintptr_t)b.x) & (b.x_bf16 ? 7 : 15)) != 0)
        return false;
    FusedBwdGeom ge;
    ge.tpr = feat / 4;
    ge.rl = FUSED_THREADS / ge.tpr;
    if ((long)rows > (long)FUSED_MAX_BLOCKS * ge.rl * FUSED_MAX_RPT) return false;       // (the residency argument above)
    int G = cdiv(rows, ge.rl);
    if (G > FUSED_MAX_BLOCKS) G = FUSED_MAX_BLOCKS;
    ge.rpb = cdiv(rows, G);
    G = cdiv(rows, ge.rpb);
    unsigned tag = ++g_fused_tag;
    if (tag == 0) tag = ++g_fused_tag;
    ge.tag = tag;
    unsigned long long* tpart = (unsigned long long*)((char*)workspace + ((two_pass_workspace_bytes(feat) + 63) & ~63L));
    unsigned long long* tfinal = tpart + (long)FUSED_MAX_BLOCKS * 2 * feat;
    const int rpt = cdiv(ge.rpb, ge.rl);
    const bool ga = !(relu_class(b.act) && relu_class(b.post_act));
#define I3D_LAUNCH_FUSED(R)                                                                                                        \
    do {                                                                                                                             \
        if (ga) hipLaunchKernelGGL((bn_bwd_fused_kernel<R, true>), dim3(G), dim3(FUSED_THREADS), 0, s, b, rows, ge, tpart, tfinal,  \
                                   grad_beta, grad_gamma);                                                                           \
        else hipLaunchKernelGGL((bn_bwd_fused_kernel<R, false>), dim3(G), dim3(FUSED_THREADS), 0, s, b, rows, ge, tpart, tfinal,    \
                                grad_beta, grad_gamma);                                                                              \
    } while (0)
    if (rpt <= 2) I3D_LAUNCH_FUSED(2);
    else if (rpt <= 4) I3D_LAUNCH_FUSED(4);
    else if (rpt <= 6) I3D_LAUNCH_FUSED(6);
    else if (rpt == 7) I3D_LAUNCH_FUSED(7);
    else I3D_LAUNCH_FUSED(8);
#undef I3D_LAUNCH_FUSED
    return true;
}

// stage 1 (+ stage 2 in the same launch, or as a second launch when I3D_FUSED_FINAL=0)
template <int MODE>
static void launch_reduction(const ReduceArgs& g, const Chunking& ch, const Final& f, hipStream_t s) {
    dim3 grid(ch.nblk, ch.ncolblk);
    const bool ga = !(relu_class(g.act) && relu_class(g.post_act));
    if (ch.V == 4) {
        if (ga) hipLaunchKernelGGL((colreduce_partial_kernel<MODE, 4, true>), grid, dim3(256), 0, s, g, ch, f);
        else hipLaunchKernelGGL((colreduce_partial_kernel<MODE, 4, false>), grid, dim3(256), 0, s, g, ch, f);
    } else {
        if (ga) hipLaunchKernelGGL((colreduce_partial_kernel<MODE, 1, true>), grid, dim3(256), 0, s, g, ch, f);
        else hipLaunchKernelGGL((colreduce_partial_kernel<MODE, 1, false>), grid, dim3(256), 0, s, g, ch, f);
    }
    if (f.counters != nullptr) return;
    if (f.kind == 0)
        hipLaunchKernelGGL(stats_final_kernel, dim3(cdiv(f.feat, FIN_COLS)), dim3(256), 0, s, g.partial, ch.nblk, f.pre_row0,
                           f.act, f.rows, f.feat, f.eps, f.momentum, f.mean, f.invstd, f.running_mean, f.running_var, f.sums_out,
                           f.batches_tracked);
    else
        hipLaunchKernelGGL(pair_final_kernel, dim3(cdiv(f.feat, FIN_COLS)), dim3(256), 0, s, g.partial, ch.nblk, f.feat, f.out1,
                           f.out2, f.sums_out);
}

__global__ void set_double_kernel(double* p, double v) { *p = v; }

}  // namespace i3d

using namespace i3d;

extern "C" long i3d_colreduce_workspace_bytes(int rows, int feat) {
    (void)rows;      // the two-pass kernels' part, then the tagged words of the one-launch backward: [256][2][feat] + [2][feat] x 8 bytes
    return ((two_pass_workspace_bytes(feat) + 63) & ~63L) + ((long)FUSED_MAX_BLOCKS + 1) * 2 * feat * 8;
}

extern "C" int i3d_act_stats_fwd(const float* pre, int rows, int feat, int act, float* x, float eps, float momentum,
                                 float* mean, float* invstd, float* running_mean, float* running_var,
                                 double* sums_out, void* workspace, void* stream) {
    return i3d_act_stats_fwd_counted(pre, rows, feat, act, x, eps, momentum, mean, invstd, running_mean, running_var,
                                     sums_out, nullptr, workspace, stream);
}

extern "C" int i3d_act_stats_fwd_counted(const float* pre, int rows, int feat, int act, float* x, float eps, float momentum,
                                         float* mean, float* invstd, float* running_mean, float* running_var,
                                         double* sums_out, long long* num_batches_tracked, void* workspace, void* stream) {
    I3D_CHECK_ARG(rows > 0 && feat > 0, "rows > 0 and feat > 0 required");
    I3D_CHECK_ARG(workspace != nullptr, "workspace required");
    I3D_CHECK_ARG(fused_act(act), "this activation exists as an elementwise pass only (i3d_act_fwd / i3d_act_bwd)");
    hipStream_t s = (hipStream_t)stream;
    if (const I3dCollectives* coll = sums_out == nullptr ? collectives() : nullptr) {
        // synchronised BatchNorm (comm.hip): un-shifted fp64 [sum, sum of squares, count] of this rank -> all-reduce on this
        // stream -> mean / invstd / running statistics over all ranks
        PeerCtx* pc = peer_active(stream);      // (a context and a scratch per stream that issues collectives)
        if (pc != nullptr && fused_final() && peer_fits(feat)) {
            // peer-write exchange inside the reduction's in-launch finalisation: ONE launch, as without synchronisation
            Chunking ch = make_chunking(rows, feat);
            ReduceArgs g = {};
            g.a = pre; g.out = x; g.rows = rows; g.feat = feat; g.act = act; g.post_act = I3D_ACT_NONE;
            g.partial = partial_of(workspace);
            Final f = stats_final_desc(workspace, pre, act, rows, feat, eps, momentum, mean, invstd, running_mean, running_var, nullptr,
                                       num_batches_tracked);
            f.peer_on = 1;
            if (int rc = peer_next(pc, &f.peer)) return rc;
            launch_reduction<MODE_STATS>(g, ch, f, s);
            I3D_CHECK_LAUNCH();
            return I3D_OK;
        }
        I3D_CHECK_ARG((pc ? peer_scratch_bytes(pc) : coll->scratch_bytes) >= (long)(2 * feat + 1) * 8, "collective scratch too small");
        double* s64 = (double*)(pc ? peer_scratch(pc) : coll->scratch);
        int rc = i3d_act_stats_fwd_counted(pre, rows, feat, act, x, eps, momentum, mean, invstd, nullptr, nullptr, s64, nullptr,
                                           workspace, stream);
        if (rc != I3D_OK) return rc;
        if (pc != nullptr) rc = peer_sum_f64(pc, s64, 2 * feat + 1, 0, 0.0, s64, nullptr, nullptr, stream);
        else rc = coll->all_reduce_f64(coll->user, s64, 2 * feat + 1, stream);
        if (rc != I3D_OK) return rc;
        hipLaunchKernelGGL(stats_from_sums_kernel, dim3(cdiv(feat, 128)), dim3(128), 0, s, s64, feat, eps, momentum, mean, invstd,
                           running_mean, running_var, num_batches_tracked);
        I3D_CHECK_LAUNCH();
        return I3D_OK;
    }
    Chunking ch = make_chunking(rows, feat);
    ReduceArgs g = {};
    g.a = pre; g.out = x; g.rows = rows; g.feat = feat; g.act = act; g.post_act = I3D_ACT_NONE;
    g.partial = partial_of(workspace);
    launch_reduction<MODE_STATS>(g, ch, stats_final_desc(workspace, pre, act, rows, feat, eps, momentum, mean, invstd,
                                                         running_mean, running_var, sums_out, num_batches_tracked), s);
    I3D_CHECK_LAUNCH();
    return I3D_OK;
}

extern "C" int i3d_bn_finalize_stats(const double* sums, int feat, float eps, float momentum, float* mean,
                                     float* invstd, float* running_mean, float* running_var, void* stream) {
    I3D_CHECK_ARG(feat > 0, "feat > 0 required");
    hipLaunchKernelGGL(stats_from_sums_kernel, dim3(cdiv(feat, 128)), dim3(128), 0, (hipStream_t)stream, sums, feat, eps,
                       momentum, mean, invstd, running_mean, running_var);
    I3D_CHECK_LAUNCH();
    return I3D_OK;
}

static int bn_apply_common(const float* x, int rows, int feat, const float* mean, const float* invstd_or_var,
                           const float* gamma, const float* beta, int post_act, const float* residual, float* y,
                           int eval_mode, float eps, void* stream) {
    if (rows == 0) return I3D_OK;
    ApplyArgs g;
    g.x = x; g.mean = mean; g.invstd = invstd_or_var; g.gamma = gamma; g.beta = beta; g.residual = residual; g.y = y;
    g.feat = feat; g.post_act = post_act; g.eval_mode = eval_mode; g.eps = eps;
    hipStream_t s = (hipStream_t)stream;
    g.items = 0;
    const int V = feat % 4 == 0 ? 4 : 1;
    const RowTiling rt = make_row_tiling(feat, V);
    dim3 grid(cdiv(rows, rt.rl * RU), rt.ncolblk);
    const bool ga = !relu_class(post_act);
    if (V == 4) {
        if (ga) hipLaunchKernelGGL((bn_apply_kernel<4, true>), grid, dim3(256), 0, s, g, rt, rows);
        else hipLaunchKernelGGL((bn_apply_kernel<4, false>), grid, dim3(256), 0, s, g, rt, rows);
    } else {
        if (ga) hipLaunchKernelGGL((bn_apply_kernel<1, true>), grid, dim3(256), 0, s, g, rt, rows);
        else hipLaunchKernelGGL((bn_apply_kernel<1, false>), grid, dim3(256), 0, s, g, rt, rows);
    }
    return I3D_OK;
}

extern "C" int i3d_bn_apply_fwd(const float* x, int rows, int feat, const float* mean, const float* invstd,
                                const float* gamma, const float* beta, int post_act, const float* residual, float* y,
                                void* stream) {
    I3D_CHECK_ARG(rows >= 0 && feat > 0, "bad shape");
    I3D_CHECK_ARG(fused_act(post_act), "this activation exists as an elementwise pass only (i3d_act_fwd / i3d_act_bwd)");
    bn_apply_common(x, rows, feat, mean, invstd, gamma, beta, post_act, residual, y, 0, 0.f, stream);
    I3D_CHECK_LAUNCH();
    return I3D_OK;
}

extern "C" int i3d_bn_eval_fwd(const float* x, int rows, int feat, const float* running_mean,
                               const float* running_var, float eps, const float* gamma, const float* beta,
                               int post_act, const float* residual, float* y, void* stream) {
    I3D_CHECK_ARG(rows >= 0 && feat > 0, "bad shape");
    I3D_CHECK_ARG(fused_act(post_act), "this activation exists as an elementwise pass only (i3d_act_fwd / i3d_act_bwd)");
    bn_apply_common(x, rows, feat, running_mean, running_var, gamma, beta, post_act, residual, y, 1, eps, stream);
    I3D_CHECK_LAUNCH();
    return I3D_OK;
}

static void launch_bwd_apply(BwdApplyArgs& g, int rows, int feat, hipStream_t s) {
    g.items = 0;
    const int V = feat % 4 == 0 ? 4 : 1;
    const RowTiling rt = make_row_tiling(feat, V);
    dim3 grid(cdiv(rows, rt.rl * RU), rt.ncolblk);
    const bool ga = !(relu_class(g.act) && relu_class(g.post_act));
    if (V == 4) {
        if (ga) hipLaunchKernelGGL((bn_bwd_apply_kernel<4, true>), grid, dim3(256), 0, s, g, rt, rows);
        else hipLaunchKernelGGL((bn_bwd_apply_kernel<4, false>), grid, dim3(256), 0, s, g, rt, rows);
    } else {
        if (ga) hipLaunchKernelGGL((bn_bwd_apply_kernel<1, true>), grid, dim3(256), 0, s, g, rt, rows);
        else hipLaunchKernelGGL((bn_bwd_apply_kernel<1, false>), grid, dim3(256), 0, s, g, rt, rows);
    }
}

extern "C" int i3d_bn_bwd(const float* grad_y, const float* x, const float* pre, int rows, int feat, int act,
                          int post_act, const float* mean, const float* invstd, const float* gamma,
                          const float* beta, float* grad_gamma, float* grad_beta, float* grad_pre, float* grad_bias,
                          double* sums_out, const double* sums_in, long total_rows, void* workspace, void* stream) {
    return i3d_bn_bwd_deferred_bias(grad_y, x, pre, rows, feat, act, post_act, mean, invstd, gamma, beta, grad_gamma, grad_beta,
                                    grad_pre, grad_bias, sums_out, sums_in, total_rows, workspace, nullptr, stream);
}

// would i3d_bn_bwd (activation none / ReLU / LeakyReLU, fp32 x, no bias column sums) take the one-launch kernel for this shape?
extern "C" int i3d_bn_bwd_one_launch_supported(int rows, int feat) {
    if (!g_one_launch || !fused_final() || collectives() != nullptr || feat % 4 != 0 || feat > 2048 || rows <= 0) return 0;
    return (long)rows <= (long)FUSED_MAX_BLOCKS * (FUSED_THREADS / (feat / 4)) * FUSED_MAX_RPT ? 1 : 0;
}

extern "C" int i3d_set_bn_bwd_one_launch(int on) {
    const int was = g_one_launch;
    g_one_launch = on ? 1 : 0;
    return was;
}

extern "C" long i3d_bn_bias_partial_floats(int feat) { return (long)MAX_PARTIAL_BLOCKS * 2 * feat; }

// grad_bias = column sums of grad_pre from the row-chunk partials the data-gradient pass left in bias_partial
extern "C" int i3d_bn_bias_finalize(const float* bias_partial, int rows, int feat, float* grad_bias, void* stream) {
    I3D_CHECK_ARG(bias_partial != nullptr && rows > 0 && feat > 0 && grad_bias != nullptr, "bad arguments");
    const Chunking ch = make_chunking(rows, feat);
    hipLaunchKernelGGL(pair_final_kernel, dim3(cdiv(feat, FIN_COLS)), dim3(256), 0, (hipStream_t)stream, bias_partial, ch.nblk,
                       feat, grad_bias, (float*)nullptr, (double*)nullptr);
    I3D_CHECK_LAUNCH();
    return I3D_OK;
}

// bias_partial != null (and grad_bias != null): the data-gradient pass stores the row-chunk partials of the bias gradient
// there and does NOT finalise them - the in-launch finalisation is a ~10 us serial tail on the backward chain for a value
// only the optimizer needs; the caller runs i3d_bn_bias_finalize later (the layer composite: on its side stream).
// the BatchNorm input x of the next bn_bwd_impl calls of this thread is stored as bf16 (i3d_bn_bwd_x_bf16)
static thread_local int g_x_bf16 = 0;
// the data-gradient pass of the next bn_bwd_impl call of this thread is bn_bwd_apply_edge_sums_kernel (i3d_bn_bwd_edge_sums)
static thread_local const EdgeSums* g_edge_sums = nullptr;
// phase 2 of the synchronised backward finds its fp32 sum vectors + 1 / rows already in the workspace (peer exchange)
static thread_local int g_sums_ready = 0;

static int bn_bwd_impl(const float* grad_y, const float* x, const float* pre, int rows, int feat, int act,
                       int post_act, const float* mean, const float* invstd, const float* gamma,
                       const float* beta, float* grad_gamma, float* grad_beta, float* grad_pre,
                       float* grad_bias, double* sums_out, const double* sums_in, long total_rows,
                       void* workspace, float* bias_partial, int ld_out, void* stream) {
    I3D_CHECK_ARG(rows > 0 && feat > 0, "rows > 0 and feat > 0 required");
    I3D_CHECK_ARG(workspace != nullptr, "workspace required");
    I3D_CHECK_ARG(act == I3D_ACT_NONE || act == I3D_ACT_RELU || act == I3D_ACT_LEAKY_RELU || pre != nullptr, "pre required for this activation");
    I3D_CHECK_ARG(fused_act(act) && fused_act(post_act), "this activation exists as an elementwise pass only (i3d_act_fwd / i3d_act_bwd)");
    hipStream_t s = (hipStream_t)stream;
    if (const I3dCollectives* coll = (sums_out == nullptr && sums_in == nullptr) ? collectives() : nullptr) {
        // synchronised BatchNorm (comm.hip): this rank's fp64 [sum dy, sum dy xhat] and row count -> all-reduce on this
        // stream -> the data gradient from the sums over all ranks; grad_gamma / grad_beta keep this rank's share (the
        // gradient all-reduce adds the ranks up)
        PeerCtx* pc = peer_active(stream);
        if (pc != nullptr && fused_final() && peer_fits(feat)) {
            // peer-write exchange inside the reduction's in-launch finalisation: grad_gamma / grad_beta get this rank's share,
            // the workspace the global sums as fp32 + 1 / global rows - then the data-gradient pass; no extra launch
            Chunking ch = make_chunking(rows, feat);
            float* partial = partial_of(workspace);
            float* tmp = partial + (long)MAX_PARTIAL_BLOCKS * 2 * feat;
            ReduceArgs g = {};
            g.a = grad_y; g.b = x; g.mean = mean; g.invstd = invstd; g.gamma = gamma; g.beta = beta;
            g.rows = rows; g.feat = feat; g.act = act; g.post_act = post_act; g.partial = partial;
            g.b_bf16 = g_x_bf16;
            Final f = pair_final_desc(workspace, feat, grad_beta, grad_gamma, nullptr);
            f.rows = rows; f.peer_on = 1; f.g1 = tmp; f.g2 = tmp + feat; f.ginv = tmp + 2 * feat;
            if (int rc = peer_next(pc, &f.peer)) return rc;
            launch_reduction<MODE_BN_BWD>(g, ch, f, s);
            I3D_CHECK_LAUNCH();
            g_sums_ready = 1;
            const int rc = bn_bwd_impl(grad_y, x, pre, rows, feat, act, post_act, mean, invstd, gamma, beta, grad_gamma, grad_beta,
                                       grad_pre, grad_bias, nullptr, (const double*)tmp /* non-null: phase 2 */, 0, workspace,
                                       bias_partial, ld_out, stream);
            g_sums_ready = 0;
            return rc;
        }
        I3D_CHECK_ARG((pc ? peer_scratch_bytes(pc) : coll->scratch_bytes) >= (long)(2 * feat + 1) * 8, "collective scratch too small");
        double* s64 = (double*)(pc ? peer_scratch(pc) : coll->scratch);
        int rc = bn_bwd_impl(grad_y, x, pre, rows, feat, act, post_act, mean, invstd, gamma, beta, grad_gamma, grad_beta,
                             nullptr, nullptr, s64, nullptr, rows, workspace, nullptr, ld_out, stream);
        if (rc != I3D_OK) return rc;
        if (pc != nullptr) {
            // peer-write exchange (peer.h): row count appended, sums over the ranks, conversion to the fp32 vectors + 1 / rows
            // the data-gradient pass reads - ONE launch instead of set_double -> all-reduce -> sums_to_float
            float* tmp = partial_of(workspace) + (long)MAX_PARTIAL_BLOCKS * 2 * feat;
            rc = peer_sum_f64(pc, s64, 2 * feat, 1, (double)rows, nullptr, tmp, tmp + 2 * feat, stream);
            if (rc != I3D_OK) return rc;
            g_sums_ready = 1;
            rc = bn_bwd_impl(grad_y, x, pre, rows, feat, act, post_act, mean, invstd, gamma, beta, grad_gamma, grad_beta,
                             grad_pre, grad_bias, nullptr, s64, 0, workspace, bias_partial, ld_out, stream);
            g_sums_ready = 0;
            return rc;
        }
        hipLaunchKernelGGL(set_double_kernel, dim3(1), dim3(1), 0, s, s64 + 2 * feat, (double)rows);
        I3D_CHECK_LAUNCH();
        rc = coll->all_reduce_f64(coll->user, s64, 2 * feat + 1, stream);
        if (rc != I3D_OK) return rc;
        return bn_bwd_impl(grad_y, x, pre, rows, feat, act, post_act, mean, invstd, gamma, beta, grad_gamma, grad_beta,
                           grad_pre, grad_bias, nullptr, s64, 0, workspace, bias_partial, ld_out, stream);
    }
    Chunking ch = make_chunking(rows, feat);
    float* partial = partial_of(workspace);
    if (sums_in == nullptr && sums_out == nullptr && g_edge_sums == nullptr && grad_pre != nullptr &&
        (grad_bias == nullptr || (act == I3D_ACT_NONE && exact_zero_bias_grad()))) {
        // one launch: reduction, finalisation and data gradient (bn_bwd_fused_kernel) - tensors of up to 256 * rl * 4 rows
        BwdApplyArgs b = {};
        b.grad_y = grad_y; b.x = x; b.pre = relu_class(act) ? nullptr : pre; b.mean = mean; b.invstd = invstd; b.gamma = gamma;
        b.beta = beta; b.sum_dy = grad_beta; b.sum_dy_xhat = grad_gamma; b.inv_n = 1.f / (float)rows; b.grad_pre = grad_pre;
        b.ld_out = ld_out; b.feat = feat; b.act = act; b.post_act = post_act; b.zero_out = grad_bias; b.x_bf16 = g_x_bf16;
        if (bn_bwd_one_launch(b, rows, feat, workspace, grad_beta, grad_gamma, s)) {
            I3D_CHECK_LAUNCH();
            return I3D_OK;
        }
    }
    if (sums_in == nullptr) {   // phase 1: column sums of dy and dy*xhat
        ReduceArgs g = {};
        g.a = grad_y; g.b = x; g.mean = mean; g.invstd = invstd; g.gamma = gamma; g.beta = beta;
        g.rows = rows; g.feat = feat; g.act = act; g.post_act = post_act; g.partial = partial;
        g.b_bf16 = g_x_bf16;
        launch_reduction<MODE_BN_BWD>(g, ch, pair_final_desc(workspace, feat, grad_beta, grad_gamma, sums_out), s);
        I3D_CHECK_LAUNCH();
        if (sums_out != nullptr) return I3D_OK;   // caller all-reduces, then calls again with sums_in
        total_rows = rows;
    }
    const float* sum_dy = grad_beta;
    const float* sum_dy_xhat = grad_gamma;
    if (sums_in != nullptr) {   // phase 2 of synchronised BN: global sums drive the data gradient
        float* tmp = partial + (long)MAX_PARTIAL_BLOCKS * 2 * feat;
        if (!g_sums_ready) {       // (the peer exchange has written them already)
            hipLaunchKernelGGL(sums_to_float_kernel, dim3(cdiv(feat, 128)), dim3(128), 0, s, sums_in, feat, tmp, tmp + feat);
            I3D_CHECK_LAUNCH();
        }
        sum_dy = tmp;
        sum_dy_xhat = tmp + feat;
    }
    BwdApplyArgs b;
    b.inv_n_ptr = sums_in != nullptr ? (partial + (long)MAX_PARTIAL_BLOCKS * 2 * feat + 2 * feat) : nullptr;
    b.grad_y = grad_y; b.x = x; b.pre = (act == I3D_ACT_NONE || act == I3D_ACT_RELU || act == I3D_ACT_LEAKY_RELU) ? nullptr : pre;
    b.mean = mean; b.invstd = invstd; b.gamma = gamma; b.beta = beta;
    b.sum_dy = sum_dy; b.sum_dy_xhat = sum_dy_xhat; b.grad_pre = grad_pre; b.ld_out = ld_out; b.feat = feat; b.act = act;
    b.post_act = post_act; b.eval_mode = 0; b.inv_n = 1.f / (float)total_rows; b.eps = 0.f;
    b.zero_out = nullptr; b.x_bf16 = g_x_bf16;
    if (g_edge_sums != nullptr) {      // i3d_bn_bwd_edge_sums: the data gradient formed inside the segmented sums that consume it
        const EdgeSums es = *g_edge_sums;
        const long lanes = 2L * es.num_nodes * (feat / 4);
        hipLaunchKernelGGL(bn_bwd_apply_edge_sums_kernel, dim3(cdiv(lanes, 256)), dim3(256), 0, s, b, es);
        I3D_CHECK_LAUNCH();
        return I3D_OK;
    }
    if (grad_bias != nullptr && act == I3D_ACT_NONE && exact_zero_bias_grad()) {
        // No activation between the Linear and the BatchNorm: the bias gradient is the column sum of the BatchNorm input
        // gradient  s (dy - mean(dy) - xhat mean(dy xhat))  over the rows the statistics were taken over, which is
        // IDENTICALLY zero (sum xhat = 0).  The reference sums it up in fp32 and gets rounding noise (~1e-9 of the scale,
        // which its Adam turns into +-lr steps on a bias the BatchNorm removes anyway); here it is the exact value, and the
        // data-gradient pass loses its reduction and the ~10 us finalisation tail.  (Synchronised BatchNorm: the sum over the
        // rows of ALL ranks is zero, a rank's share is not - but the gradient all-reduce adds the shares up, so every rank
        // writing the exact total, zero, gives the same all-reduced value.)
        b.zero_out = grad_bias;
        launch_bwd_apply(b, rows, feat, s);
        I3D_CHECK_LAUNCH();
        return I3D_OK;
    }
    if (grad_bias != nullptr) {     // data gradient and its column sums (bias gradient of the Linear in front) in one pass
        b.items = 0;
        dim3 grid(ch.nblk, ch.ncolblk);
        Final f = pair_final_desc(workspace, feat, grad_bias, nullptr, nullptr);
        float* bias_part = partial;
        if (bias_partial != nullptr) {       // deferred: partials to the caller's buffer, no finalisation here
            f.counters = nullptr;
            bias_part = bias_partial;
        }
        const bool ga = !(relu_class(b.act) && relu_class(b.post_act));
        if (ch.V == 4) {
            if (ga) hipLaunchKernelGGL((bn_bwd_apply_colsum_kernel<4, true>), grid, dim3(256), 0, s, b, ch, rows, bias_part, f);
            else hipLaunchKernelGGL((bn_bwd_apply_colsum_kernel<4, false>), grid, dim3(256), 0, s, b, ch, rows, bias_part, f);
        } else {
            if (ga) hipLaunchKernelGGL((bn_bwd_apply_colsum_kernel<1, true>), grid, dim3(256), 0, s, b, ch, rows, bias_part, f);
            else hipLaunchKernelGGL((bn_bwd_apply_colsum_kernel<1, false>), grid, dim3(256), 0, s, b, ch, rows, bias_part, f);
        }
        I3D_CHECK_LAUNCH();
        if (bias_partial != nullptr) return I3D_OK;
        if (f.counters == nullptr) {
            hipLaunchKernelGGL(pair_final_kernel, dim3(cdiv(feat, FIN_COLS)), dim3(256), 0, s, partial, ch.nblk, feat, grad_bias,
                               (float*)nullptr, (double*)nullptr);
            I3D_CHECK_LAUNCH();
        }
        return I3D_OK;
    }
    launch_bwd_apply(b, rows, feat, s);
    I3D_CHECK_LAUNCH();
    return I3D_OK;
}

extern "C" int i3d_bn_bwd_deferred_bias(const float* grad_y, const float* x, const float* pre, int rows, int feat, int act,
                                        int post_act, const float* mean, const float* invstd, const float* gamma,
                                        const float* beta, float* grad_gamma, float* grad_beta, float* grad_pre,
                                        float* grad_bias, double* sums_out, const double* sums_in, long total_rows,
                                        void* workspace, float* bias_partial, void* stream) {
    return bn_bwd_impl(grad_y, x, pre, rows, feat, act, post_act, mean, invstd, gamma, beta, grad_gamma, grad_beta, grad_pre,
                       grad_bias, sums_out, sums_in, total_rows, workspace, bias_partial, feat, stream);
}

extern "C" int i3d_bn_bwd_edge_sums(const float* grad_y, const float* x, int rows, int feat, int act, const float* mean,
                                    const float* invstd, const float* gamma, const float* beta, float* grad_gamma, float* grad_beta,
                                    float* grad_pre, const int* in_ptr, const int* out_ptr, const int* out_epos, int num_nodes,
                                    float* out_src, float* out_dst, int ldo, void* workspace, void* stream) {
    I3D_CHECK_ARG(feat % 4 == 0 && feat <= 512 && ldo % 4 == 0 && ldo >= feat, "feat % 4 == 0, feat <= 512, 16-byte rows");
    I3D_CHECK_ARG(relu_class(act), "activation: none, ReLU or LeakyReLU (act' from the stored activation)");
    I3D_CHECK_ARG(in_ptr != nullptr && out_ptr != nullptr && out_epos != nullptr && num_nodes > 0 && out_src != nullptr && out_dst != nullptr &&
                      grad_pre != nullptr, "null argument");
    I3D_CHECK_ARG(((((uintptr_t)grad_y) | ((uintptr_t)x) | ((uintptr_t)grad_pre) | ((uintptr_t)out_src) | ((uintptr_t)out_dst)) & 15) == 0,
                  "16-byte aligned tensors");
    const EdgeSums es{in_ptr, out_ptr, out_epos, num_nodes, out_src, out_dst, ldo};
    g_edge_sums = &es;
    const int rc = bn_bwd_impl(grad_y, x, nullptr, rows, feat, act, I3D_ACT_NONE, mean, invstd, gamma, beta, grad_gamma, grad_beta, grad_pre,
                               nullptr, nullptr, nullptr, rows, workspace, nullptr, feat, stream);
    g_edge_sums = nullptr;
    return rc;
}

// out[c] = sum over the rows of x[r * ldx + c]: the two-stage deterministic column reduction of this file on a column block of a
// wider matrix, partials through the caller's buffer (i3d_bn_bias_partial_floats(feat) floats - a stream of its own next to the
// BatchNorm passes must not share their workspace), finalised by a second launch
extern "C" int i3d_colsum_strided(const float* x, int ldx, int rows, int feat, float* out, float* partial, void* stream) {
    I3D_CHECK_ARG(x != nullptr && out != nullptr && partial != nullptr && rows > 0 && feat > 0 && ldx >= feat, "bad arguments");
    I3D_CHECK_ARG(feat % 4 != 0 || (ldx % 4 == 0 && (((uintptr_t)x) & 15) == 0), "16-byte rows for feat % 4 == 0");
    hipStream_t s = (hipStream_t)stream;
    const Chunking ch = make_chunking(rows, feat);
    ReduceArgs g = {};
    g.a = x; g.lda = ldx; g.rows = rows; g.feat = feat; g.partial = partial;
    Final f = {};        // counters == nullptr: no in-launch finalisation, launch_reduction adds the pair finalisation launch
    f.kind = 1; f.feat = feat; f.out1 = out;
    launch_reduction<MODE_COLSUM>(g, ch, f, s);
    I3D_CHECK_LAUNCH();
    return I3D_OK;
}

// i3d_bn_bwd_deferred_bias with the BatchNorm input x stored as bf16 (row r at (bf16*)x + r * feat; feat % 4 == 0; activations
// none / ReLU / LeakyReLU: act' is taken from x): the bf16 mode's storage form of a PNA layer's messages
extern "C" int i3d_bn_bwd_x_bf16(const float* grad_y, const void* x, int rows, int feat, int act, int post_act, const float* mean,
                                 const float* invstd, const float* gamma, const float* beta, float* grad_gamma, float* grad_beta,
                                 float* grad_pre, float* grad_bias, void* workspace, float* bias_partial, void* stream) {
    I3D_CHECK_ARG(feat % 4 == 0 && (((uintptr_t)x) & 7) == 0, "bf16 x: feat % 4 == 0, 8-byte aligned");
    I3D_CHECK_ARG(act == I3D_ACT_NONE || act == I3D_ACT_RELU || act == I3D_ACT_LEAKY_RELU, "bf16 x: act' must not need the Linear output");
    g_x_bf16 = 1;
    const int rc = bn_bwd_impl(grad_y, (const float*)x, nullptr, rows, feat, act, post_act, mean, invstd, gamma, beta, grad_gamma, grad_beta,
                               grad_pre, grad_bias, nullptr, nullptr, rows, workspace, bias_partial, feat, stream);
    g_x_bf16 = 0;
    return rc;
}

// the same with grad_pre as a column block of a wider matrix (row pitch ld_out floats)
extern "C" int i3d_bn_bwd_strided(const float* grad_y, const float* x, const float* pre, int rows, int feat, int act,
                                  int post_act, const float* mean, const float* invstd, const float* gamma, const float* beta,
                                  float* grad_gamma, float* grad_beta, float* grad_pre, int ld_out, float* grad_bias,
                                  void* workspace, float* bias_partial, void* stream) {
    I3D_CHECK_ARG(ld_out >= feat && (feat % 4 != 0 || (ld_out % 4 == 0 && (((uintptr_t)grad_pre) & 15) == 0)), "bad output pitch");
    return bn_bwd_impl(grad_y, x, pre, rows, feat, act, post_act, mean, invstd, gamma, beta, grad_gamma, grad_beta, grad_pre,
                       grad_bias, nullptr, nullptr, rows, workspace, bias_partial, ld_out, stream);
}

extern "C" int i3d_bn_eval_bwd(const float* grad_y, const float* x, const float* pre, int rows, int feat, int act,
                               int post_act, const float* running_mean, const float* running_var, float eps,
                               const float* gamma, const float* beta, float* grad_gamma, float* grad_beta,
                               float* grad_pre, void* workspace, void* stream) {
    I3D_CHECK_ARG(rows > 0 && feat > 0, "rows > 0 and feat > 0 required");
    I3D_CHECK_ARG(workspace != nullptr, "workspace required");
    I3D_CHECK_ARG(fused_act(act) && fused_act(post_act), "this activation exists as an elementwise pass only (i3d_act_fwd / i3d_act_bwd)");
    hipStream_t s = (hipStream_t)stream;
    // grad_gamma / grad_beta need xhat with invstd from running_var: materialise invstd in the workspace tail
    Chunking ch = make_chunking(rows, feat);
    float* partial = partial_of(workspace);
    float* invstd = partial + (long)MAX_PARTIAL_BLOCKS * 2 * feat;
    hipLaunchKernelGGL(invstd_from_var_kernel, dim3(cdiv(feat, 128)), dim3(128), 0, s, running_var, feat, eps, invstd);
    I3D_CHECK_LAUNCH();
    ReduceArgs g = {};
    g.a = grad_y; g.b = x; g.mean = running_mean; g.invstd = invstd; g.gamma = gamma; g.beta = beta;
    g.rows = rows; g.feat = feat; g.act = act; g.post_act = post_act; g.partial = partial;
    launch_reduction<MODE_BN_BWD>(g, ch, pair_final_desc(workspace, feat, grad_beta, grad_gamma, nullptr), s);
    I3D_CHECK_LAUNCH();
    BwdApplyArgs b;
    b.x_bf16 = 0;
    b.grad_y = grad_y; b.x = x; b.pre = (act == I3D_ACT_NONE || act == I3D_ACT_RELU || act == I3D_ACT_LEAKY_RELU) ? nullptr : pre;
    b.mean = running_mean; b.invstd = running_var; b.gamma = gamma; b.beta = beta; b.sum_dy = nullptr;
    b.sum_dy_xhat = nullptr; b.grad_pre = grad_pre; b.ld_out = feat; b.feat = feat; b.act = act; b.post_act = post_act;
    b.inv_n_ptr = nullptr; b.eval_mode = 1; b.inv_n = 0.f; b.eps = eps; b.zero_out = nullptr;
    launch_bwd_apply(b, rows, feat, s);
    I3D_CHECK_LAUNCH();
    return I3D_OK;
}

extern "C" int i3d_colsum(const float* x, const float* w, int rows, int feat, float* out, void* workspace,
                          void* stream) {
    I3D_CHECK_ARG(rows > 0 && feat > 0, "rows > 0 and feat > 0 required");
    I3D_CHECK_ARG(workspace != nullptr, "workspace required");
    hipStream_t s = (hipStream_t)stream;
    Chunking ch = make_chunking(rows, feat);
    ReduceArgs g = {};
    g.a = x; g.b = w; g.rows = rows; g.feat = feat; g.partial = partial_of(workspace);
    launch_reduction<MODE_COLSUM>(g, ch, pair_final_desc(workspace, feat, out, nullptr, nullptr), s);
    I3D_CHECK_LAUNCH();
    return I3D_OK;
}

extern "C" int i3d_act_fwd(const float* x, long n, int act, float* y, void* stream) {
    if (n <= 0) return I3D_OK;
    hipLaunchKernelGGL(act_fwd_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, x, n, act, y);
    I3D_CHECK_LAUNCH();
    return I3D_OK;
}

extern "C" int i3d_act_bwd(const float* grad_y, const float* x, long n, int act, float* grad_x, void* stream) {
    if (n <= 0) return I3D_OK;
    hipLaunchKernelGGL(act_bwd_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, grad_y, x, n, act, grad_x);
    I3D_CHECK_LAUNCH();
    return I3D_OK;
}

extern "C" int i3d_add_inplace(float* dst, const float* src, long n, void* stream) {
    if (n <= 0) return I3D_OK;
    hipLaunchKernelGGL(add_inplace_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, dst, src, n);
    I3D_CHECK_LAUNCH();
    return I3D_OK;
}

extern "C" int i3d_add(const float* a, const float* b, long n, float* out, void* stream) {
    if (n <= 0) return I3D_OK;
    hipLaunchKernelGGL(add_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, a, b, n, out);
    I3D_CHECK_LAUNCH();
    return I3D_OK;
}

extern "C" int i3d_mul(const float* a, const float* b, long n, float* out, void* stream) {
    if (n <= 0) return I3D_OK;
    hipLaunchKernelGGL(mul_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, a, b, n, out);
    I3D_CHECK_LAUNCH();
    return I3D_OK;
}

extern "C" int i3d_broadcast_row(const float* row, long rows, int feat, float* out, void* stream) {
    I3D_CHECK_ARG(rows >= 0 && feat > 0, "bad shape");
    if (rows == 0) return I3D_OK;
    hipLaunchKernelGGL(broadcast_row_kernel, dim3(grid_for(rows * feat)), dim3(256), 0, (hipStream_t)stream, row, rows, feat, out);
    I3D_CHECK_LAUNCH();
    return I3D_OK;
}
