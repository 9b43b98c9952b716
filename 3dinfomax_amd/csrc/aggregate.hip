// K4 - the PNA aggregation kernel (the HBM-roofline kernel of this path).
//
// Replaces, in ONE pass over a destination-sorted message tensor, what the reference gets
// from DGL's degree-bucketed update_all + reduce_func + aggregators + scalers
// (reference models/pna.py:206, 221-235, 17-37, 57-68): per destination node the
// mean / max / min / std (/sum /var) over its in-edge messages, times the degree scalers
// {1, log(D+1)/avg, avg/log(D+1)}, written as [N, n_scalers*n_aggregators*F]
// (column order: scaler-major, aggregator-minor, each block F wide = reference :229-233).
//
// Mapping (CDNA4, wave64): messages are stored in CSR (by destination) order, so node v's
// mailbox is the contiguous row range [in_ptr[v], in_ptr[v+1]).  The reduction axis is the
// neighbour axis (D <= 4 on QM9) and is walked sequentially by one lane; the feature axis is
// the parallel one.  One lane owns one (node, 4-feature) item: a wave reads 1 KiB contiguous
// per neighbour step and writes 1 KiB contiguous per output block with 16-byte accesses, all
// 64 lanes active (a wave-per-node mapping would idle 14 of 64 lanes at F=200).  The 12 output
// blocks of a node are produced from registers - the fan-out costs no re-read of the input.
// Algorithmic bytes per launch: 4*E*F + 4*N*12*F + 4*(N+1).
#include <math.h>

#include <hip/hip_ext.h>

#include "common.h"

namespace i3d {

struct AggCfg {
    int n_agg;
    int agg[8];
    int n_scaler;       // number of scaler blocks actually written (>= 1)
    int scaler[4];      // I3D_SCALE_*; reference quirk: a single configured scaler is not applied
    float amp[32];      // log(D+1)/avg for D < 32 (host-computed in double, like np.log in the reference)
    float att[32];      // avg/log(D+1)
    float avg;
};

__device__ __forceinline__ void scaler_values(const AggCfg& cfg, int D, float& amp, float& att) {
    if (D < 32) {
        amp = cfg.amp[D];
        att = cfg.att[D];
    } else {
        double l = log((double)(D + 1));
        amp = (float)(l / (double)cfg.avg);
        att = (float)((double)cfg.avg / l);
    }
}

struct Stats4 {
    float4 sum, sq, mx, mn;
};

// MODE 0: any aggregator/scaler list   1: (mean,max,min,std) x (identity,amplification,attenuation), 12 blocks
//      2: (mean,max,min,std) x identity, 4 blocks (degree-grouped posttrans: the scalers live in the combined weights)
// aff (optional, [3F] = mean | scale | shift): the messages are read as (e - mean) * scale + shift - the BatchNorm of the
// last pretrans block applied on the fly (fused_bn.hip), so the normalised message tensor never exists in memory
__device__ __forceinline__ float4 aff4(const float4 x, const float4 mu, const float4 sc, const float4 sh) {
    return make_float4((x.x - mu.x) * sc.x + sh.x, (x.y - mu.y) * sc.y + sh.y, (x.z - mu.z) * sc.z + sh.z,
                       (x.w - mu.w) * sc.w + sh.w);
}

// TM (tower-major output, the tower variant's stacked layers: csrc/tower.hip): the features are `towers` groups of FVT chunks and a
// node's row is [tower][block][feature of the tower] instead of [block][feature] - tower t's nblk blocks are ONE contiguous
// range of columns, the K range of its own posttrans product (i3d_gemm_f32_batched)
template <int MODE, bool MB16 = false, bool TM = false>      // MB16: the messages are stored as bf16 (rows of F bf16 values; bf16 matmul mode)
__global__ void __launch_bounds__(256)
pna_aggregate_fwd_kernel(const float4* __restrict__ e, const int* __restrict__ in_ptr, int N, int FV,
                         AggCfg cfg, float4* __restrict__ out, const float4* __restrict__ aff, int FVT) {
    I3D_CHAIN_PRIO();
    long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long)N * FV) return;
    int v = (int)(t / FV), c = (int)(t - (long)v * FV);
    int beg = in_ptr[v], end = in_ptr[v + 1];
    int D = end - beg;
    const int nblk = cfg.n_agg * cfg.n_scaler;
    float4* o = out + (long)v * nblk * FV + c;
    const int BS = TM ? FVT : FV;                 // distance of two blocks of one feature chunk
    if constexpr (TM) {
        const int tw = c / FVT;
        o = out + (long)v * nblk * FV + (long)tw * nblk * FVT + (c - tw * FVT);
    }
    if (D <= 0) {  // DGL leaves zero rows for isolated nodes
        for (int b = 0; b < nblk; ++b) o[(long)b * BS] = make_float4(0.f, 0.f, 0.f, 0.f);
        return;
    }
    const float4* p = e + (long)beg * FV + c;
    const uint2* p16 = reinterpret_cast<const uint2*>(e) + (long)beg * FV + c;      // (MB16: a chunk of 4 values is 8 bytes)
    auto msg = [&](long j) -> float4 {
        if constexpr (MB16) {
            const uint2 t = p16[j * FV];
            return make_float4(__uint_as_float(t.x << 16), __uint_as_float(t.x & 0xffff0000u), __uint_as_float(t.y << 16),
                               __uint_as_float(t.y & 0xffff0000u));
        } else {
            return p[j * FV];
        }
    };
    // molecules: D <= 4 almost always.  The first four message rows are loaded unconditionally (clamped index, the
    // duplicates hit L1) so four loads are in flight instead of a dependent load-accumulate chain; the accumulation
    // order stays j = 0 .. D-1.
    float4 x = msg(0);
    float4 x1 = msg(min(1, D - 1)), x2 = msg(min(2, D - 1)), x3 = msg(min(3, D - 1));
    float4 a_mu, a_sc, a_sh;
    if (aff != nullptr) {
        a_mu = aff[c]; a_sc = aff[FV + c]; a_sh = aff[2 * FV + c];
        x = aff4(x, a_mu, a_sc, a_sh); x1 = aff4(x1, a_mu, a_sc, a_sh); x2 = aff4(x2, a_mu, a_sc, a_sh); x3 = aff4(x3, a_mu, a_sc, a_sh);
    }
    float4 sum = x, mx = x, mn = x;
    float4 sq = make_float4(x.x * x.x, x.y * x.y, x.z * x.z, x.w * x.w);
#define I3D_AGG_ACC(X)                                                                                              \
    do {                                                                                                            \
        sum.x += (X).x; sum.y += (X).y; sum.z += (X).z; sum.w += (X).w;                                             \
        sq.x += (X).x * (X).x; sq.y += (X).y * (X).y; sq.z += (X).z * (X).z; sq.w += (X).w * (X).w;                 \
        mx.x = fmaxf(mx.x, (X).x); mx.y = fmaxf(mx.y, (X).y); mx.z = fmaxf(mx.z, (X).z); mx.w = fmaxf(mx.w, (X).w); \
        mn.x = fminf(mn.x, (X).x); mn.y = fminf(mn.y, (X).y); mn.z = fminf(mn.z, (X).z); mn.w = fminf(mn.w, (X).w); \
    } while (0)
    if (D > 1) I3D_AGG_ACC(x1);
    if (D > 2) I3D_AGG_ACC(x2);
    if (D > 3) I3D_AGG_ACC(x3);
    // longer segments (per-graph readouts: ~18 atoms; hub atoms): four more rows in flight per trip, same order
    for (int j = 4; j < D; j += 4) {
        float4 y0 = msg(j), y1 = msg(min(j + 1, D - 1)), y2 = msg(min(j + 2, D - 1)), y3 = msg(min(j + 3, D - 1));
        if (aff != nullptr) {
            y0 = aff4(y0, a_mu, a_sc, a_sh); y1 = aff4(y1, a_mu, a_sc, a_sh); y2 = aff4(y2, a_mu, a_sc, a_sh); y3 = aff4(y3, a_mu, a_sc, a_sh);
        }
        I3D_AGG_ACC(y0);
        if (j + 1 < D) I3D_AGG_ACC(y1);
        if (j + 2 < D) I3D_AGG_ACC(y2);
        if (j + 3 < D) I3D_AGG_ACC(y3);
    }
#undef I3D_AGG_ACC
    const float fD = (float)D;
    float4 mean = make_float4(sum.x / fD, sum.y / fD, sum.z / fD, sum.w / fD);
    float4 msq = make_float4(sq.x / fD, sq.y / fD, sq.z / fD, sq.w / fD);
    float4 var = make_float4(fmaxf(msq.x - mean.x * mean.x, 0.f), fmaxf(msq.y - mean.y * mean.y, 0.f),
                             fmaxf(msq.z - mean.z * mean.z, 0.f), fmaxf(msq.w - mean.w * mean.w, 0.f));
    float amp, att;
    scaler_values(cfg, D, amp, att);
    if (MODE == 2) {
        o[0] = mean;
        o[(long)BS] = mx;
        o[(long)2 * BS] = mn;
        o[(long)3 * BS] = make_float4(sqrtf(var.x + 1e-5f), sqrtf(var.y + 1e-5f), sqrtf(var.z + 1e-5f), sqrtf(var.w + 1e-5f));
    } else if (MODE == 1) {  // fully unrolled
        float4 sd = make_float4(sqrtf(var.x + 1e-5f), sqrtf(var.y + 1e-5f), sqrtf(var.z + 1e-5f), sqrtf(var.w + 1e-5f));
        float4 a[4] = {mean, mx, mn, sd};
#pragma unroll
        for (int k = 0; k < 4; ++k) o[(long)k * BS] = a[k];
#pragma unroll
        for (int k = 0; k < 4; ++k)
            o[(long)(4 + k) * BS] = make_float4(a[k].x * amp, a[k].y * amp, a[k].z * amp, a[k].w * amp);
#pragma unroll
        for (int k = 0; k < 4; ++k)
            o[(long)(8 + k) * BS] = make_float4(a[k].x * att, a[k].y * att, a[k].z * att, a[k].w * att);
    } else {
        for (int s = 0; s < cfg.n_scaler; ++s) {
            float sc = cfg.scaler[s] == I3D_SCALE_AMPLIFICATION ? amp
                       : (cfg.scaler[s] == I3D_SCALE_ATTENUATION ? att : 1.f);
            for (int k = 0; k < cfg.n_agg; ++k) {
                float4 a;
                switch (cfg.agg[k]) {
                    case I3D_AGG_MEAN: a = mean; break;
                    case I3D_AGG_SUM: a = sum; break;
                    case I3D_AGG_MAX: a = mx; break;
                    case I3D_AGG_MIN: a = mn; break;
                    case I3D_AGG_VAR: a = var; break;
                    default:
                        a = make_float4(sqrtf(var.x + 1e-5f), sqrtf(var.y + 1e-5f), sqrtf(var.z + 1e-5f),
                                        sqrtf(var.w + 1e-5f));
                }
                if (cfg.scaler[s] != I3D_SCALE_IDENTITY) a = make_float4(a.x * sc, a.y * sc, a.z * sc, a.w * sc);
                o[(long)(s * cfg.n_agg + k) * BS] = a;
            }
        }
    }
}

// scalar-feature fallback (F % 4 != 0)
__global__ void __launch_bounds__(256)
pna_aggregate_fwd_scalar_kernel(const float* __restrict__ e, const int* __restrict__ in_ptr, int N, int F,
                                AggCfg cfg, float* __restrict__ out) {
    I3D_CHAIN_PRIO();
    long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long)N * F) return;
    int v = (int)(t / F), c = (int)(t - (long)v * F);
    int beg = in_ptr[v], D = in_ptr[v + 1] - beg;
    const int nblk = cfg.n_agg * cfg.n_scaler;
    float* o = out + (long)v * nblk * F + c;
    if (D <= 0) {
        for (int b = 0; b < nblk; ++b) o[(long)b * F] = 0.f;
        return;
    }
    const float* p = e + (long)beg * F + c;
    float x = *p, sum = x, sq = x * x, mx = x, mn = x;
    for (int j = 1; j < D; ++j) {
        x = p[(long)j * F];
        sum += x; sq += x * x; mx = fmaxf(mx, x); mn = fminf(mn, x);
    }
    float mean = sum / (float)D, msq = sq / (float)D;
    float var = fmaxf(msq - mean * mean, 0.f);
    float amp, att;
    scaler_values(cfg, D, amp, att);
    for (int s = 0; s < cfg.n_scaler; ++s) {
        float sc = cfg.scaler[s] == I3D_SCALE_AMPLIFICATION ? amp : (cfg.scaler[s] == I3D_SCALE_ATTENUATION ? att : 1.f);
        for (int k = 0; k < cfg.n_agg; ++k) {
            float a;
            switch (cfg.agg[k]) {
                case I3D_AGG_MEAN: a = mean; break;
                case I3D_AGG_SUM: a = sum; break;
                case I3D_AGG_MAX: a = mx; break;
                case I3D_AGG_MIN: a = mn; break;
                case I3D_AGG_VAR: a = var; break;
                default: a = sqrtf(var + 1e-5f);
            }
            if (cfg.scaler[s] != I3D_SCALE_IDENTITY) a *= sc;
            o[(long)(s * cfg.n_agg + k) * F] = a;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// backward: d e[j] = g_mean/D + g_sum + g_max*[j==first argmax] + g_min*[j==first argmin]
//                    + g_std*(x_j-mean)/(D*std)*[var_raw>0] + g_var*2(x_j-mean)/D*[var_raw>0]
// where g_a = sum over scaler blocks of scale_s * dOut[s, a].  Tie routing = first index, as
// torch.max/min(dim) backward on CPU; relu'(0) = 0 as torch.relu backward.
// Algorithmic bytes: 4*N*12F (dOut) + 4*E*F (re-read e) + 4*E*F (write de) + 4*(N+1).
// ---------------------------------------------------------------------------------------------
template <typename T, int V>
__device__ __forceinline__ float& comp(T& a, int i) {
    return reinterpret_cast<float*>(&a)[i];
}

// MODE as in the forward kernel (0: any list, 1: standard 12 blocks, 2: standard 4 blocks); MODE 1/2 need V = 4
template <int V, int MODE = 0, bool MB16 = false, bool TM = false>  // V = 4 (float4 items) or 1; MB16 (V = 4): messages stored as bf16; TM: tower-major gout
// (the [N, 4F] form of the step sits one register above five waves per SIMD when left to the allocator: asked for)
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(V == 4 && MODE == 2 ? 5 : 1)))
pna_aggregate_bwd_kernel(const float* __restrict__ gout, const float* __restrict__ e,
                         const int* __restrict__ in_ptr, int N, int F, AggCfg cfg, float* __restrict__ ge,
                         const float* __restrict__ aff, int FT) {
    I3D_CHAIN_PRIO();
    const int FV = F / V;
    const int BS = TM ? FT : F;       // distance (floats) of two blocks of one feature in a node's gradient row
    // mean | scale | shift of the messages' BatchNorm is ONE [3F] vector for the whole launch: staged in LDS once per
    // workgroup (round 2 loaded it per (node, chunk) item: three more 16-byte global loads per lane, 12.3 -> 16.8 us).
    // Its global loads are issued FIRST and parked in registers; the LDS stores and the barrier come after the lane has
    // issued its index, gradient and message loads - in front of them the staging was one more dependent memory round
    // trip per workgroup (round 4: 170 -> 14x us at batch 8192, where the plain kernel needs 144)
    // (one 16-byte item per thread: F <= 340; wider layers take the per-lane global loads - two more staging registers per lane
    // cost the kernel a wave of occupancy)
    constexpr int AFF_MAX_F = 340;
    __shared__ __attribute__((aligned(16))) float affs[V == 4 ? 3 * AFF_MAX_F : 4];
    const bool aff_lds = V == 4 && aff != nullptr && F <= AFF_MAX_F;
    static_assert(3 * (AFF_MAX_F / 4) <= 256, "one staged 16-byte item per thread");
    const int si0 = threadIdx.x;
    float4 st0;      // (clamped index, unconditional load: a select between a global and a private address is a flat load)
    if (aff_lds) st0 = reinterpret_cast<const float4*>(aff)[min(si0, 3 * FV - 1)];
    long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const bool in_range = t < (long)N * FV;
    int v = in_range ? (int)(t / FV) : 0, c = in_range ? (int)(t - (long)v * FV) : 0;
    int beg = in_ptr[v], D = in_ptr[v + 1] - beg;
    // lanes without work still take part in the barrier: they issue the same loads at addresses that exist (node 0's or
    // their own node's gradient rows) and leave after it
    const bool live = in_range && D > 0;
    if (!live) { beg = 0; D = 1; }
    const int nblk = cfg.n_agg * cfg.n_scaler;
    const float* go = gout + (long)v * nblk * F + (long)c * V;
    if constexpr (TM) {               // row = [tower][block][feature of the tower]
        const int f = c * V, tw = f / FT;
        go = gout + (long)v * nblk * F + (long)tw * nblk * FT + (f - tw * FT);
    }
    // (a lane without work reads its gradient row instead of a message row: a graph without edges has no message row at all)
    const float* p = live ? e + (long)beg * F + (long)c * V : go;
    const unsigned short* p16 = live ? reinterpret_cast<const unsigned short*>(e) + (long)beg * F + (long)c * 4
                                     : reinterpret_cast<const unsigned short*>(go);
    float* q = ge + (long)beg * F + (long)c * V;
    auto load_msg = [&](long row, float* dst) {
        if constexpr (MB16 && V == 4) {
            const uint2 t = *reinterpret_cast<const uint2*>(p16 + row * F);
            dst[0] = __uint_as_float(t.x << 16); dst[1 % V] = __uint_as_float(t.x & 0xffff0000u);
            dst[2 % V] = __uint_as_float(t.y << 16); dst[3 % V] = __uint_as_float(t.y & 0xffff0000u);
        } else if (V == 4) {
            const float4 xx = *reinterpret_cast<const float4*>(p + row * F);
            dst[0] = xx.x; dst[1 % V] = xx.y; dst[2 % V] = xx.z; dst[3 % V] = xx.w;
        } else {
            dst[0] = p[row * F];
        }
    };
    float amp, att;
    scaler_values(cfg, D, amp, att);

    // combined upstream gradient per aggregator kind
    float g_mean[V], g_sum[V], g_max[V], g_min[V], g_std[V], g_var[V];
#pragma unroll
    for (int i = 0; i < V; ++i) g_mean[i] = g_sum[i] = g_max[i] = g_min[i] = g_std[i] = g_var[i] = 0.f;
    if constexpr (MODE != 0 && V == 4) {
        // (mean, max, min, std) x (identity[, amplification, attenuation]): every block of the upstream gradient is loaded
        // up front - 4 or 12 independent 16-byte loads in flight next to the message rows below - instead of one load per
        // trip of a run-time loop; same products and sums, in the same order, as the general loop
        constexpr int NS = (MODE == 1) ? 3 : 1;
        float4 gb[NS][4];
#pragma unroll
        for (int s = 0; s < NS; ++s)
#pragma unroll
            for (int k = 0; k < 4; ++k) gb[s][k] = *reinterpret_cast<const float4*>(go + (long)(s * 4 + k) * BS);
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const float sc = s == 0 ? 1.f : (s == 1 ? amp : att);
            const float m_[4] = {gb[s][0].x, gb[s][0].y, gb[s][0].z, gb[s][0].w};
            const float x_[4] = {gb[s][1].x, gb[s][1].y, gb[s][1].z, gb[s][1].w};
            const float n_[4] = {gb[s][2].x, gb[s][2].y, gb[s][2].z, gb[s][2].w};
            const float d_[4] = {gb[s][3].x, gb[s][3].y, gb[s][3].z, gb[s][3].w};
#pragma unroll
            for (int i = 0; i < V; ++i) {
                g_mean[i] += m_[i] * sc;
                g_max[i] += x_[i] * sc;
                g_min[i] += n_[i] * sc;
                g_std[i] += d_[i] * sc;
            }
        }
    } else
    for (int s = 0; s < cfg.n_scaler; ++s) {
        float sc = cfg.scaler[s] == I3D_SCALE_AMPLIFICATION ? amp : (cfg.scaler[s] == I3D_SCALE_ATTENUATION ? att : 1.f);
        for (int k = 0; k < cfg.n_agg; ++k) {
            float g[V];
            if (V == 4) {
                float4 gg = *reinterpret_cast<const float4*>(go + (long)(s * cfg.n_agg + k) * BS);
                g[0] = gg.x; g[1 % V] = gg.y; g[2 % V] = gg.z; g[3 % V] = gg.w;
            } else {
                g[0] = go[(long)(s * cfg.n_agg + k) * BS];
            }
#pragma unroll
            for (int i = 0; i < V; ++i) {
                float gi = g[i] * sc;
                switch (cfg.agg[k]) {
                    case I3D_AGG_MEAN: g_mean[i] += gi; break;
                    case I3D_AGG_SUM: g_sum[i] += gi; break;
                    case I3D_AGG_MAX: g_max[i] += gi; break;
                    case I3D_AGG_MIN: g_min[i] += gi; break;
                    case I3D_AGG_VAR: g_var[i] += gi; break;
                    default: g_std[i] += gi;
                }
            }
        }
    }
    // pass 1: statistics + first arg-extrema
    float sum[V], sq[V], mx[V], mn[V];
    int amax[V], amin[V];
#pragma unroll
    for (int i = 0; i < V; ++i) { sum[i] = 0.f; sq[i] = 0.f; mx[i] = -INFINITY; mn[i] = INFINITY; amax[i] = 0; amin[i] = 0; }
    // the first four message rows (D <= 4 for almost every atom) are loaded up front - four loads in flight - and
    // stay in registers for pass 2; accumulation order j = 0 .. D-1 as in the forward kernel
    float xr[4][V];
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
        load_msg(min(jj, D - 1), xr[jj]);
    }
    if (aff_lds) {           // every thread of the workgroup; the loads above are in flight: the barrier's wait covers them all at once
        if (si0 < 3 * FV) reinterpret_cast<float4*>(affs)[si0] = st0;
        __syncthreads();
    }
    if (!live) return;
    // messages as the forward saw them: (e - mean) * scale + shift (see pna_aggregate_fwd_kernel); the result is the
    // gradient with respect to THAT value, the BatchNorm backward in front follows in its own kernel
    float a_mu[V], a_sc[V], a_sh[V];
#pragma unroll
    for (int i = 0; i < V; ++i) { a_mu[i] = 0.f; a_sc[i] = 1.f; a_sh[i] = 0.f; }
    if (aff != nullptr) {
        if (V == 4) {
            float4 m4, s4, h4;
            if (aff_lds) {      // (two address spaces: no common pointer, a flat load would wait on both counters)
                m4 = *reinterpret_cast<const float4*>(affs + c * 4);
                s4 = *reinterpret_cast<const float4*>(affs + F + c * 4);
                h4 = *reinterpret_cast<const float4*>(affs + 2 * F + c * 4);
            } else {
                m4 = *reinterpret_cast<const float4*>(aff + (long)c * 4);
                s4 = *reinterpret_cast<const float4*>(aff + F + (long)c * 4);
                h4 = *reinterpret_cast<const float4*>(aff + 2 * F + (long)c * 4);
            }
            a_mu[0] = m4.x; a_mu[1 % V] = m4.y; a_mu[2 % V] = m4.z; a_mu[3 % V] = m4.w;
            a_sc[0] = s4.x; a_sc[1 % V] = s4.y; a_sc[2 % V] = s4.z; a_sc[3 % V] = s4.w;
            a_sh[0] = h4.x; a_sh[1 % V] = h4.y; a_sh[2 % V] = h4.z; a_sh[3 % V] = h4.w;
        } else {
            a_mu[0] = aff[c]; a_sc[0] = aff[F + c]; a_sh[0] = aff[2 * F + c];
        }
    }
    if (aff != nullptr) {
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
#pragma unroll
            for (int i = 0; i < V; ++i) xr[jj][i] = (xr[jj][i] - a_mu[i]) * a_sc[i] + a_sh[i];
    }
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
        if (jj < D) {
#pragma unroll
            for (int i = 0; i < V; ++i) {
                const float xv = xr[jj][i];
                sum[i] += xv;
                sq[i] += xv * xv;
                if (xv > mx[i]) { mx[i] = xv; amax[i] = jj; }
                if (xv < mn[i]) { mn[i] = xv; amin[i] = jj; }
            }
        }
    }
    for (int j0 = 4; j0 < D; j0 += 4) {      // four rows in flight per trip (readouts, hub atoms), same order
        float xs[4][V];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            load_msg(min(j0 + k, D - 1), xs[k]);
            if (aff != nullptr) {
#pragma unroll
                for (int i = 0; i < V; ++i) xs[k][i] = (xs[k][i] - a_mu[i]) * a_sc[i] + a_sh[i];
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int j = j0 + k;
            if (j < D) {
#pragma unroll
                for (int i = 0; i < V; ++i) {
                    const float xv = xs[k][i];
                    sum[i] += xv;
                    sq[i] += xv * xv;
                    if (xv > mx[i]) { mx[i] = xv; amax[i] = j; }
                    if (xv < mn[i]) { mn[i] = xv; amin[i] = j; }
                }
            }
        }
    }
    const float fD = (float)D;
    float mean[V], kstd[V], kvar[V];
#pragma unroll
    for (int i = 0; i < V; ++i) {
        mean[i] = sum[i] / fD;
        float raw = sq[i] / fD - mean[i] * mean[i];
        bool pos = raw > 0.f;
        float sd = sqrtf(fmaxf(raw, 0.f) + 1e-5f);
        kstd[i] = pos ? g_std[i] / (fD * sd) : 0.f;          // d std / d x_j = (x_j - mean) / (D std)
        kvar[i] = pos ? g_var[i] * 2.f / fD : 0.f;           // d var / d x_j = 2 (x_j - mean) / D
        g_mean[i] = g_mean[i] / fD + g_sum[i];
    }
    // pass 2: write gradients (rows 0..3 from registers, the rest from L1/L2)
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
        if (jj < D) {
            float r[V];
#pragma unroll
            for (int i = 0; i < V; ++i) {
                r[i] = g_mean[i] + (kstd[i] + kvar[i]) * (xr[jj][i] - mean[i]);
                if (jj == amax[i]) r[i] += g_max[i];
                if (jj == amin[i]) r[i] += g_min[i];
            }
            if (V == 4) *reinterpret_cast<float4*>(q + (long)jj * F) = make_float4(r[0], r[1 % V], r[2 % V], r[3 % V]);
            else q[(long)jj * F] = r[0];
        }
    }
    for (int j0 = 4; j0 < D; j0 += 4) {
        float xs[4][V];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            load_msg(min(j0 + k, D - 1), xs[k]);
            if (aff != nullptr) {
#pragma unroll
                for (int i = 0; i < V; ++i) xs[k][i] = (xs[k][i] - a_mu[i]) * a_sc[i] + a_sh[i];
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int j = j0 + k;
            if (j < D) {
                float r[V];
#pragma unroll
                for (int i = 0; i < V; ++i) {
                    r[i] = g_mean[i] + (kstd[i] + kvar[i]) * (xs[k][i] - mean[i]);
                    if (j == amax[i]) r[i] += g_max[i];
                    if (j == amin[i]) r[i] += g_min[i];
                }
                if (V == 4) *reinterpret_cast<float4*>(q + (long)j * F) = make_float4(r[0], r[1 % V], r[2 % V], r[3 % V]);
                else q[(long)j * F] = r[0];
            }
        }
    }
}

static int make_cfg(const int* aggregators, int n_agg, const int* scalers, int n_scalers, int force_scalers,
                    float avg_d_log, AggCfg& cfg) {
    if (n_agg < 1 || n_agg > 8 || n_scalers < 1 || n_scalers > 4) return -1;
    cfg.n_agg = n_agg;
    for (int i = 0; i < n_agg; ++i) {
        if (aggregators[i] < 0 || aggregators[i] > I3D_AGG_VAR) return -1;
        cfg.agg[i] = aggregators[i];
    }
    // reference models/pna.py:232: scalers are only applied when more than one is configured
    if (n_scalers == 1 && !force_scalers) {
        cfg.n_scaler = 1;
        cfg.scaler[0] = I3D_SCALE_IDENTITY;
    } else {
        cfg.n_scaler = n_scalers;
        for (int i = 0; i < n_scalers; ++i) {
            if (scalers[i] < 0 || scalers[i] > I3D_SCALE_ATTENUATION) return -1;
            cfg.scaler[i] = scalers[i];
        }
    }
    cfg.avg = avg_d_log;
    cfg.amp[0] = 0.f;
    cfg.att[0] = 0.f;   // D = 0 rows are never scaled (zero rows)
    for (int D = 1; D < 32; ++D) {
        double l = log((double)(D + 1));
        cfg.amp[D] = (float)(l / (double)avg_d_log);
        cfg.att[D] = (float)((double)avg_d_log / l);
    }
    return 0;
}

static bool is_std_aggs(const AggCfg& c) {
    return c.n_agg == 4 && c.agg[0] == I3D_AGG_MEAN && c.agg[1] == I3D_AGG_MAX && c.agg[2] == I3D_AGG_MIN &&
           c.agg[3] == I3D_AGG_STD;
}

static bool is_std_cfg(const AggCfg& c) {
    return is_std_aggs(c) && c.n_scaler == 3 && c.scaler[0] == I3D_SCALE_IDENTITY &&
           c.scaler[1] == I3D_SCALE_AMPLIFICATION && c.scaler[2] == I3D_SCALE_ATTENUATION;
}

static bool is_ident_cfg(const AggCfg& c) {
    return is_std_aggs(c) && c.n_scaler == 1 && c.scaler[0] == I3D_SCALE_IDENTITY;
}

}  // namespace i3d

using namespace i3d;

extern "C" int i3d_pna_aggregate_fwd(const float* e, const int* in_ptr, int num_nodes, int feat,
                                     const int* aggregators, int n_aggregators, const int* scalers,
                                     int n_scalers, int force_scalers, float avg_d_log, float* out, void* stream) {
    return i3d_pna_aggregate_fwd_aff(e, nullptr, in_ptr, num_nodes, feat, aggregators, n_aggregators, scalers, n_scalers,
                                     force_scalers, avg_d_log, out, stream);
}

// Timing of the roofline kernel inside a step (bench.py, roofline.achieved): the events ride ON the forward kernel's own launch
// (hipExtLaunchKernelGGL start / stop events: the dispatch's begin and end timestamps, what rocprofv3's kernel trace reports)
// instead of bracketing it with two event records, whose dispatch + completion signalling is 4.8 us of a 9 us kernel.
// Thread-local, consumed by the next forward launch of this thread.
static thread_local hipEvent_t g_time_start = nullptr, g_time_stop = nullptr;
namespace i3d {
void k4_time_next_launch(void* start, void* stop) { g_time_start = (hipEvent_t)start; g_time_stop = (hipEvent_t)stop; }
}
#define K4_FWD_LAUNCH(KERNEL, GRID, ...)                                                                              \
    do {                                                                                                              \
        if (t_start != nullptr && t_stop != nullptr)                                                                  \
            hipExtLaunchKernelGGL(KERNEL, GRID, dim3(256), 0, s, t_start, t_stop, 0, __VA_ARGS__);                    \
        else                                                                                                          \
            hipLaunchKernelGGL(KERNEL, GRID, dim3(256), 0, s, __VA_ARGS__);                                           \
    } while (0)

extern "C" int i3d_pna_aggregate_fwd_aff(const float* e, const float* aff, const int* in_ptr, int num_nodes, int feat,
                                         const int* aggregators, int n_aggregators, const int* scalers,
                                         int n_scalers, int force_scalers, float avg_d_log, float* out, void* stream) {
    return i3d_pna_aggregate_fwd_ex(e, 0, aff, in_ptr, num_nodes, feat, aggregators, n_aggregators, scalers, n_scalers, force_scalers,
                                    avg_d_log, out, stream);
}

extern "C" int i3d_pna_aggregate_fwd_ex(const void* e_, int e_bf16, const float* aff, const int* in_ptr, int num_nodes, int feat,
                                        const int* aggregators, int n_aggregators, const int* scalers,
                                        int n_scalers, int force_scalers, float avg_d_log, float* out, void* stream) {
    const float* e = (const float*)e_;
    const hipEvent_t t_start = g_time_start, t_stop = g_time_stop;      // (consumed by THIS call whatever it returns)
    g_time_start = g_time_stop = nullptr;
    I3D_CHECK_ARG(!e_bf16 || (feat % 4 == 0 && (((uintptr_t)e_) & 7) == 0), "bf16 messages need feat % 4 == 0 and 8-byte alignment");
    I3D_CHECK_ARG(aff == nullptr || (feat % 4 == 0 && (((uintptr_t)aff) & 15) == 0), "aff needs feat % 4 == 0 and 16-byte alignment");
    I3D_CHECK_ARG(num_nodes >= 0 && feat > 0, "num_nodes >= 0 and feat > 0 required");
    AggCfg cfg;
    I3D_CHECK_ARG(make_cfg(aggregators, n_aggregators, scalers, n_scalers, force_scalers, avg_d_log, cfg) == 0,
                  "bad aggregator/scaler list");
    if (num_nodes == 0) return I3D_OK;
    hipStream_t s = (hipStream_t)stream;
    if (feat % 4 == 0) {
        int FV = feat / 4;
        long items = (long)num_nodes * FV;
        dim3 grid(cdiv(items, 256));
        if (e_bf16) {
            if (is_std_cfg(cfg))
                K4_FWD_LAUNCH((pna_aggregate_fwd_kernel<1, true>), grid, (const float4*)e, in_ptr, num_nodes, FV, cfg, (float4*)out, (const float4*)aff, 0);
            else if (is_ident_cfg(cfg))
                K4_FWD_LAUNCH((pna_aggregate_fwd_kernel<2, true>), grid, (const float4*)e, in_ptr, num_nodes, FV, cfg, (float4*)out, (const float4*)aff, 0);
            else
                K4_FWD_LAUNCH((pna_aggregate_fwd_kernel<0, true>), grid, (const float4*)e, in_ptr, num_nodes, FV, cfg, (float4*)out, (const float4*)aff, 0);
        } else if (is_std_cfg(cfg))
            K4_FWD_LAUNCH(pna_aggregate_fwd_kernel<1>, grid, (const float4*)e, in_ptr, num_nodes, FV, cfg, (float4*)out, (const float4*)aff, 0);
        else if (is_ident_cfg(cfg))      // (tried, not better back to back at batch 512: two items per lane 9.4 us vs 8.7 us;
            //                              one wavefront per node with scalar row-pointer loads 8.8 us vs 8.8 us)
            K4_FWD_LAUNCH(pna_aggregate_fwd_kernel<2>, grid, (const float4*)e, in_ptr, num_nodes, FV, cfg, (float4*)out, (const float4*)aff, 0);
        else
            K4_FWD_LAUNCH(pna_aggregate_fwd_kernel<0>, grid, (const float4*)e, in_ptr, num_nodes, FV, cfg, (float4*)out, (const float4*)aff, 0);
    } else {
        long items = (long)num_nodes * feat;
        K4_FWD_LAUNCH(pna_aggregate_fwd_scalar_kernel, dim3(cdiv(items, 256)), e, in_ptr, num_nodes, feat, cfg, out);
    }
    I3D_CHECK_LAUNCH();
    return I3D_OK;
}

// tower-major forms (include/infomax3d_hip.h): fp32 messages, feat and tower_feat multiples of 4
extern "C" int i3d_pna_aggregate_fwd_towers(const float* e, const int* in_ptr, int num_nodes, int feat, int tower_feat,
                                            const int* aggregators, int n_aggregators, const int* scalers, int n_scalers,
                                            int force_scalers, float avg_d_log, float* out, void* stream) {
    I3D_CHECK_ARG(num_nodes >= 0 && feat > 0 && tower_feat > 0 && feat % 4 == 0 && tower_feat % 4 == 0 && feat % tower_feat == 0,
                  "feat and tower_feat must be multiples of 4, tower_feat a divisor of feat");
    AggCfg cfg;
    I3D_CHECK_ARG(make_cfg(aggregators, n_aggregators, scalers, n_scalers, force_scalers, avg_d_log, cfg) == 0,
                  "bad aggregator/scaler list");
    if (num_nodes == 0) return I3D_OK;
    const int FV = feat / 4;
    dim3 grid(cdiv((long)num_nodes * FV, 256));
    if (is_std_cfg(cfg))
        hipLaunchKernelGGL((pna_aggregate_fwd_kernel<1, false, true>), grid, dim3(256), 0, (hipStream_t)stream, (const float4*)e, in_ptr,
                           num_nodes, FV, cfg, (float4*)out, (const float4*)nullptr, tower_feat / 4);
    else if (is_ident_cfg(cfg))
        hipLaunchKernelGGL((pna_aggregate_fwd_kernel<2, false, true>), grid, dim3(256), 0, (hipStream_t)stream, (const float4*)e, in_ptr,
                           num_nodes, FV, cfg, (float4*)out, (const float4*)nullptr, tower_feat / 4);
    else
        hipLaunchKernelGGL((pna_aggregate_fwd_kernel<0, false, true>), grid, dim3(256), 0, (hipStream_t)stream, (const float4*)e, in_ptr,
                           num_nodes, FV, cfg, (float4*)out, (const float4*)nullptr, tower_feat / 4);
    I3D_CHECK_LAUNCH();
    return I3D_OK;
}

extern "C" int i3d_pna_aggregate_bwd_towers(const float* grad_out, const float* e, const int* in_ptr, int num_nodes, int feat,
                                            int tower_feat, const int* aggregators, int n_aggregators, const int* scalers,
                                            int n_scalers, int force_scalers, float avg_d_log, float* grad_e, void* stream) {
    I3D_CHECK_ARG(num_nodes >= 0 && feat > 0 && tower_feat > 0 && feat % 4 == 0 && tower_feat % 4 == 0 && feat % tower_feat == 0,
                  "feat and tower_feat must be multiples of 4, tower_feat a divisor of feat");
    AggCfg cfg;
    I3D_CHECK_ARG(make_cfg(aggregators, n_aggregators, scalers, n_scalers, force_scalers, avg_d_log, cfg) == 0,
                  "bad aggregator/scaler list");
    if (num_nodes == 0) return I3D_OK;
    const long items = (long)num_nodes * (feat / 4);
    if (is_std_cfg(cfg))
        hipLaunchKernelGGL((pna_aggregate_bwd_kernel<4, 1, false, true>), dim3(cdiv(items, 256)), dim3(256), 0, (hipStream_t)stream,
                           grad_out, e, in_ptr, num_nodes, feat, cfg, grad_e, (const float*)nullptr, tower_feat);
    else if (is_ident_cfg(cfg))
        hipLaunchKernelGGL((pna_aggregate_bwd_kernel<4, 2, false, true>), dim3(cdiv(items, 256)), dim3(256), 0, (hipStream_t)stream,
                           grad_out, e, in_ptr, num_nodes, feat, cfg, grad_e, (const float*)nullptr, tower_feat);
    else
        hipLaunchKernelGGL((pna_aggregate_bwd_kernel<4, 0, false, true>), dim3(cdiv(items, 256)), dim3(256), 0, (hipStream_t)stream,
                           grad_out, e, in_ptr, num_nodes, feat, cfg, grad_e, (const float*)nullptr, tower_feat);
    I3D_CHECK_LAUNCH();
    return I3D_OK;
}

extern "C" int i3d_pna_aggregate_bwd(const float* grad_out, const float* e, const int* in_ptr, int num_nodes,
                                     int feat, const int* aggregators, int n_aggregators, const int* scalers,
                                     int n_scalers, int force_scalers, float avg_d_log, float* grad_e, void* stream) {
    return i3d_pna_aggregate_bwd_aff(grad_out, e, nullptr, in_ptr, num_nodes, feat, aggregators, n_aggregators, scalers,
                                     n_scalers, force_scalers, avg_d_log, grad_e, stream);
}

extern "C" int i3d_pna_aggregate_bwd_aff(const float* grad_out, const float* e, const float* aff, const int* in_ptr,
                                         int num_nodes, int feat, const int* aggregators, int n_aggregators,
                                         const int* scalers, int n_scalers, int force_scalers, float avg_d_log,
                                         float* grad_e, void* stream) {
    return i3d_pna_aggregate_bwd_ex(grad_out, e, 0, aff, in_ptr, num_nodes, feat, aggregators, n_aggregators, scalers, n_scalers,
                                    force_scalers, avg_d_log, grad_e, stream);
}

extern "C" int i3d_pna_aggregate_bwd_ex(const float* grad_out, const void* e_, int e_bf16, const float* aff, const int* in_ptr,
                                        int num_nodes, int feat, const int* aggregators, int n_aggregators,
                                        const int* scalers, int n_scalers, int force_scalers, float avg_d_log,
                                        float* grad_e, void* stream) {
    const float* e = (const float*)e_;
    I3D_CHECK_ARG(!e_bf16 || (feat % 4 == 0 && (((uintptr_t)e_) & 7) == 0), "bf16 messages need feat % 4 == 0 and 8-byte alignment");
    I3D_CHECK_ARG(num_nodes >= 0 && feat > 0, "num_nodes >= 0 and feat > 0 required");
    AggCfg cfg;
    I3D_CHECK_ARG(make_cfg(aggregators, n_aggregators, scalers, n_scalers, force_scalers, avg_d_log, cfg) == 0,
                  "bad aggregator/scaler list");
    if (num_nodes == 0) return I3D_OK;
    hipStream_t s = (hipStream_t)stream;
    if (feat % 4 == 0) {
        long items = (long)num_nodes * (feat / 4);
        if (e_bf16) {
            if (is_std_cfg(cfg))
                hipLaunchKernelGGL((pna_aggregate_bwd_kernel<4, 1, true>), dim3(cdiv(items, 256)), dim3(256), 0, s, grad_out, e, in_ptr,
                                   num_nodes, feat, cfg, grad_e, aff, 0);
            else if (is_ident_cfg(cfg))
                hipLaunchKernelGGL((pna_aggregate_bwd_kernel<4, 2, true>), dim3(cdiv(items, 256)), dim3(256), 0, s, grad_out, e, in_ptr,
                                   num_nodes, feat, cfg, grad_e, aff, 0);
            else
                hipLaunchKernelGGL((pna_aggregate_bwd_kernel<4, 0, true>), dim3(cdiv(items, 256)), dim3(256), 0, s, grad_out, e, in_ptr,
                                   num_nodes, feat, cfg, grad_e, aff, 0);
        } else if (is_std_cfg(cfg))
            hipLaunchKernelGGL((pna_aggregate_bwd_kernel<4, 1>), dim3(cdiv(items, 256)), dim3(256), 0, s, grad_out, e, in_ptr,
                               num_nodes, feat, cfg, grad_e, aff, 0);
        else if (is_ident_cfg(cfg))
            hipLaunchKernelGGL((pna_aggregate_bwd_kernel<4, 2>), dim3(cdiv(items, 256)), dim3(256), 0, s, grad_out, e, in_ptr,
                               num_nodes, feat, cfg, grad_e, aff, 0);
        else
            hipLaunchKernelGGL((pna_aggregate_bwd_kernel<4, 0>), dim3(cdiv(items, 256)), dim3(256), 0, s, grad_out, e, in_ptr,
                               num_nodes, feat, cfg, grad_e, aff, 0);
    } else {
        long items = (long)num_nodes * feat;
        hipLaunchKernelGGL((pna_aggregate_bwd_kernel<1, 0>), dim3(cdiv(items, 256)), dim3(256), 0, s, grad_out, e, in_ptr,
                           num_nodes, feat, cfg, grad_e, aff, 0);
    }
    I3D_CHECK_LAUNCH();
    return I3D_OK;
}

// debug / test entry: the messages exactly as the aggregation kernels see them, (e - mean) * scale + shift with the same
// expression (and the same -ffp-contract=off build) - a test derives the kernels' arg-max / arg-min choices from them
__global__ void __launch_bounds__(256) pna_messages_kernel(const float* __restrict__ e, const float* __restrict__ aff, long rows,
                                                           int F, float* __restrict__ out, int e_bf16) {
    I3D_CHAIN_PRIO();
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= rows * F) return;
    const int c = (int)(t % F);
    float a_mu = 0.f, a_sc = 1.f, a_sh = 0.f;
    if (aff != nullptr) { a_mu = aff[c]; a_sc = aff[F + c]; a_sh = aff[2 * F + c]; }
    const float v = e_bf16 ? __uint_as_float((unsigned)reinterpret_cast<const unsigned short*>(e)[t] << 16) : e[t];
    out[t] = aff != nullptr ? (v - a_mu) * a_sc + a_sh : v;
}

extern "C" int i3d_pna_messages_normalized(const float* e, const float* aff, long rows, int feat, float* out, void* stream) {
    return i3d_pna_messages_normalized_ex(e, 0, aff, rows, feat, out, stream);
}

extern "C" int i3d_pna_messages_normalized_ex(const void* e, int e_bf16, const float* aff, long rows, int feat, float* out, void* stream) {
    I3D_CHECK_ARG(e != nullptr && out != nullptr && rows >= 0 && feat > 0, "bad arguments");
    if (rows == 0) return I3D_OK;
    hipLaunchKernelGGL(pna_messages_kernel, dim3(cdiv(rows * feat, 256)), dim3(256), 0, (hipStream_t)stream, (const float*)e, aff, rows, feat, out,
                       e_bf16);
    I3D_CHECK_LAUNCH();
    return I3D_OK;
}

// ---- K6: per-graph readout = the same segmented reduction driven by graph_ptr, no scalers ---------------
// reference models/pna.py:133-134, models/net3d.py:73-74 (dgl.readout_nodes for op in readout_aggregators + cat)
extern "C" int i3d_segment_readout_fwd(const float* x, const int* graph_ptr, int num_graphs, int feat, const int* ops,
                                       int n_ops, float* out, void* stream) {
    for (int i = 0; i < n_ops; ++i)
        I3D_CHECK_ARG(ops[i] == I3D_AGG_MEAN || ops[i] == I3D_AGG_SUM || ops[i] == I3D_AGG_MAX || ops[i] == I3D_AGG_MIN,
                      "readout op must be mean/sum/max/min");
    int ident = I3D_SCALE_IDENTITY;
    return i3d_pna_aggregate_fwd(x, graph_ptr, num_graphs, feat, ops, n_ops, &ident, 1, 0, 1.0f, out, stream);
}

extern "C" int i3d_segment_readout_bwd(const float* grad_out, const float* x, const int* graph_ptr, int num_graphs,
                                       int feat, const int* ops, int n_ops, float* grad_x, void* stream) {
    int ident = I3D_SCALE_IDENTITY;
    return i3d_pna_aggregate_bwd(grad_out, x, graph_ptr, num_graphs, feat, ops, n_ops, &ident, 1, 0, 1.0f, grad_x, stream);
}
