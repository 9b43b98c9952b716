// BatchNorm out of the memory path (round 2).
//
// The reference's FCLayer is Linear -> activation -> BatchNorm1d (models/base_layers.py:100-111).  Round 1 ran it as
// "producer kernel -> statistics pass -> apply pass": every [E, F] activation of a PNA layer was written once and
// re-read / re-written twice more.  Here
//   * the column statistics are produced by the kernel that PRODUCES the activation - the GEMM epilogue
//     (gemm.hip, FUSE bit 1) or the edge gather-combine below - as per-row-tile partials
//         partial[tile][0][c] = sum of the tile's rows,  [1][c] = M2 about the TILE mean,  [2][c] = row count
//     (centred per tile: no E[x^2]-E[x]^2 cancellation whatever the mean/std ratio of the column);
//   * bn_finalize_partials_kernel merges the tiles exactly (parallel-axis theorem, fp64, fixed order), updates the
//     running statistics and writes `aff` = mean | gamma*invstd | beta;
//   * the CONSUMER applies (x - mean) * (gamma invstd) + beta while it loads x: the next GEMM in its LDS staging
//     (gemm.hip, FUSE bit 0), the aggregation kernel in its message loads (aggregate.hip) - the normalised activation is
//     never materialised.
// Deterministic: fixed tile -> lane -> tree order everywhere, no atomics.
#include "common.h"
#include "peer.h"
#include <algorithm>

namespace i3d {

constexpr int FIN_COLS = 8, FIN_LANES = 32;

// 256 threads = 8 columns x 32 tile-lanes.  Two passes over the lane's tiles, no fp64 division in the loops:
//   1. count and sum of the lane's tiles -> LDS -> every lane adds the 32 lane sums in lane order: the column mean;
//   2. M2 = sum over tiles of  M2_b + (S_b - n_b mean)^2 / n_b  (the parallel-axis term, exact algebra; 1 / n_b of a row
//      count <= 64 through the fp32 reciprocal, 6e-8 relative) -> LDS -> lane 0 adds the 32 values in lane order.
constexpr int FIN_KEEP = 16;      // tiles per lane kept in registers between the passes (more: re-read, L2-resident)
constexpr int FIN_TAIL = 8;       // tiles per lane and trip of the re-read loops

struct MergedTiles {
    double tot, m2, n, mu;      // m2 is valid in the threads with ly == 0 only
};

// the exact merge of `n_tiles` {sum, M2, count} tiles (tile b at partial + b * tile_stride) of column c; every thread of the
// workgroup calls it (two barriers inside; a barrier must separate two calls: `sm` is reused)
__device__ __forceinline__ MergedTiles merge_tiles(const float* __restrict__ partial, long tile_stride, int n_tiles, int feat, int c,
                                                   bool live, int cx, int ly, double (&sm)[2][FIN_LANES][FIN_COLS]) {
    float ks[FIN_KEEP], km[FIN_KEEP], kn[FIN_KEEP];
    double n_l = 0.0, s_l = 0.0;
#pragma unroll
    for (int k = 0; k < FIN_KEEP; ++k) {
        const int b = ly + k * FIN_LANES;
        const bool ok = live && b < n_tiles;
        const float* p = partial + (long)(ok ? b : 0) * tile_stride + (live ? c : 0);
        ks[k] = *p; km[k] = p[feat]; kn[k] = ok ? p[2 * feat] : 0.f;
    }
#pragma unroll
    for (int k = 0; k < FIN_KEEP; ++k) {
        if (kn[k] > 0.f) { n_l += (double)kn[k]; s_l += (double)ks[k]; }
    }
    // beyond the kept tiles (batches of several thousand molecules, the 3D network's edges): FIN_TAIL independent loads in
    // flight per trip - one load per trip made this loop 100 us at 2000 tiles
    if (live) {
        for (int b0 = ly + FIN_KEEP * FIN_LANES; b0 < n_tiles; b0 += FIN_TAIL * FIN_LANES) {
            float ts[FIN_TAIL], tn[FIN_TAIL];
#pragma unroll
            for (int u = 0; u < FIN_TAIL; ++u) {
                const int b = b0 + u * FIN_LANES;
                const bool ok = b < n_tiles;
                const float* p = partial + (long)(ok ? b : 0) * tile_stride + c;
                ts[u] = *p; tn[u] = ok ? p[2 * feat] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < FIN_TAIL; ++u)
                if (tn[u] > 0.f) { n_l += (double)tn[u]; s_l += (double)ts[u]; }
        }
    }
    sm[0][ly][cx] = n_l; sm[1][ly][cx] = s_l;
    __syncthreads();
    MergedTiles r;
    r.n = 0.0; r.tot = 0.0; r.m2 = 0.0;
#pragma unroll
    for (int k = 0; k < FIN_LANES; ++k) { r.n += sm[0][k][cx]; r.tot += sm[1][k][cx]; }
    const double mu = r.n > 0.0 ? r.tot / r.n : 0.0;
    r.mu = mu;
    __syncthreads();
    double m2_l = 0.0;
#pragma unroll
    for (int k = 0; k < FIN_KEEP; ++k) {
        if (kn[k] > 0.f) {
            const double d = (double)ks[k] - (double)kn[k] * mu;
            m2_l += (double)km[k] + d * d * (double)__frcp_rn(kn[k]);
        }
    }
    if (live) {
        for (int b0 = ly + FIN_KEEP * FIN_LANES; b0 < n_tiles; b0 += FIN_TAIL * FIN_LANES) {
            float ts[FIN_TAIL], tm[FIN_TAIL], tn[FIN_TAIL];
#pragma unroll
            for (int u = 0; u < FIN_TAIL; ++u) {
                const int b = b0 + u * FIN_LANES;
                const bool ok = b < n_tiles;
                const float* p = partial + (long)(ok ? b : 0) * tile_stride + c;
                ts[u] = *p; tm[u] = p[feat]; tn[u] = ok ? p[2 * feat] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < FIN_TAIL; ++u) {
                if (tn[u] > 0.f) {
                    const double d = (double)ts[u] - (double)tn[u] * mu;
                    m2_l += (double)tm[u] + d * d * (double)__frcp_rn(tn[u]);
                }
            }
        }
    }
    sm[0][ly][cx] = m2_l;
    __syncthreads();
    if (ly == 0) {
#pragma unroll
        for (int k = 0; k < FIN_LANES; ++k) r.m2 += sm[0][k][cx];
    }
    return r;
}

// PEER (synchronised BatchNorm through the peer-write exchange, peer.h): the workgroup merges this rank's tiles of its 8
// columns into one {sum, M2, count} triple, writes it into every rank's mailbox, waits for every rank's triple of the same
// columns and merges those `world` triples with the same code - the exchange sits INSIDE the finalisation, no launch
// and no communicator on the chain.  The result has the bits of the two-launch form (local merge -> all-gather -> merge
// over `world` tiles of pitch 3 feat), which the other providers run.
template <bool PEER>
__global__ void __launch_bounds__(256)
bn_finalize_partials_kernel(const float* __restrict__ partial, int n_tiles, int feat, float eps, float momentum,
                            const float* __restrict__ gamma, const float* __restrict__ beta, float* __restrict__ mean,
                            float* __restrict__ invstd, float* running_mean, float* running_var,
                            long long* batches_tracked, float* __restrict__ aff, float* __restrict__ triple_out, const PeerDev peer) {
    I3D_CHAIN_PRIO();
    __shared__ double sm[2][FIN_LANES][FIN_COLS];
    __shared__ float gathered[PEER ? PEER_MAX_WORLD : 1][3][FIN_COLS];
    const int cx = threadIdx.x & (FIN_COLS - 1), ly = threadIdx.x / FIN_COLS;
    const int c = blockIdx.x * FIN_COLS + cx;
    const bool live = c < feat;
    MergedTiles r = merge_tiles(partial, 3L * feat, n_tiles, feat, c, live, cx, ly, sm);
    if (PEER) {
        if (ly == 0 && live) {
            const float tot = (float)r.tot, m2 = (float)r.m2, n = (float)r.n;
            for (int p = 0; p < peer.world; ++p) {
                peer_put_f32(peer, p, c, tot);
                peer_put_f32(peer, p, feat + c, m2);
                peer_put_f32(peer, p, 2 * feat + c, n);
            }
        }
        // lane `ly` < world fetches rank ly's triple of its column (the words say when they are there) into LDS: the `world`
        // triples are then merged as tiles of pitch 3 FIN_COLS by the same code as the rank's own tiles
        __syncthreads();                                 // (separates the two uses of `sm`)
        if (ly < peer.world && live) {
            float t[3];
            peer_get3_f32(peer, ly, c, feat + c, 2 * feat + c, t);
            gathered[ly][0][cx] = t[0]; gathered[ly][1][cx] = t[1]; gathered[ly][2][cx] = t[2];
        }
        __syncthreads();
        r = merge_tiles(&gathered[0][0][0], 3L * FIN_COLS, peer.world, FIN_COLS, cx, live, cx, ly, sm);
    }
    if (ly != 0 || !live) return;
    const double tot = r.tot, m2 = r.m2, n = r.n, mu = r.mu;
    if (!PEER && triple_out != nullptr) {        // synchronised BatchNorm: this rank's tiles merged into ONE {sum, M2, count} "tile"
        triple_out[c] = (float)tot;
        triple_out[feat + c] = (float)m2;
        triple_out[2 * feat + c] = (float)n;
        return;
    }
    const double nn = n > 0.0 ? n : 1.0;
    const double var = m2 / nn;
    const float muf = (float)mu, is = (float)(1.0 / sqrt(var + (double)eps));
    mean[c] = muf;
    invstd[c] = is;
    if (running_mean != nullptr) {
        const double unbiased = n > 1.0 ? m2 / (n - 1.0) : var;
        running_mean[c] = (float)((1.0 - momentum) * (double)running_mean[c] + momentum * mu);
        running_var[c] = (float)((1.0 - momentum) * (double)running_var[c] + momentum * unbiased);
    }
    if (c == 0 && batches_tracked != nullptr) *batches_tracked += 1;
    if (aff != nullptr) {
        aff[c] = muf;
        aff[feat + c] = gamma[c] * is;
        aff[2 * feat + c] = beta[c];
    }
}

// ---- edge gather-combine + activation + statistics ----------------------------------------------------------------
// x[j,:] = act(P[src[j], 0:F] + P[dst[j], F:2F] + Q[q_code ? q_code[j] : j, :] + bias)   (edge.hip: edge_combine_fwd)
// and the per-tile column statistics of x.  A thread owns one column vector and ECS_RPT rows (kept in registers for the
// centred second moment), the rows of a workgroup are consecutive: tile = rl * ECS_RPT rows.
constexpr int ECS_RPT = 8;

struct EcsTiling {
    int tpr, rl, cv, ncolblk, rows_per_tile;
};

static EcsTiling ecs_tiling(int feat, int V) {
    EcsTiling t;
    t.cv = feat / V;
    t.tpr = t.cv < 256 ? t.cv : 256;
    t.rl = 256 / t.tpr;
    t.ncolblk = cdiv(t.cv, t.tpr);
    t.rows_per_tile = t.rl * ECS_RPT;
    return t;
}

template <int V>
__global__ void __launch_bounds__(256)
edge_combine_act_stats_kernel(const float* __restrict__ P, int ldp, const float* __restrict__ Q, const int* __restrict__ q_code,
                              const float* __restrict__ bias, const int* __restrict__ src, const int* __restrict__ dst, int E,
                              int feat, int act, EcsTiling tl, float* __restrict__ x_out, float* __restrict__ partial) {
    I3D_CHAIN_PRIO();
    __shared__ float sm[256 * 4];
    __shared__ float smean[256 * 4];
    const int t = threadIdx.x;
    const int cl = t % tl.tpr, rlane = t / tl.tpr;
    const int cvi = blockIdx.y * tl.tpr + cl;
    const bool active = rlane < tl.rl && cvi < tl.cv;
    const int c0 = cvi * V;
    const int row0 = blockIdx.x * tl.rows_per_tile;
    float x[ECS_RPT][V];
    float s1[V];
#pragma unroll
    for (int i = 0; i < V; ++i) s1[i] = 0.f;
    float bb[V];
#pragma unroll
    for (int i = 0; i < V; ++i) bb[i] = (active && bias != nullptr) ? bias[c0 + i] : 0.f;
    if (active) {
#pragma unroll
        for (int h = 0; h < ECS_RPT; h += 4) {          // four rows (their indices, then their three gathers) in flight
            long ps[4], pd[4], pq[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int j = min(row0 + rlane + (h + u) * tl.rl, E - 1);
                ps[u] = (long)src[j] * ldp + c0;
                pd[u] = (long)dst[j] * ldp + feat + c0;
                pq[u] = (long)(q_code ? q_code[j] : j) * feat + c0;
            }
            float a[4][V], b[4][V], d[4][V];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (V == 4) {
                    const float4 va = *reinterpret_cast<const float4*>(P + ps[u]);
                    const float4 vb = *reinterpret_cast<const float4*>(P + pd[u]);
                    a[u][0] = va.x; a[u][1 % V] = va.y; a[u][2 % V] = va.z; a[u][3 % V] = va.w;
                    b[u][0] = vb.x; b[u][1 % V] = vb.y; b[u][2 % V] = vb.z; b[u][3 % V] = vb.w;
                    if (Q != nullptr) {
                        const float4 vq = *reinterpret_cast<const float4*>(Q + pq[u]);
                        d[u][0] = vq.x; d[u][1 % V] = vq.y; d[u][2 % V] = vq.z; d[u][3 % V] = vq.w;
                    }
                } else {
                    a[u][0] = P[ps[u]];
                    b[u][0] = P[pd[u]];
                    if (Q != nullptr) d[u][0] = Q[pq[u]];
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int j = row0 + rlane + (h + u) * tl.rl;
                const bool live = j < E;
#pragma unroll
                for (int i = 0; i < V; ++i) {
                    float r = a[u][i] + b[u][i] + (Q != nullptr ? d[u][i] : 0.f);       // summation order of edge_combine_fwd
                    if (bias != nullptr) r += bb[i];
                    r = apply_act_c<false>(r, act);      // (none / ReLU / LeakyReLU: checked by the host)
                    x[h + u][i] = live ? r : 0.f;
                    if (live) s1[i] += r;
                }
                if (live) {
                    float* o = x_out + (long)j * feat + c0;
                    if (V == 4) *reinterpret_cast<float4*>(o) = make_float4(x[h + u][0], x[h + u][1 % V], x[h + u][2 % V], x[h + u][3 % V]);
                    else o[0] = x[h + u][0];
                }
            }
        }
    }
    // tile sum -> tile mean (row lanes combined in lane order)
#pragma unroll
    for (int i = 0; i < V; ++i) sm[t * V + i] = s1[i];
    __syncthreads();
    const int n_rows = max(min(tl.rows_per_tile, E - row0), 0);
    if (rlane == 0 && cvi < tl.cv) {
#pragma unroll
        for (int i = 0; i < V; ++i) {
            float tot = 0.f;
            for (int k = 0; k < tl.rl; ++k) tot += sm[(k * tl.tpr + cl) * V + i];
            smean[cl * V + i] = tot;
        }
    }
    __syncthreads();
    float m2[V];
#pragma unroll
    for (int i = 0; i < V; ++i) m2[i] = 0.f;
    if (active) {
#pragma unroll
        for (int i = 0; i < V; ++i) {
            const float mu = smean[cl * V + i] / (float)max(n_rows, 1);
#pragma unroll
            for (int k = 0; k < ECS_RPT; ++k) {
                const float dlt = x[k][i] - mu;
                if (row0 + rlane + k * tl.rl < E) m2[i] += dlt * dlt;
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < V; ++i) sm[t * V + i] = m2[i];
    __syncthreads();
    if (rlane == 0 && cvi < tl.cv) {
        float* o = partial + (long)blockIdx.x * 3 * feat + c0;
#pragma unroll
        for (int i = 0; i < V; ++i) {
            float tot = 0.f;
            for (int k = 0; k < tl.rl; ++k) tot += sm[(k * tl.tpr + cl) * V + i];
            o[i] = smean[cl * V + i];
            o[feat + i] = tot;
            o[2 * feat + i] = (float)n_rows;
        }
    }
}

// eval mode: the affine vectors mean | gamma / sqrt(var + eps) | beta of up to 64 BatchNorms from their running
// statistics, one launch (they depend on parameters and buffers only)
struct EvalAffTable {
    int n;
    I3dBnEvalAff e[64];
};

__global__ void __launch_bounds__(256) bn_eval_aff_kernel(const EvalAffTable t) {
    I3D_CHAIN_PRIO();
    const I3dBnEvalAff& e = t.e[blockIdx.y];
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= e.feat) return;
    e.aff[c] = e.running_mean[c];
    e.aff[e.feat + c] = e.gamma[c] / sqrtf(e.running_var[c] + e.eps);
    e.aff[2 * e.feat + c] = e.beta[c];
}

}  // namespace i3d

using namespace i3d;

extern "C" int i3d_bn_eval_aff_multi(const I3dBnEvalAff* entries, int n, void* stream) {
    I3D_CHECK_ARG(entries != nullptr && n >= 1 && n <= 64, "1..64 entries");
    EvalAffTable t;
    t.n = n;
    int fmax = 0;
    for (int i = 0; i < n; ++i) {
        I3D_CHECK_ARG(entries[i].feat > 0 && entries[i].aff != nullptr && entries[i].running_mean != nullptr &&
                          entries[i].running_var != nullptr && entries[i].gamma != nullptr && entries[i].beta != nullptr, "null");
        t.e[i] = entries[i];
        fmax = std::max(fmax, entries[i].feat);
    }
    hipLaunchKernelGGL(bn_eval_aff_kernel, dim3(cdiv(fmax, 256), n), dim3(256), 0, (hipStream_t)stream, t);
    I3D_CHECK_LAUNCH();
    return I3D_OK;
}

extern "C" int i3d_bn_finalize_partials(const float* partial, int n_tiles, int feat, float eps, float momentum,
                                        const float* gamma, const float* beta, float* mean, float* invstd,
                                        float* running_mean, float* running_var, long long* num_batches_tracked,
                                        float* aff, void* stream) {
    I3D_CHECK_ARG(partial != nullptr && n_tiles > 0 && feat > 0 && mean != nullptr && invstd != nullptr, "bad arguments");
    I3D_CHECK_ARG(aff == nullptr || (gamma != nullptr && beta != nullptr), "aff needs gamma and beta");
    const PeerDev no_peer = {};
    if (PeerCtx* pc = peer_active(stream)) {
        // synchronised BatchNorm through the peer-write exchange (peer.h): ONE launch, the exchange inside it
        I3D_CHECK_ARG(3L * feat <= PEER_PAYLOAD_WORDS, "BatchNorm too wide for the peer mailbox");
        PeerDev d;
        if (int rc = peer_next(pc, &d)) return rc;
        hipLaunchKernelGGL(bn_finalize_partials_kernel<true>, dim3(cdiv(feat, FIN_COLS)), dim3(256), 0, (hipStream_t)stream, partial,
                           n_tiles, feat, eps, momentum, gamma, beta, mean, invstd, running_mean, running_var,
                           num_batches_tracked, aff, (float*)nullptr, d);
        I3D_CHECK_LAUNCH();
        return I3D_OK;
    }
    if (const I3dCollectives* coll = collectives()) {
        // synchronised BatchNorm (comm.hip): local tiles -> one {sum, M2, count} triple -> all-gather on this stream -> the
        // same exact merge over the ranks' triples (the parallel-axis theorem does not care whose tiles they are)
        I3D_CHECK_ARG(coll->scratch_bytes >= (long)(1 + coll->world) * 3 * feat * 4, "collective scratch too small");
        float* send = (float*)coll->scratch;
        float* recv = send + 3L * feat;
        hipLaunchKernelGGL(bn_finalize_partials_kernel<false>, dim3(cdiv(feat, FIN_COLS)), dim3(256), 0, (hipStream_t)stream, partial,
                           n_tiles, feat, eps, momentum, gamma, beta, mean, invstd, (float*)nullptr, (float*)nullptr,
                           (long long*)nullptr, (float*)nullptr, send, no_peer);
        I3D_CHECK_LAUNCH();
        const int rc = coll->all_gather_f32(coll->user, send, recv, 3L * feat, stream);
        if (rc != I3D_OK) return rc;
        partial = recv;
        n_tiles = coll->world;
    }
    hipLaunchKernelGGL(bn_finalize_partials_kernel<false>, dim3(cdiv(feat, FIN_COLS)), dim3(256), 0, (hipStream_t)stream, partial,
                       n_tiles, feat, eps, momentum, gamma, beta, mean, invstd, running_mean, running_var,
                       num_batches_tracked, aff, (float*)nullptr, no_peer);
    I3D_CHECK_LAUNCH();
    return I3D_OK;
}

extern "C" int i3d_edge_stats_rows_per_tile(int feat) {
    if (feat <= 0) return 0;
    return ecs_tiling(feat, feat % 4 == 0 ? 4 : 1).rows_per_tile;
}

extern "C" int i3d_edge_combine_act_stats(const float* P, int ldp, const float* Q, const int* q_code, const float* bias,
                                          const int* src_s, const int* dst_s, int num_edges, int feat, int act, float* x,
                                          float* partial, void* stream) {
    I3D_CHECK_ARG(num_edges > 0 && feat > 0 && ldp >= 2 * feat, "bad shape");
    I3D_CHECK_ARG(act == I3D_ACT_NONE || act == I3D_ACT_RELU || act == I3D_ACT_LEAKY_RELU,
                  "only activations whose derivative follows from the output (none, ReLU, LeakyReLU)");
    const bool v4 = feat % 4 == 0;      // i3d_edge_stats_rows_per_tile(feat) must describe the tiling used here
    I3D_CHECK_ARG(!v4 || (ldp % 4 == 0 && ((((uintptr_t)P | (uintptr_t)Q | (uintptr_t)x | (uintptr_t)bias) & 15) == 0)),
                  "feat % 4 == 0 needs 16-byte aligned operands");
    const EcsTiling tl = ecs_tiling(feat, v4 ? 4 : 1);
    dim3 grid(cdiv(num_edges, tl.rows_per_tile), tl.ncolblk);
    if (v4)
        hipLaunchKernelGGL(edge_combine_act_stats_kernel<4>, grid, dim3(256), 0, (hipStream_t)stream, P, ldp, Q, q_code, bias,
                           src_s, dst_s, num_edges, feat, act, tl, x, partial);
    else
        hipLaunchKernelGGL(edge_combine_act_stats_kernel<1>, grid, dim3(256), 0, (hipStream_t)stream, P, ldp, Q, q_code, bias,
                           src_s, dst_s, num_edges, feat, act, tl, x, partial);
    I3D_CHECK_LAUNCH();
    return I3D_OK;
}
