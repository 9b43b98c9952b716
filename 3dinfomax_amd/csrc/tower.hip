// One PNALayer of the tower variant (reference models/pna_original.py:264-319: `towers` PNATowers + mixing network) from ONE C
// call per direction, on the stacked form of its towers (3dinfomax_amd/pna_original.py: _TowerStacks; csrc/pack.hip):
//   forward   P = h [Wp_s | Wp_d]^T,  Q = e Wp_e^T,  msg = P[src] + P[dst] + Q + bp          (pretrans Linear of every tower, :216-222, 246)
//             agg = aggregators x scalers over each node's in-edges of msg                    (:224-236, 249)
//             lin = [h | agg] Wq^T + bq,  y = BatchNorm(lin)                                  (posttrans + BatchNorm, :250-253)
//             y *= snorm_n                                                                    (graph norm, :256-258)
//             out = LeakyReLU(y Wm^T + bm) (+ h)                                              (mixing network + residual, :308-314)
//   backward  the same blocks in reverse; weight gradients through the split-K scratch (deterministic).
// Sequencing only - every kernel is one the per-block Python path launches (same kernels, same operands), so that hidden
// sizes that are not multiples of 4 (the yml's 90) take the kernels' unaligned variants exactly as they do there.  It exists
// because that path costs the host ~40 us per block and direction: the variant was 100 % host-bound.
#include "common.h"

#include <cstdlib>

using namespace i3d;

#define TRY(call)                 \
    do {                          \
        int rc_ = (call);         \
        if (rc_ != I3D_OK) return rc_; \
    } while (0)

static inline long al4(long n) { return (n + 3) & ~3L; }

// floats of `saved` (written by the forward pass, read by the backward pass): msg | agg | lin | mean | invstd | xs | mixpre
extern "C" long i3d_tower_layer_saved_floats(const I3dTowerLayerArgs* a) {
    if (a == nullptr) return 0;
    const long N = a->num_nodes, E = a->num_edges, Mp = a->f_msg, Mq = a->f_out, B = (long)a->n_aggregators * a->n_scalers;
    const long wd = a->n_deg_groups > 0 ? al4((long)a->n_deg_groups * Mq * a->n_aggregators * Mp) : 0;
    return al4(E * Mp) + al4(N * B * Mp) + al4(N * Mq) + 2 * al4(Mq) + 2 * al4(N * Mq) + wd;
}

// floats of `scratch`: forward P | Q | y;  backward g_mixpre | g_y | g_lin | g_agg | g_msg | gP
extern "C" long i3d_tower_layer_scratch_floats(const I3dTowerLayerArgs* a) {
    if (a == nullptr) return 0;
    const long N = a->num_nodes, E = a->num_edges, Mp = a->f_msg, Mq = a->f_out, B = (long)a->n_aggregators * a->n_scalers;
    const long fwd = al4(N * 2 * Mp) + al4(E * Mp) + al4(N * Mq);
    const long wd = a->n_deg_groups > 0 ? al4((long)a->n_deg_groups * Mq * a->n_aggregators * Mp) : 0;
    const long bwd = 3 * al4(N * Mq) + al4(N * B * Mp) + al4(E * Mp) + al4(N * 2 * Mp) + wd;
    return fwd > bwd ? fwd : bwd;
}

namespace {

struct Saved {
    float *msg, *agg, *lin, *mean, *invstd, *xs, *mixpre, *WD;
};

Saved saved_of(const I3dTowerLayerArgs* a) {
    const long N = a->num_nodes, E = a->num_edges, Mp = a->f_msg, Mq = a->f_out, B = (long)a->n_aggregators * a->n_scalers;
    Saved s;
    float* p = a->saved;
    s.msg = p; p += al4(E * Mp);
    s.agg = p; p += al4(N * B * Mp);
    s.lin = p; p += al4(N * Mq);
    s.mean = p; p += al4(Mq);
    s.invstd = p; p += al4(Mq);
    s.xs = p; p += al4(N * Mq);
    s.mixpre = p; p += al4(N * Mq);
    s.WD = p;
    return s;
}

bool args_ok(const I3dTowerLayerArgs* a) {
    return a != nullptr && a->num_nodes > 0 && a->num_edges > 0 && a->f_in > 0 && a->f_msg > 0 && a->f_out > 0 && a->f_edge >= 0 &&
           a->h != nullptr && a->Wp != nullptr && a->Wq != nullptr && a->Wm != nullptr && a->saved != nullptr && a->scratch != nullptr &&
           a->workspace != nullptr && a->n_aggregators >= 1 && a->n_aggregators <= 8 && a->n_scalers >= 1 && a->n_scalers <= 4 &&
           (a->f_edge == 0 || a->e != nullptr) && a->ldp >= 2 * a->f_in + a->f_edge && a->ldq >= a->f_in + a->n_aggregators * a->n_scalers * a->f_msg &&
           (a->n_towers <= 1 || (a->n_towers <= 32 && a->f_msg % (4 * a->n_towers) == 0 && a->f_out % (4 * a->n_towers) == 0)) &&
           (a->n_deg_groups <= 0 || (a->n_deg_groups <= 32 && a->n_deg_groups * a->n_scalers <= 128 &&
                                     a->deg_rows != nullptr && a->deg_tile_group != nullptr && a->m_padded % 64 == 0));
}

// the scalers folded into per-degree weights (I3dTowerLayerArgs::n_deg_groups)
inline bool scalers_folded(const I3dTowerLayerArgs* a) { return a->n_deg_groups > 0; }
const int kIdentityScaler[1] = {I3D_SCALE_IDENTITY};

// the posttrans products on the aggregation as `n_towers` diagonal blocks (I3dTowerLayerArgs::n_towers)
inline bool tower_major(const I3dTowerLayerArgs* a) { return a->n_towers > 1; }

}  // namespace

extern "C" int i3d_tower_layer_fwd(const I3dTowerLayerArgs* a, void* stream) {
    I3D_CHECK_ARG(args_ok(a) && a->out != nullptr, "bad arguments");
    const int N = a->num_nodes, E = a->num_edges, D = a->f_in, Fe = a->f_edge, Mp = a->f_msg, Mq = a->f_out;
    const int B = a->n_aggregators * a->n_scalers;
    const Saved s = saved_of(a);
    float* P = a->scratch;
    float* Q = P + al4((long)N * 2 * Mp);
    float* y = Q + al4((long)E * Mp);
    // pretrans of every tower: node-level products, then gather-combine (the [E, 2 D + Fe] concatenation never exists)
    // P = h [W_src | W_dst]^T as ONE product: row n >= Mp of the operand is row n - Mp of Wp, D columns further (i3d_gemm_f32_blocks)
    const long wp_delta = (long)D - (long)Mp * a->ldp, wp_view = (long)(Mp - 1) * a->ldp + 2L * D;
    TRY(i3d_gemm_f32_blocks(0, 1, N, 2 * Mp, D, a->h, D, a->Wp, a->ldp, Mp, wp_delta, wp_view, P, 2 * Mp, 0, 0, 0, nullptr, 0, stream));
    if (Fe > 0) TRY(i3d_gemm_f32(0, 1, E, Mp, Fe, a->e, Fe, a->Wp + 2 * D, a->ldp, Q, Mp, nullptr, 0, stream));
    TRY(i3d_edge_combine_fwd(P, 2 * Mp, Fe > 0 ? Q : nullptr, nullptr, a->bp, a->src_s, a->dst_s, E, Mp, s.msg, stream));
    const bool tm = tower_major(a);
    const int T = tm ? a->n_towers : 1, Ft = Mp / T, Fo = Mq / T;
    const bool fold = scalers_folded(a);
    const int AW = a->n_aggregators * Mp;      // width of the identity blocks
    const int nA = a->n_aggregators;
    if (fold && tm)      // identity blocks only, tower-major: [tower][aggregator][feature]
        TRY(i3d_pna_aggregate_fwd_towers(s.msg, a->in_ptr, N, Mp, Ft, a->aggregators, nA, kIdentityScaler, 1, 0, a->avg_d_log, s.agg,
                                         stream));
    else if (fold)
        TRY(i3d_pna_aggregate_fwd(s.msg, a->in_ptr, N, Mp, a->aggregators, a->n_aggregators, kIdentityScaler, 1, 0, a->avg_d_log, s.agg,
                                  stream));
    else if (tm)
        TRY(i3d_pna_aggregate_fwd_towers(s.msg, a->in_ptr, N, Mp, Ft, a->aggregators, a->n_aggregators, a->scalers, a->n_scalers, 1,
                                         a->avg_d_log, s.agg, stream));
    else
        TRY(i3d_pna_aggregate_fwd(s.msg, a->in_ptr, N, Mp, a->aggregators, a->n_aggregators, a->scalers, a->n_scalers, 1, a->avg_d_log,
                                  s.agg, stream));
    // posttrans on [h | agg] without the concatenation
    TRY(i3d_gemm_f32(0, 1, N, Mq, D, a->h, D, a->Wq, a->ldq, s.lin, Mq, a->bq, 0, stream));
    if (fold) {  // lin[r, :] += agg[r, :] W_D^T for the nodes r of in-degree D, W_D = sum_s c_s(D) W_s
        TRY(i3d_pna_combine_weights_fwd(a->Wq, a->ldq, D, Mq, AW, a->n_deg_groups, a->n_scalers, a->coef, s.WD, stream));
        if (tm)      // the diagonal blocks of W_D: tower t's rows of W_D on its own nA Ft aggregated columns
            TRY(i3d_gemm_f32_grouped_batched(1, a->m_padded, Fo, nA * Ft, s.agg, AW, (long)nA * Ft, N, a->deg_rows, a->deg_tile_group, s.WD,
                                             AW, (long)Mq * AW, (long)Fo * AW + (long)nA * Ft, s.lin, Mq, Fo, T, 1, stream));
        else
            TRY(i3d_gemm_f32_grouped(1, a->m_padded, Mq, AW, s.agg, AW, N, a->deg_rows, a->deg_tile_group, s.WD, AW, (long)Mq * AW, s.lin,
                                     Mq, 1, stream));
    } else if (tm)     // tower t: lin[:, t Fo ..] += agg[:, t B Ft ..] Wq[t Fo .., D + t B Ft ..]^T
        TRY(i3d_gemm_f32_batched(0, 1, N, Fo, B * Ft, s.agg, B * Mp, (long)B * Ft, a->Wq + D, a->ldq, (long)Fo * a->ldq + (long)B * Ft,
                                 s.lin, Mq, Fo, T, 1, nullptr, 0, stream));
    else
        TRY(i3d_gemm_f32(0, 1, N, Mq, B * Mp, s.agg, B * Mp, a->Wq + D, a->ldq, s.lin, Mq, nullptr, 1, stream));
    const float* yv = s.lin;
    if (a->gamma != nullptr) {
        if (a->training) {
            TRY(i3d_act_stats_fwd_counted(s.lin, N, Mq, I3D_ACT_NONE, s.lin, a->eps, a->momentum, s.mean, s.invstd, a->running_mean,
                                          a->running_var, nullptr, nullptr, a->workspace, stream));
            TRY(i3d_bn_apply_fwd(s.lin, N, Mq, s.mean, s.invstd, a->gamma, a->beta, I3D_ACT_NONE, nullptr, y, stream));
        } else {
            TRY(i3d_bn_eval_fwd(s.lin, N, Mq, a->running_mean, a->running_var, a->eps, a->gamma, a->beta, I3D_ACT_NONE, nullptr, y,
                                stream));
        }
        yv = y;
    }
    const float* xs = yv;
    if (a->snorm != nullptr) {
        TRY(i3d_row_scale(yv, a->snorm, N, Mq, s.xs, stream));
        xs = s.xs;
    } else if (yv != s.xs) {      // the mixing block's input is kept for its weight gradient
        if (hipMemcpyAsync(s.xs, yv, (size_t)N * Mq * 4, hipMemcpyDeviceToDevice, (hipStream_t)stream) != hipSuccess) {
            i3d::set_error("i3d_tower_layer_fwd: copy failed");
            return I3D_ERR_LAUNCH;
        }
        xs = s.xs;
    }
    // mixing network + LeakyReLU (+ residual)
    TRY(i3d_gemm_f32(0, 1, N, a->f_mix, Mq, xs, Mq, a->Wm, Mq, s.mixpre, a->f_mix, a->bm, 0, stream));
    TRY(i3d_act_fwd(s.mixpre, (long)N * a->f_mix, I3D_ACT_LEAKY_RELU, a->out, stream));
    if (a->residual) TRY(i3d_add_inplace(a->out, a->h, (long)N * a->f_mix, stream));
    return I3D_OK;
}

extern "C" int i3d_tower_layer_bwd(const I3dTowerLayerArgs* a, void* stream) {
    I3D_CHECK_ARG(args_ok(a) && a->grad_out != nullptr && a->grad_h != nullptr && a->grad_Wp != nullptr && a->grad_Wq != nullptr &&
                      a->grad_Wm != nullptr && a->grad_bp != nullptr && a->grad_bq != nullptr && a->grad_bm != nullptr &&
                      a->gemm_workspace != nullptr, "bad arguments");
    I3D_CHECK_ARG(a->gamma == nullptr || a->training, "backward through an eval-mode BatchNorm is not sequenced here");
    const int N = a->num_nodes, E = a->num_edges, D = a->f_in, Fe = a->f_edge, Mp = a->f_msg, Mq = a->f_out, Fm = a->f_mix;
    const int B = a->n_aggregators * a->n_scalers;
    const Saved s = saved_of(a);
    float* g_mixpre = a->scratch;
    float* g_y = g_mixpre + al4((long)N * Mq);
    float* g_lin = g_y + al4((long)N * Mq);
    float* g_agg = g_lin + al4((long)N * Mq);
    float* g_msg = g_agg + al4((long)N * B * Mp);
    float* gP = g_msg + al4((long)E * Mp);
    void* ws = a->gemm_workspace;
    const long wsb = a->gemm_workspace_bytes;
    // mixing network: out = LeakyReLU(mixpre) (+ h)
    TRY(i3d_act_bwd(a->grad_out, s.mixpre, (long)N * Fm, I3D_ACT_LEAKY_RELU, g_mixpre, stream));
    TRY(i3d_gemm_f32_ws(1, 0, Fm, Mq, N, g_mixpre, Fm, s.xs, Mq, a->grad_Wm, Mq, nullptr, 0, ws, wsb, stream));
    TRY(i3d_colsum(g_mixpre, nullptr, N, Fm, a->grad_bm, a->workspace, stream));
    TRY(i3d_gemm_f32(0, 0, N, Mq, Fm, g_mixpre, Fm, a->Wm, Mq, g_y, Mq, nullptr, 0, stream));
    // graph norm
    if (a->snorm != nullptr) TRY(i3d_row_scale(g_y, a->snorm, N, Mq, g_y, stream));
    // BatchNorm of the posttrans block (no activation in front of it)
    const float* gl = g_y;
    if (a->gamma != nullptr) {
        TRY(i3d_bn_bwd(g_y, s.lin, nullptr, N, Mq, I3D_ACT_NONE, I3D_ACT_NONE, s.mean, s.invstd, a->gamma, a->beta, a->grad_gamma,
                       a->grad_beta, g_lin, nullptr, nullptr, nullptr, N, a->workspace, stream));
        gl = g_lin;
    }
    // posttrans Linear on [h | agg]
    const bool tm = tower_major(a);
    const int T = tm ? a->n_towers : 1, Ft = Mp / T, Fo = Mq / T;
    const bool fold = scalers_folded(a);
    const int AW = a->n_aggregators * Mp;
    float* gWD = gP + al4((long)N * 2 * Mp);
    if (fold) {  // dW_D = dY_D^T agg_D over the rows of each in-degree group (one launch), folded back into the scaler blocks
        TRY(i3d_gemm_f32_rowsubset_multi(Mq, AW, a->n_deg_groups, a->group_start, a->group_count, gl, Mq, s.agg, AW, a->deg_rows, N, gWD,
                                         (long)Mq * AW, AW, 0, -1, 0, ws, wsb, stream));
        TRY(i3d_pna_combine_weights_bwd(gWD, a->ldgq, D, Mq, AW, a->n_deg_groups, a->n_scalers, a->coef, a->grad_Wq, stream));
    } else if (tm)     // (the blocks off the diagonal of grad_Wq are not written: nothing reads them)
        TRY(i3d_gemm_f32_batched(1, 0, Fo, B * Ft, N, gl, Mq, Fo, s.agg, B * Mp, (long)B * Ft, a->grad_Wq + D, a->ldgq,
                                 (long)Fo * a->ldgq + (long)B * Ft, T, 0, ws, wsb, stream));
    else
        TRY(i3d_gemm_f32_ws(1, 0, Mq, B * Mp, N, gl, Mq, s.agg, B * Mp, a->grad_Wq + D, a->ldgq, nullptr, 0, ws, wsb, stream));
    TRY(i3d_gemm_f32_ws(1, 0, Mq, D, N, gl, Mq, a->h, D, a->grad_Wq, a->ldgq, nullptr, 0, ws, wsb, stream));
    TRY(i3d_colsum(gl, nullptr, N, Mq, a->grad_bq, a->workspace, stream));
    if (fold && tm)
        TRY(i3d_gemm_f32_grouped_batched(0, a->m_padded, a->n_aggregators * Ft, Fo, gl, Mq, Fo, N, a->deg_rows, a->deg_tile_group, s.WD, AW,
                                         (long)Mq * AW, (long)Fo * AW + (long)a->n_aggregators * Ft, g_agg, AW, (long)a->n_aggregators * Ft,
                                         T, 0, stream));
    else if (fold)
        TRY(i3d_gemm_f32_grouped(0, a->m_padded, AW, Mq, gl, Mq, N, a->deg_rows, a->deg_tile_group, s.WD, AW, (long)Mq * AW, g_agg, AW, 0,
                                 stream));
    else if (tm)
        TRY(i3d_gemm_f32_batched(0, 0, N, B * Ft, Fo, gl, Mq, Fo, a->Wq + D, a->ldq, (long)Fo * a->ldq + (long)B * Ft, g_agg, B * Mp,
                                 (long)B * Ft, T, 0, nullptr, 0, stream));
    else
        TRY(i3d_gemm_f32(0, 0, N, B * Mp, Mq, gl, Mq, a->Wq + D, a->ldq, g_agg, B * Mp, nullptr, 0, stream));
    // dL/dh: the residual's share (grad_out itself), the posttrans block's, the edge block's below
    TRY(i3d_gemm_f32(0, 0, N, D, Mq, gl, Mq, a->Wq, a->ldq, a->grad_h, D, nullptr, 0, stream));
    if (a->residual) TRY(i3d_add_inplace(a->grad_h, a->grad_out, (long)N * D, stream));
    // aggregation
    if (fold && tm)
        TRY(i3d_pna_aggregate_bwd_towers(g_agg, s.msg, a->in_ptr, N, Mp, Ft, a->aggregators, a->n_aggregators, kIdentityScaler, 1, 0,
                                         a->avg_d_log, g_msg, stream));
    else if (fold)
        TRY(i3d_pna_aggregate_bwd(g_agg, s.msg, a->in_ptr, N, Mp, a->aggregators, a->n_aggregators, kIdentityScaler, 1, 0, a->avg_d_log,
                                  g_msg, stream));
    else if (tm)
        TRY(i3d_pna_aggregate_bwd_towers(g_agg, s.msg, a->in_ptr, N, Mp, Ft, a->aggregators, a->n_aggregators, a->scalers,
                                         a->n_scalers, 1, a->avg_d_log, g_msg, stream));
    else
        TRY(i3d_pna_aggregate_bwd(g_agg, s.msg, a->in_ptr, N, Mp, a->aggregators, a->n_aggregators, a->scalers, a->n_scalers, 1,
                                  a->avg_d_log, g_msg, stream));
    // pretrans: msg[j] = P[src_j, :Mp] + P[dst_j, Mp:] + Q[j] + bp
    TRY(i3d_segment_sum(g_msg, Mp, a->out_ptr, a->out_epos, N, Mp, 0, gP, 2 * Mp, stream));
    TRY(i3d_segment_sum(g_msg, Mp, a->in_ptr, nullptr, N, Mp, 0, gP + Mp, 2 * Mp, stream));
    if (Fe > 0) {
        TRY(i3d_gemm_f32_ws(1, 0, Mp, Fe, E, g_msg, Mp, a->e, Fe, a->grad_Wp + 2 * D, a->ldgp, nullptr, 0, ws, wsb, stream));
        if (a->grad_e != nullptr)
            TRY(i3d_gemm_f32(0, 0, E, Fe, Mp, g_msg, Mp, a->Wp + 2 * D, a->ldp, a->grad_e, Fe, nullptr, a->grad_e_accumulate, stream));
    }
    // d[W_src | W_dst] = gP^T h as ONE product whose rows >= Mp land D columns further in rows - Mp of grad_Wp
    TRY(i3d_gemm_f32_blocks(1, 0, 2 * Mp, D, N, gP, 2 * Mp, a->h, D, 0, 0, 0, a->grad_Wp, a->ldgp, Mp, (long)D - (long)Mp * a->ldgp, 0, ws,
                            wsb, stream));
    TRY(i3d_colsum(g_msg, nullptr, E, Mp, a->grad_bp, a->workspace, stream));
    // dh += gP [W_src; W_dst] as ONE product (K = 2 Mp)
    TRY(i3d_gemm_f32_blocks(0, 0, N, D, 2 * Mp, gP, 2 * Mp, a->Wp, a->ldp, Mp, (long)D - (long)Mp * a->ldp,
                            (long)(Mp - 1) * a->ldp + 2L * D, a->grad_h, D, 0, 0, 1, nullptr, 0, stream));
    return I3D_OK;
}
