// Host-side composites: one C call enqueues the whole kernel sequence of an FCLayer-shaped block.
//
// The step is ~400 small launches; with one Python->C transition per kernel the host needed ~5 ms to enqueue a
// 4.6 ms step.  These entry points keep the kernels unchanged and move the *sequencing* of an
// "input operator -> Linear -> activation -> BatchNorm(train) -> post-activation (+ residual)" block and of its
// backward into C++ (reference models/base_layers.py:100-111 with the three input operators of layers.py: plain,
// edge gather-combine, degree-grouped concat).  Training mode with local batch statistics only; eval mode,
// synchronised BN and BN-less layers keep using the per-kernel entry points.
#include "common.h"

using namespace i3d;

#define TRY(call)                 \
    do {                          \
        int rc_ = (call);         \
        if (rc_ != I3D_OK) return rc_; \
    } while (0)

// ---- weight gradients next to the data-gradient chain ---------------------------------------------------------------
// The backward pass of a block is a chain (BatchNorm backward -> data gradient -> next block) of small kernels that each
// occupy a fraction of the chip, plus weight-gradient GEMMs (dW = dY^T X and their split-K reductions, ~20 % of the step's
// kernel time) that nothing downstream in the chain reads.  The PNA layer composite enqueues those on a second HIP stream
// of its own (one per caller stream, created on first use): fork = an event recorded on the caller's stream that the side
// stream waits for, join = the caller's stream waits for the side stream before the layer returns, so every tensor the
// caller sees afterwards is complete and the caching allocator's stream-ordered reuse stays valid.  All weight gradients
// of one caller stream share that stream's split-K scratch: they are serialised on the one side stream.  Same kernels,
// same arguments, same order per stream: results are bit-identical.  A fork or join costs the host ~7 us
// (tools/probes/forkjoin_probe.hip), so a layer forks twice and joins once, and the stand-alone block entry points (heads,
// the 3D network - whose stream has slack anyway) stay on one stream.  I3D_WGRAD_STREAM=0: off.
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <utility>

namespace {

struct Aux {
    hipStream_t s = nullptr;
    hipEvent_t fork = nullptr, join = nullptr;
};

Aux* aux_for(hipStream_t main) {
    static const bool on = [] { const char* e = getenv("I3D_WGRAD_STREAM"); return e == nullptr || e[0] != '0'; }();
    if (!on) return nullptr;
    static std::mutex mu;
    static std::map<std::pair<int, hipStream_t>, Aux*> table;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    std::lock_guard<std::mutex> lock(mu);
    auto it = table.find({dev, main});
    if (it != table.end()) return it->second;
    Aux* a = new Aux();
    // I3D_WGRAD_PRIORITY=low: the side stream at the lowest priority the device offers - the kernels of the backward CHAIN (the
    // critical path: the main stream is busy back to back through the whole step) get the CUs first, the weight gradients
    // fill in
    constexpr bool low = false;      // (a low-priority side stream was measured and lost: docs/history)
    int lo = 0, hi = 0;
    hipError_t made = hipErrorUnknown;
    if (low && hipDeviceGetStreamPriorityRange(&lo, &hi) == hipSuccess)
        made = hipStreamCreateWithPriority(&a->s, hipStreamNonBlocking, lo);
    if (made != hipSuccess) made = hipStreamCreateWithFlags(&a->s, hipStreamNonBlocking);
    // fork / join events order two streams of ONE device: released to the device, not to the system (the default scope of an
    // event record is a system-scope release - a cache write-back the host and the peers would need, the other stream does
    // not; 5 records per step sit on the chain: 2.091 -> 2.074 ms).
    constexpr bool dev_scope = true;
    const unsigned evf = hipEventDisableTiming | (dev_scope ? hipEventDisableSystemFence : 0);
    if (made != hipSuccess ||
        hipEventCreateWithFlags(&a->fork, evf) != hipSuccess ||
        hipEventCreateWithFlags(&a->join, evf) != hipSuccess) {
        delete a;
        a = nullptr;          // remembered: no retry on every call
    }
    table[{dev, main}] = a;
    return a;
}

// the stream for the weight gradients that are enqueued next: the side stream, ordered after everything the caller's
// stream holds now - or the caller's stream itself when there is no side stream
void* fork_wgrad(Aux* x, void* main) {
    if (x == nullptr) return main;
    if (hipEventRecord(x->fork, (hipStream_t)main) != hipSuccess || hipStreamWaitEvent(x->s, x->fork, 0) != hipSuccess) return main;
    return (void*)x->s;
}

int join_wgrad(Aux* x, void* main) {
    if (x == nullptr) return I3D_OK;
    if (hipEventRecord(x->join, x->s) != hipSuccess || hipStreamWaitEvent((hipStream_t)main, x->join, 0) != hipSuccess) {
        i3d::set_error("joining the weight-gradient stream failed");
        return I3D_ERR_LAUNCH;
    }
    return I3D_OK;
}

}  // namespace

// the row-panel product (panel.hip) takes a packed weight: fp32 mode with split products, K % 8 == 0, N % 4 == 0 - and pays where the
// reduction is short (K <= 256: 34 -> 21 us at [E,200]x[200,200], 31 -> 22 us at [N,200]x[600,200]^T; level at K = 600 / 800:
// profiles/r06_panel_bench.txt)
static bool panel_ok(const void* packed, int K, int N, int k_max = 256) {
    static const bool on = [] { const char* e = getenv("I3D_PANEL_GEMM"); return e == nullptr || e[0] != '0'; }();      // (A/B)
    return on && packed != nullptr && i3d_get_matmul_precision() == 0 && i3d_get_fp32_products() == 1 && K % 8 == 0 && K <= k_max && N % 4 == 0;
}

static int tail_fwd(const I3dBnTail* t, int rows, int f_out, float* pre, float* xact, const float* residual, float* y,
                    void* stream) {
    // pre holds the Linear output.  xact == pre: activation in place (ReLU/none); else pre is kept for act'
    TRY(i3d_act_stats_fwd_counted(pre, rows, f_out, t->act, xact, t->eps, t->momentum, t->mean, t->invstd, t->running_mean,
                                  t->running_var, nullptr, t->num_batches_tracked, t->workspace, stream));
    return i3d_bn_apply_fwd(xact, rows, f_out, t->mean, t->invstd, t->gamma, t->beta, t->post_act, residual, y, stream);
}

// BN/activation backward; grad_bias = column sums of grad_pre from the same pass
static int tail_bwd(const I3dBnTail* t, int rows, int f_out, const float* grad_y, const float* xact, const float* pre,
                    float* grad_gamma, float* grad_beta, float* grad_pre, float* grad_bias, void* stream) {
    return i3d_bn_bwd_deferred_bias(grad_y, xact, pre, rows, f_out, t->act, t->post_act, t->mean, t->invstd, t->gamma, t->beta,
                                    grad_gamma, grad_beta, grad_pre, grad_bias, nullptr, nullptr, rows, t->workspace,
                                    t->bias_partial, stream);
}

// the deferred half of tail_bwd: first thing on the stream the block's weight gradients run on
static int bias_final(const I3dBnTail* t, int rows, int f_out, float* grad_bias, void* wst) {
    if (t->bias_partial == nullptr || grad_bias == nullptr) return I3D_OK;
    return i3d_bn_bias_finalize(t->bias_partial, rows, f_out, grad_bias, wst);
}

// ---- plain FC ------------------------------------------------------------------------------------------
extern "C" int i3d_fc_bn_fwd(const I3dFcArgs* a, void* stream) {
    I3D_CHECK_ARG(a != nullptr && a->rows > 0, "bad arguments");
    float* lin = a->pre_keep ? a->pre_keep : a->xact;
    TRY(i3d_gemm_f32(0, 1, a->rows, a->f_out, a->f_in, a->x, a->f_in, a->W, a->ldw, lin, a->f_out, a->bias, 0, stream));
    return tail_fwd(&a->tail, a->rows, a->f_out, lin, a->xact, a->residual, a->y, stream);
}

// The backward of a block in two parts: the chain (BatchNorm backward, data gradient: what the next block's backward
// waits for) on `stream`, and the weight gradients, which only need the block's grad_pre, on `wst` - the same stream for
// the stand-alone entry points, the side stream for a PNA layer (which issues the weight gradients of several blocks behind
// ONE fork: a fork or join costs the host ~7 us, tools/probes/forkjoin_probe.hip).
// the bias gradient of a block next to its weight gradients (on their stream)
// A block with an activation in front of its BatchNorm: the bias gradient is the column sum of grad_pre.  Where the one-launch
// BatchNorm backward takes the shape (bn.hip), the chain runs it WITHOUT those sums (reduction + data gradient + their column sums
// were two launches: 10 + 11.5 us at the head's 512 rows) and the sums are taken next to the weight gradients, from grad_pre.
static bool bias_aside(const I3dFcArgs* a, int xact_bf16 = 0) {
    return !xact_bf16 && a->grad_bias != nullptr && a->tail.bias_partial != nullptr && a->tail.act != I3D_ACT_NONE && relu_class(a->tail.act) &&
           relu_class(a->tail.post_act) && a->pre_keep == nullptr && i3d_bn_bwd_one_launch_supported(a->rows, a->f_out) != 0;
}

static int bias_final_fc(const I3dFcArgs* a, void* wst, int xact_bf16 = 0) {
    if (bias_aside(a, xact_bf16)) return i3d_colsum_strided(a->grad_pre, a->f_out, a->rows, a->f_out, a->grad_bias, a->tail.bias_partial, wst);
    return bias_final(&a->tail, a->rows, a->f_out, a->grad_bias, wst);
}

static int fc_bn_bwd_chain(const I3dFcArgs* a, void* stream, int xact_bf16 = 0) {
    if (xact_bf16)      // the block's activation (a PNA layer's messages) is stored as bf16
        TRY(i3d_bn_bwd_x_bf16(a->grad_y, a->xact, a->rows, a->f_out, a->tail.act, a->tail.post_act, a->tail.mean, a->tail.invstd,
                              a->tail.gamma, a->tail.beta, a->grad_gamma, a->grad_beta, a->grad_pre, a->grad_bias, a->tail.workspace,
                              a->tail.bias_partial, stream));
    else
        TRY(tail_bwd(&a->tail, a->rows, a->f_out, a->grad_y, a->xact, a->pre_keep, a->grad_gamma, a->grad_beta, a->grad_pre,
                     bias_aside(a, xact_bf16) ? nullptr : a->grad_bias, stream));
    if (a->grad_x != nullptr) {
        if (panel_ok(a->W_dgrad_panel, a->f_out, a->f_in) && (((uintptr_t)a->grad_pre | (uintptr_t)a->grad_x) & 15) == 0)
            TRY(i3d_panel_gemm(a->rows, a->f_in, a->f_out, a->grad_pre, a->f_out, a->W_dgrad_panel, a->grad_x, a->f_in, nullptr, 0, stream));
        else
            TRY(i3d_gemm_f32(0, 0, a->rows, a->f_in, a->f_out, a->grad_pre, a->f_out, a->W, a->ldw, a->grad_x, a->f_in, nullptr,
                             0, stream));
    }
    return I3D_OK;
}

// x_aff != null (fused BatchNorm): a->x is the RAW activation of the block in front, its BatchNorm output
// (x - mean) * scale + shift was never materialised; the product is corrected in the slice reduction (gemm.hip)
static int fc_bn_bwd_wgrad(const I3dFcArgs* a, void* wst, const float* x_aff = nullptr) {
    TRY(bias_final_fc(a, wst));
    if (x_aff != nullptr)
        return i3d_gemm_f32_wgrad_bn(a->f_out, a->f_in, a->rows, a->grad_pre, a->f_out, a->x, a->f_in, a->grad_W, a->ldw,
                                     a->grad_bias, x_aff, a->tail.gemm_workspace, a->tail.gemm_workspace_bytes, wst);
    return i3d_gemm_f32_ws(1, 0, a->f_out, a->f_in, a->rows, a->grad_pre, a->f_out, a->x, a->f_in, a->grad_W, a->ldw, nullptr, 0,
                           a->tail.gemm_workspace, a->tail.gemm_workspace_bytes, wst);
}

extern "C" int i3d_fc_bn_bwd(const I3dFcArgs* a, void* stream) {
    I3D_CHECK_ARG(a != nullptr && a->rows > 0, "bad arguments");
    TRY(fc_bn_bwd_chain(a, stream));
    return fc_bn_bwd_wgrad(a, stream);
}

// the two halves on their own (the whole-model sequencer runs the head's chain on its stream and the head's weight gradients
// behind one fork of the weight-gradient stream: they are leaves, nothing of the chain waits for them)
extern "C" int i3d_fc_bn_bwd_chain(const I3dFcArgs* a, void* stream) {
    I3D_CHECK_ARG(a != nullptr && a->rows > 0, "bad arguments");
    return fc_bn_bwd_chain(a, stream);
}

extern "C" int i3d_fc_bn_bwd_wgrad(const I3dFcArgs* a, void* stream) {
    I3D_CHECK_ARG(a != nullptr && a->rows > 0, "bad arguments");
    return fc_bn_bwd_wgrad(a, stream);
}

// ---- edge FC: [h_src | h_dst | q] -> Linear as node-level P + gather-combine -------------------------
extern "C" int i3d_edge_fc_bn_fwd(const I3dEdgeFcArgs* a, void* stream) {
    I3D_CHECK_ARG(a != nullptr && a->num_edges > 0 && a->num_nodes > 0, "bad arguments");
    const int Fh = a->f_h, Fo = a->f_out;
    // P = h [W_s | W_d]^T: one GEMM over both column blocks of W (rows n >= Fo of the virtual [2Fo, Fh] operand are
    // rows n - Fo of W shifted by Fh columns)
    const long wdelta = (long)Fh - (long)Fo * a->ldw, wview = (long)(Fo - 1) * a->ldw + 2 * Fh;
    TRY(i3d_gemm_f32_blocks(0, 1, a->num_nodes, 2 * Fo, Fh, a->h, Fh, a->W, a->ldw, Fo, wdelta, wview, a->P, 2 * Fo, 0, 0, 0,
                            nullptr, 0, stream));
    if (a->q != nullptr)     // table mode: Q = table W_q^T has q_rows rows, the combine gathers row q_code[j]
        TRY(i3d_gemm_f32(0, 1, a->q_rows > 0 ? a->q_rows : a->num_edges, Fo, a->f_q, a->q, a->f_q, a->W + 2 * Fh, a->ldw, a->Q,
                         Fo, nullptr, 0, stream));
    float* lin = a->pre_keep ? a->pre_keep : a->xact;
    TRY(i3d_edge_combine_fwd(a->P, 2 * Fo, a->q ? a->Q : nullptr, a->q_rows > 0 ? a->q_code : nullptr, a->bias, a->src_s,
                             a->dst_s, a->num_edges, Fo, lin, stream));
    return tail_fwd(&a->tail, a->num_edges, Fo, lin, a->xact, nullptr, a->y, stream);
}

static int edge_fc_bn_bwd_tail(const I3dEdgeFcArgs* a, void* stream) {
    return tail_bwd(&a->tail, a->num_edges, a->f_out, a->grad_y, a->xact, a->pre_keep, a->grad_gamma, a->grad_beta, a->grad_pre,
                    a->grad_bias, stream);
}

// d P[src] (out-edges through the source index), d P[dst] (in-edges are contiguous): segmented sums, no atomics
static int edge_fc_bn_bwd_sums(const I3dEdgeFcArgs* a, void* stream) {
    const int Fo = a->f_out, N = a->num_nodes;
    return i3d_segment_sum_pair(a->grad_pre, Fo, a->out_ptr, a->out_epos, a->grad_P, a->in_ptr, nullptr, a->grad_P + Fo, N, Fo, 2 * Fo,
                                stream);
}

static int edge_fc_bn_bwd_chain(const I3dEdgeFcArgs* a, void* stream) {
    TRY(edge_fc_bn_bwd_tail(a, stream));
    return edge_fc_bn_bwd_sums(a, stream);
}

// dh = dP [W_s; W_d], the input of the next block's backward; `into` != null: added onto that buffer instead (a PNA layer
// sums the edge block's dh into the posttrans block's: the GEMM epilogue does the addition, no separate kernel)
static int edge_fc_bn_bwd_dgrad(const I3dEdgeFcArgs* a, void* stream, float* into = nullptr) {
    const int Fh = a->f_h, Fo = a->f_out, N = a->num_nodes;
    const long wdelta = (long)Fh - (long)Fo * a->ldw, wview = (long)(Fo - 1) * a->ldw + 2 * Fh;
    return i3d_gemm_f32_blocks(0, 0, N, Fh, 2 * Fo, a->grad_P, 2 * Fo, a->W, a->ldw, Fo, wdelta, wview,
                               into != nullptr ? into : a->grad_h, Fh, 0, 0, into != nullptr ? 1 : 0, nullptr, 0, stream);
}

// d[W_s | W_d] = dP^T h (rows >= Fo of the [2Fo, Fh] result land in the second column block); every column block of dW is
// written exactly once: no zero-fill (the split-K slices go through the scratch).  Needs dP.
static int edge_fc_bn_bwd_wgrad_p(const I3dEdgeFcArgs* a, void* wst) {
    const int Fh = a->f_h, Fo = a->f_out, N = a->num_nodes;
    const long wdelta = (long)Fh - (long)Fo * a->ldw;
    return i3d_gemm_f32_blocks(1, 0, 2 * Fo, Fh, N, a->grad_P, 2 * Fo, a->h, Fh, 0, 0, 0, a->grad_W, a->ldw, Fo, wdelta, 0,
                               a->tail.gemm_workspace, a->tail.gemm_workspace_bytes, wst);
}

// the W_q column block and, in table mode, everything behind dQ (only the caller reads grad_q, after the join).  Needs
// grad_pre only.
static int edge_fc_bn_bwd_wgrad_q(const I3dEdgeFcArgs* a, void* wst) {
    const int Fh = a->f_h, Fo = a->f_out, E = a->num_edges;
    TRY(bias_final(&a->tail, E, Fo, a->grad_bias, wst));
    void* ws = a->tail.gemm_workspace;
    const long wsb = a->tail.gemm_workspace_bytes;
    if (a->q != nullptr && a->q_rows > 0) {
        // table mode: dQ[v] = sum of dpre over the edges of category v (one-hot^T dpre), then two [V, .] products
        const int V = a->q_rows;
        TRY(i3d_gemm_f32_ws(1, 0, a->v_pad, Fo, E, a->onehot, a->v_pad, a->grad_pre, Fo, a->grad_Q, Fo, nullptr, 0, ws, wsb, wst));
        TRY(i3d_gemm_f32_ws(1, 0, Fo, a->f_q, V, a->grad_Q, Fo, a->q, a->f_q, a->grad_W + 2 * Fh, a->ldw, nullptr, 0, ws, wsb, wst));
        if (a->grad_q != nullptr)
            TRY(i3d_gemm_f32(0, 0, V, a->f_q, Fo, a->grad_Q, Fo, a->W + 2 * Fh, a->ldw, a->grad_q, a->f_q, nullptr,
                             a->grad_q_accumulate, wst));
    } else if (a->q != nullptr) {
        TRY(i3d_gemm_f32_ws(1, 0, Fo, a->f_q, E, a->grad_pre, Fo, a->q, a->f_q, a->grad_W + 2 * Fh, a->ldw, nullptr, 0, ws, wsb, wst));
        if (a->grad_q != nullptr)
            TRY(i3d_gemm_f32(0, 0, E, a->f_q, Fo, a->grad_pre, Fo, a->W + 2 * Fh, a->ldw, a->grad_q, a->f_q, nullptr, 0, wst));
    }
    return I3D_OK;
}

static int edge_fc_bn_bwd_wgrad(const I3dEdgeFcArgs* a, void* wst) {
    TRY(edge_fc_bn_bwd_wgrad_p(a, wst));
    return edge_fc_bn_bwd_wgrad_q(a, wst);
}

extern "C" int i3d_edge_fc_bn_bwd(const I3dEdgeFcArgs* a, void* stream) {
    I3D_CHECK_ARG(a != nullptr && a->num_edges > 0 && a->num_nodes > 0, "bad arguments");
    TRY(edge_fc_bn_bwd_chain(a, stream));
    TRY(edge_fc_bn_bwd_wgrad(a, stream));
    return edge_fc_bn_bwd_dgrad(a, stream);
}

// ---- degree-grouped concat FC: [h | scaler blocks of a] -> Linear with per-degree combined weights -----------
extern "C" int i3d_grouped_fc_bn_fwd(const I3dGroupedFcArgs* a, void* stream) {
    I3D_CHECK_ARG(a != nullptr && a->num_nodes > 0 && a->n_groups > 0, "bad arguments");
    const int Fh = a->f_h, Fo = a->f_out, A = a->agg_width, N = a->num_nodes;
    float* lin = a->pre_keep ? a->pre_keep : a->xact;
    TRY(i3d_gemm_f32(0, 1, N, Fo, Fh, a->h, Fh, a->W, a->ldw, lin, Fo, a->bias, 0, stream));
    TRY(i3d_pna_combine_weights_fwd(a->W, a->ldw, Fh, Fo, A, a->n_groups, a->n_scalers, a->coef, a->WD, stream));
    TRY(i3d_gemm_f32_grouped(1, a->m_padded, Fo, A, a->agg, A, N, a->deg_rows, a->deg_tile_group, a->WD, A, (long)Fo * A, lin,
                             Fo, 1, stream));
    return tail_fwd(&a->tail, N, Fo, lin, a->xact, a->residual, a->y, stream);
}

// grad_h_accumulate: grad_h already holds a gradient w.r.t. h (the residual's: grad_h aliases grad_y), add to it
static int grouped_fc_bn_bwd_chain(const I3dGroupedFcArgs* a, void* stream, int grad_h_accumulate = 0) {
    const int Fh = a->f_h, Fo = a->f_out, A = a->agg_width, N = a->num_nodes;
    TRY(tail_bwd(&a->tail, N, Fo, a->grad_y, a->xact, a->pre_keep, a->grad_gamma, a->grad_beta, a->grad_pre, a->grad_bias,
                 stream));
    TRY(i3d_gemm_f32(0, 0, N, Fh, Fo, a->grad_pre, Fo, a->W, a->ldw, a->grad_h, Fh, nullptr, grad_h_accumulate, stream));
    return i3d_gemm_f32_grouped(0, a->m_padded, A, Fo, a->grad_pre, Fo, N, a->deg_rows, a->deg_tile_group, a->WD, A,
                                (long)Fo * A, a->grad_agg, A, 0, stream);
}

static int grouped_fc_bn_bwd_wgrad(const I3dGroupedFcArgs* a, void* wst) {
    const int Fh = a->f_h, Fo = a->f_out, A = a->agg_width, N = a->num_nodes;
    TRY(bias_final(&a->tail, N, Fo, a->grad_bias, wst));
    TRY(i3d_gemm_f32_ws(1, 0, Fo, Fh, N, a->grad_pre, Fo, a->h, Fh, a->grad_W, a->ldw, nullptr, 0, a->tail.gemm_workspace,
                        a->tail.gemm_workspace_bytes, wst));
    // dW_D = dY_D^T a_D over the rows of each in-degree group, all groups in one launch
    TRY(i3d_gemm_f32_rowsubset_multi(Fo, A, a->n_groups, a->group_start, a->group_count, a->grad_pre, Fo, a->agg, A,
                                     a->deg_rows, N, a->grad_WD, (long)Fo * A, A, 0, -1, 0, a->tail.gemm_workspace,
                                     a->tail.gemm_workspace_bytes, wst));
    return i3d_pna_combine_weights_bwd(a->grad_WD, a->ldw, Fh, Fo, A, a->n_groups, a->n_scalers, a->coef, a->grad_W, wst);
}

extern "C" int i3d_grouped_fc_bn_bwd(const I3dGroupedFcArgs* a, void* stream) {
    I3D_CHECK_ARG(a != nullptr && a->num_nodes > 0 && a->n_groups > 0, "bad arguments");
    TRY(grouped_fc_bn_bwd_chain(a, stream));
    return grouped_fc_bn_bwd_wgrad(a, stream);
}

// the h-products of a layer merged into one GEMM per direction (I3dPnaLayerArgs.merge_h): fused form, one posttrans block
static bool merge_h_ok(const I3dPnaLayerArgs* a) {
    return a->merge_h && a->fused_bn && a->n_post_extra == 0 && a->Wcat != nullptr && a->bcat != nullptr && a->PL != nullptr &&
           a->post.f_h == a->edge.f_h && a->post.pre_keep == nullptr;
}

// Wcat / bcat of a merged layer, and the packed images the row-panel products read (Wcat for the forward h-product, the later pretrans
// blocks' weights for their data gradients): parameters only - once per forward pass, off the chain when the sequencer hoists it
static int pack_layer_weights(const I3dPnaLayerArgs* a, void* stream) {
    const I3dEdgeFcArgs* e = &a->edge;
    const I3dGroupedFcArgs* p = &a->post;
    TRY(i3d_pna_pack_h_weights(e->W, e->ldw, e->f_out, p->W, p->ldw, p->f_out, p->bias, e->f_h, a->Wcat, a->bcat, stream));
    I3dPanelPack w[8];          // ONE pack launch for the layer (16 launches of ~6 us per step as separate ones)
    int n = 0;
    const int WL = 2 * e->f_out + p->f_out;
    if (a->Wcat_panel != nullptr) w[n++] = I3dPanelPack{a->Wcat, e->f_h, WL, e->f_h, 1, a->Wcat_panel};
    if (a->Wcat_dgrad_panel != nullptr) w[n++] = I3dPanelPack{a->Wcat, e->f_h, e->f_h, WL, 0, a->Wcat_dgrad_panel};
    for (int i = 0; i < a->n_pre_extra && n + 2 <= 8; ++i) {
        const I3dFcArgs* c = &a->pre[i];
        if (c->W_dgrad_panel != nullptr) w[n++] = I3dPanelPack{c->W, c->ldw, c->f_in, c->f_out, 0, c->W_dgrad_panel};
        if (c->W_fwd_panel != nullptr) w[n++] = I3dPanelPack{c->W, c->ldw, c->f_out, c->f_in, 1, c->W_fwd_panel};
    }
    if (n > 0) TRY(i3d_panel_pack_multi(w, n, stream));
    return I3D_OK;
}

// the edge block's BatchNorm backward fused with the segmented sums behind it (bn.hip: i3d_bn_bwd_edge_sums): merged h-products
// (dP lands in the first two column blocks of DL), an activation whose derivative follows from the stored activation, 16-byte rows.
// I3D_EDGE_BWD_FUSED=0: BatchNorm backward + i3d_segment_sum_pair (the same bits).
// The LAST layer of a backward pass (wgrad_split: nothing of the chain runs next to its weight gradients any more): the chain's
// stream is idle behind its data gradient while the weight-gradient stream still has the layer's panels, their reduction and
// the bond-table products in front of the join - the two small launches of the bias gradient go to the chain's stream there
// (19 us off the tail of the step).
static bool edge_bias_on_chain(const I3dPnaLayerArgs* a) {
    return a->wgrad_split != 0;
}

static bool edge_bwd_fused_ok(const I3dPnaLayerArgs* a) {
    const I3dEdgeFcArgs* e = &a->edge;
    return merge_h_ok(a) && a->DL != nullptr && e->pre_keep == nullptr && relu_class(e->tail.act) && e->tail.post_act == I3D_ACT_NONE &&
           e->f_out % 4 == 0 && e->f_out <= 512 && a->post.f_out % 4 == 0 && e->out_ptr != nullptr && e->out_epos != nullptr &&
           a->edge_bias_partial != nullptr && e->grad_bias != nullptr;
}

// ---- all weight gradients of a PNA layer behind ONE fork, from ONE launch + one reduction (wgrad.hip) ---------------
// posttrans h-block | per-degree posttrans blocks folded into the scaler blocks | later pretrans blocks (BatchNorm fix-up in
// the fused form) | [W_s | W_d] of the edge block | dQ of the bond table; then the two [V, .] products behind dQ.  Returns
// 1 when it took the layer, 0 when the layer's shape is not covered (the caller issues the per-block launches), < 0: error.
// I3D_WGRAD_MULTI=0: off.
static int pna_layer_wgrad_multi(const I3dPnaLayerArgs* a, void* wst, bool dry_run, int part = 0) {
    static const bool on = [] { const char* e = getenv("I3D_WGRAD_MULTI"); return e == nullptr || e[0] != '0'; }();
    const I3dEdgeFcArgs* e = &a->edge;
    const I3dGroupedFcArgs* g = &a->post;
    // (bf16 matmul mode: the panel kernel's bf16 form)
    if (!on || a->n_post_extra != 0 || e->q == nullptr || e->q_rows <= 0) return 0;
    const int Fh = e->f_h, Fo = e->f_out, N = e->num_nodes, E = e->num_edges, A = g->agg_width;
    const bool merged = merge_h_ok(a) && a->DL != nullptr;
    const int WL = 2 * Fo + g->f_out;
    const float* dlin = merged ? a->DL + 2 * Fo : g->grad_pre;      // [N, f_out(post)]
    const int ld_dlin = merged ? WL : g->f_out;
    const float* dP = merged ? a->DL : e->grad_P;                    // [N, 2 Fo]
    const int ld_dP = merged ? WL : 2 * Fo;
    constexpr int MAX_PROBLEMS = 40, MAX_OUTPUTS = 8, MAX_GROUPS = 32, MAX_SCALERS = 4;     // the tables of wgrad.hip
    I3dWgradProblem pr[MAX_PROBLEMS];
    I3dWgradOutput out[MAX_OUTPUTS];
    float coef[MAX_GROUPS * MAX_SCALERS];
    // everything that is not a per-degree group: dW_h, the later pretrans blocks, [W_s | W_d], dQ; the group loop below
    // stops short of what they need, so neither table can be overrun whatever degree distribution the caller hands in
    const int fixed_problems = 3 + (a->n_pre_extra > 0 ? a->n_pre_extra : 0);
    if (g->n_scalers > MAX_SCALERS || g->n_scalers < 0 || fixed_problems >= MAX_PROBLEMS || 4 + a->n_pre_extra > MAX_OUTPUTS) return 0;
    const int max_group_problems = MAX_PROBLEMS - fixed_problems < MAX_GROUPS ? MAX_PROBLEMS - fixed_problems : MAX_GROUPS;
    std::memset(pr, 0, sizeof(pr));
    std::memset(out, 0, sizeof(out));
    int np = 0, no = 0;
    auto problem = [&](const float* Ap, int lda, int M, const float* Bp, int ldb, int Nn, long rows_total, const int* rows,
                       int k_begin, int k_count) {
        I3dWgradProblem& p = pr[np++];
        p.A = Ap; p.B = Bp; p.rows = rows; p.rows_total = rows_total; p.lda = lda; p.ldb = ldb; p.M = M; p.N = Nn;
        p.k_begin = k_begin; p.k_count = k_count;
    };
    // part 1: the posttrans products only (they need the posttrans chain's dlin only: the LAST layer of a backward pass
    // issues them early, next to the rest of its chain, so that only part 2 is left when the chain ends); part 2: the rest
    const bool do_post = part != 2, do_pre = part != 1;
    // posttrans: dW_h = dlin^T h
    if (do_post) {
        out[no].kind = I3D_WGRAD_PLAIN; out[no].n_groups = 1; out[no].first_problem = np; out[no].C = g->grad_W; out[no].ldc = g->ldw;
        ++no;
        problem(dlin, ld_dlin, g->f_out, g->h, Fh, Fh, N, nullptr, 0, N);
    }
    // posttrans: dW_s = sum_D c_s(D) dlin_D^T a_D over the in-degree groups with a non-zero coefficient
    if (do_post) {
        I3dWgradOutput& o = out[no];
        o.kind = I3D_WGRAD_COMBINE; o.first_problem = np; o.C = g->grad_W + Fh; o.ldc = g->ldw;
        o.n_scalers = g->n_scalers; o.scaler_stride = A; o.coef = coef;
        int ng = 0;
        for (int k = 0; k < g->n_groups; ++k) {
            bool any = false;
            for (int s = 0; s < g->n_scalers; ++s) any = any || g->coef[k * g->n_scalers + s] != 0.f;
            if (!any || g->group_count[k] == 0) continue;
            if (ng >= max_group_problems) return 0;                // more groups than the tables hold: per-block launches
            for (int s = 0; s < g->n_scalers; ++s) coef[ng * g->n_scalers + s] = g->coef[k * g->n_scalers + s];
            problem(dlin, ld_dlin, g->f_out, g->agg, A, A, N, g->deg_rows, g->group_start[k], g->group_count[k]);
            ++ng;
        }
        if (ng == 0) return 0;
        o.n_groups = ng;
        ++no;
    }
    // later pretrans blocks: dW = dpre^T BN(x) from the raw x (fused form) or dpre^T x
    for (int i = a->n_pre_extra - 1; i >= 0 && do_pre; --i) {
        const I3dFcArgs* c = &a->pre[i];
        I3dWgradOutput& o = out[no++];
        o.n_groups = 1; o.first_problem = np; o.C = c->grad_W; o.ldc = c->ldw;
        if (a->fused_bn) { o.kind = I3D_WGRAD_BN; o.aff = a->aff[i]; o.row = c->grad_bias; }
        else o.kind = I3D_WGRAD_PLAIN;
        problem(c->grad_pre, c->f_out, c->f_out, c->x, c->f_in, c->f_in, c->rows, nullptr, 0, c->rows);
    }
    // edge block: d[W_s | W_d] = dP^T h (rows >= Fo of the [2 Fo, Fh] product are the second column block of dW)
    if (do_pre) {
        I3dWgradOutput& o = out[no++];
        o.kind = I3D_WGRAD_PLAIN; o.n_groups = 1; o.first_problem = np; o.C = e->grad_W; o.ldc = e->ldw;
        o.c_split = Fo; o.c_delta = (long)Fh - (long)Fo * e->ldw;
        problem(dP, ld_dP, 2 * Fo, e->h, Fh, Fh, N, nullptr, 0, N);
    }
    // bond table: dQ = onehot^T dpre
    if (do_pre) {
        I3dWgradOutput& o = out[no++];
        o.kind = I3D_WGRAD_PLAIN; o.n_groups = 1; o.first_problem = np; o.C = e->grad_Q; o.ldc = Fo;
        problem(e->onehot, e->v_pad, e->v_pad, e->grad_pre, Fo, Fo, E, nullptr, 0, E);
    }
    void* ws = e->tail.gemm_workspace;
    const long wsb = e->tail.gemm_workspace_bytes;
    if (ws == nullptr || !i3d_wgrad_multi_supported(pr, np, out, no) || i3d_wgrad_multi_min_workspace_bytes(pr, np) > wsb) return 0;
    if (dry_run) return 1;                             // the layer is covered
    if (do_post) TRY(bias_final(&g->tail, N, g->f_out, g->grad_bias, wst));
    for (int i = a->n_pre_extra - 1; i >= 0 && do_pre; --i)
        TRY(bias_final_fc(&a->pre[i], wst, (a->fused_bn && a->msg_bf16 && i == a->n_pre_extra - 1) ? 1 : 0));
    if (do_pre) {
        if (edge_bwd_fused_ok(a) && edge_bias_on_chain(a)) {
            // (taken on the caller's stream at the end of the layer's backward: see there)
        } else if (edge_bwd_fused_ok(a))      // the bias gradient = column sum of dP[dst] (the layer's backward took i3d_bn_bwd_edge_sums)
            TRY(i3d_colsum_strided(a->DL + Fo, 2 * Fo + a->post.f_out, N, Fo, e->grad_bias, a->edge_bias_partial, wst));
        else
            TRY(bias_final(&e->tail, E, Fo, e->grad_bias, wst));
    }
    TRY(i3d_wgrad_multi(pr, np, out, no, ws, wsb, wst));
    if (!do_pre) return 1;
    const int V = e->q_rows;
    TRY(i3d_gemm_f32_ws(1, 0, Fo, e->f_q, V, e->grad_Q, Fo, e->q, e->f_q, e->grad_W + 2 * Fh, e->ldw, nullptr, 0, ws, wsb, wst));
    if (e->grad_q != nullptr)
        TRY(i3d_gemm_f32(0, 0, V, e->f_q, Fo, e->grad_Q, Fo, e->W + 2 * Fh, e->ldw, e->grad_q, e->f_q, nullptr, e->grad_q_accumulate, wst));
    return 1;
}

// the cheap part of the decision, taken before the chain is enqueued (the expensive part - alignment, scratch - is
// re-checked by pna_layer_wgrad_multi itself; dry_run: decide only)
static bool wgrad_multi_layer_ok(const I3dPnaLayerArgs* a);

// ---- one PNA layer ---------------------------------------------------------------------------------------
// Fused-BatchNorm form of the layer (a->fused_bn, fused_bn.hip): the statistics of every block come out of the epilogue of
// the kernel that produces its activation, the BatchNorm-apply of the pretrans blocks happens in the loads of their
// consumers (next GEMM / aggregation kernel): per layer three statistics passes and two apply passes over [E, F] / [N, F]
// tensors less than the block composites above, and edge.y / pre[i].y (the normalised activations) do not exist.  Only the
// posttrans output - which the next layer gathers from - is materialised by i3d_bn_apply_fwd.
static bool simple_act(int act) { return act == I3D_ACT_NONE || act == I3D_ACT_RELU || act == I3D_ACT_LEAKY_RELU; }

static int finalize_stats(const I3dBnTail* t, const float* partial, int tiles, int feat, float* aff, void* stream,
                          int eval_mode = 0) {
    if (eval_mode) return I3D_OK;      // running statistics: the affine vectors are already there, nothing is updated
    return i3d_bn_finalize_partials(partial, tiles, feat, t->eps, t->momentum, t->gamma, t->beta, t->mean, t->invstd,
                                    t->running_mean, t->running_var, t->num_batches_tracked, aff, stream);
}

static int pna_layer_fwd_fused(const I3dPnaLayerArgs* a, void* stream) {
    const I3dEdgeFcArgs* e = &a->edge;
    const int Fh = e->f_h, Fo = e->f_out, N = e->num_nodes, E = e->num_edges;
    I3D_CHECK_ARG(a->stats_ws != nullptr && a->n_post_extra == 0 && e->pre_keep == nullptr && simple_act(e->tail.act) &&
                      e->tail.post_act == I3D_ACT_NONE && a->aff[0] != nullptr, "fused BatchNorm: unsupported block shape");
    // edge block: P and Q as before, then gather-combine + activation + statistics in one pass
    const I3dGroupedFcArgs* pg = &a->post;
    const bool merged = merge_h_ok(a);
    const int WL = 2 * Fo + pg->f_out;        // merged: PL = h [W_s ; W_d ; W_h]^T + [0 | 0 | b_post], P = its first 2 Fo columns
    const float* P = e->P;
    int ldp = 2 * Fo;
    if (merged) {
        if (!a->weights_ready) TRY(pack_layer_weights(a, stream));
        if (panel_ok(a->Wcat_panel, Fh, WL))      // row-panel form: h read and split once per 208-column block (panel.hip)
            TRY(i3d_panel_gemm(N, WL, Fh, e->h, Fh, a->Wcat_panel, a->PL, WL, a->bcat, 0, stream));
        else
            TRY(i3d_gemm_f32(0, 1, N, WL, Fh, e->h, Fh, a->Wcat, Fh, a->PL, WL, a->bcat, 0, stream));
        P = a->PL;
        ldp = WL;
    } else {
        const long wdelta = (long)Fh - (long)Fo * e->ldw, wview = (long)(Fo - 1) * e->ldw + 2 * Fh;
        TRY(i3d_gemm_f32_blocks(0, 1, N, 2 * Fo, Fh, e->h, Fh, e->W, e->ldw, Fo, wdelta, wview, e->P, 2 * Fo, 0, 0, 0, nullptr, 0,
                                stream));
    }
    if (e->q != nullptr && !a->weights_ready)
        TRY(i3d_gemm_f32(0, 1, e->q_rows > 0 ? e->q_rows : E, Fo, e->f_q, e->q, e->f_q, e->W + 2 * Fh, e->ldw, e->Q, Fo,
                         nullptr, 0, stream));
    TRY(i3d_edge_combine_act_stats(P, ldp, e->q ? e->Q : nullptr, e->q_rows > 0 ? e->q_code : nullptr, e->bias, e->src_s,
                                   e->dst_s, E, Fo, e->tail.act, e->xact, a->stats_ws, stream));
    TRY(finalize_stats(&e->tail, a->stats_ws, cdiv(E, i3d_edge_stats_rows_per_tile(Fo)), Fo, a->aff[0], stream, a->eval_mode));
    const float* x = e->xact;
    const float* aff = a->aff[0];
    int f_in = Fo;
    for (int i = 0; i < a->n_pre_extra; ++i) {
        const I3dFcArgs* c = &a->pre[i];
        I3D_CHECK_ARG(c->pre_keep == nullptr && simple_act(c->tail.act) && c->tail.post_act == I3D_ACT_NONE &&
                          a->aff[i + 1] != nullptr && c->rows == E && c->f_in == f_in, "fused BatchNorm: unsupported block shape");
        // lin = BN_prev(x) W^T + b with the BatchNorm applied while x is staged; activation + statistics in the epilogue
        int stat_tiles = cdiv(E, 64);
        if (a->msg_bf16 && i == a->n_pre_extra - 1)      // the messages: stored as bf16 (K4 and this block's BatchNorm backward read them)
            TRY(i3d_gemm_f32_fused_bf16out(E, c->f_out, f_in, x, f_in, E, c->W, c->ldw, c->xact, c->f_out, c->bias, aff, c->tail.act,
                                           a->stats_ws, stream));
        else if (merged && panel_ok(c->W_fwd_panel, f_in, c->f_out) && (((uintptr_t)x | (uintptr_t)c->xact | (uintptr_t)aff) & 15) == 0) {
            // row-panel form (panel.hip): the BatchNorm prologue, bias, activation and the statistics of 32-row tiles in one launch
            TRY(i3d_panel_gemm_fused(E, c->f_out, f_in, x, f_in, c->W_fwd_panel, c->xact, c->f_out, c->bias, aff, c->tail.act, a->stats_ws,
                                     stream));
            stat_tiles = i3d_panel_stats_tiles(E);
        } else
            TRY(i3d_gemm_f32_fused(E, c->f_out, f_in, x, f_in, E, c->W, c->ldw, c->xact, c->f_out, c->bias, 0, aff, c->tail.act,
                                   a->stats_ws, nullptr, nullptr, 0, stream));
        TRY(finalize_stats(&c->tail, a->stats_ws, stat_tiles, c->f_out, a->aff[i + 1], stream, a->eval_mode));
        x = c->xact;
        aff = a->aff[i + 1];
        f_in = c->f_out;
    }
    if (a->agg_event_start != nullptr && a->agg_event_stop != nullptr) k4_time_next_launch(a->agg_event_start, a->agg_event_stop);
    TRY(i3d_pna_aggregate_fwd_ex(x, a->msg_bf16 && a->n_pre_extra > 0, aff, e->in_ptr, N, f_in, a->aggregators, a->n_aggregators, a->scalers,
                                 a->n_scalers, a->force_scalers, a->avg_d_log, const_cast<float*>(a->post.agg), stream));
    // posttrans: lin = h W_h^T + b, += agg W_D^T per in-degree group with the statistics in that launch's epilogue (the
    // degree groups cover every node - in-degree 0 included, with zero coefficients), then apply + residual
    const I3dGroupedFcArgs* p = &a->post;
    I3D_CHECK_ARG(p->pre_keep == nullptr && simple_act(p->tail.act), "fused BatchNorm: unsupported block shape");
    const int A = p->agg_width, Fp = p->f_out;
    if (!merged) TRY(i3d_gemm_f32(0, 1, N, Fp, p->f_h, p->h, p->f_h, p->W, p->ldw, p->xact, Fp, p->bias, 0, stream));
    if (!a->weights_ready)
        TRY(i3d_pna_combine_weights_fwd(p->W, p->ldw, p->f_h, Fp, A, p->n_groups, p->n_scalers, p->coef, p->WD, stream));
    // merged: the h-product is the last Fp columns of PL - the grouped GEMM takes its addend from there
    TRY(i3d_gemm_f32_fused_src(p->m_padded, Fp, A, p->agg, A, N, p->WD, A, p->xact, Fp, merged ? a->PL + 2 * Fo : nullptr, WL,
                               nullptr, 1, nullptr, p->tail.act, a->stats_ws, p->deg_rows, p->deg_tile_group, (long)Fp * A, stream));
    if (a->eval_mode)
        return i3d_bn_eval_fwd(p->xact, N, Fp, p->tail.running_mean, p->tail.running_var, p->tail.eps, p->tail.gamma, p->tail.beta,
                               p->tail.post_act, p->residual, p->y, stream);
    TRY(finalize_stats(&p->tail, a->stats_ws, p->m_padded / 64, Fp, nullptr, stream));
    return i3d_bn_apply_fwd(p->xact, N, Fp, p->tail.mean, p->tail.invstd, p->tail.gamma, p->tail.beta, p->tail.post_act,
                            p->residual, p->y, stream);
}

extern "C" int i3d_pna_layer_weights_fwd(const I3dPnaLayerArgs* a, void* stream) {
    I3D_CHECK_ARG(a != nullptr && a->fused_bn, "fused_bn layers only");
    const I3dEdgeFcArgs* e = &a->edge;
    const I3dGroupedFcArgs* p = &a->post;
    if (e->q != nullptr)
        TRY(i3d_gemm_f32(0, 1, e->q_rows > 0 ? e->q_rows : e->num_edges, e->f_out, e->f_q, e->q, e->f_q, e->W + 2 * e->f_h, e->ldw,
                         e->Q, e->f_out, nullptr, 0, stream));
    if (merge_h_ok(a)) TRY(pack_layer_weights(a, stream));
    return i3d_pna_combine_weights_fwd(p->W, p->ldw, p->f_h, p->f_out, p->agg_width, p->n_groups, p->n_scalers, p->coef, p->WD,
                                       stream);
}

extern "C" int i3d_wgrad_stream_fork(void* stream, void** side) {
    I3D_CHECK_ARG(side != nullptr, "null");
    *side = fork_wgrad(aux_for((hipStream_t)stream), stream);
    return I3D_OK;
}

// the weight-gradient stream of `stream` as it stands (no new fork: work enqueued there runs behind whatever the last fork
// ordered it after); `stream` itself when there is none
extern "C" int i3d_wgrad_stream_peek(void* stream, void** side) {
    I3D_CHECK_ARG(side != nullptr, "null");
    Aux* x = aux_for((hipStream_t)stream);
    *side = x != nullptr ? (void*)x->s : stream;
    return I3D_OK;
}

extern "C" long i3d_pna_layer_stats_floats(int num_nodes, int num_edges, int m_padded, int f) {
    (void)num_nodes;
    const int rpt = i3d_edge_stats_rows_per_tile(f);
    long tiles = cdiv(num_edges, rpt > 0 ? rpt : 1);
    tiles = std::max<long>(tiles, 2L * cdiv(num_edges, 64));       // (the row-panel product's 32-row tiles)
    tiles = std::max<long>(tiles, m_padded / 64 + 1);
    return tiles * 3 * f;
}

extern "C" int i3d_pna_layer_fwd(const I3dPnaLayerArgs* a, void* stream) {
    I3D_CHECK_ARG(a != nullptr && a->n_pre_extra >= 0 && a->n_pre_extra <= I3D_MAX_EXTRA_FC && a->n_post_extra >= 0 &&
                      a->n_post_extra <= I3D_MAX_EXTRA_FC, "bad arguments");
    if (a->fused_bn) return pna_layer_fwd_fused(a, stream);
    TRY(i3d_edge_fc_bn_fwd(&a->edge, stream));
    for (int i = 0; i < a->n_pre_extra; ++i) TRY(i3d_fc_bn_fwd(&a->pre[i], stream));
    if (a->agg_event_start != nullptr && a->agg_event_stop != nullptr) k4_time_next_launch(a->agg_event_start, a->agg_event_stop);
    TRY(i3d_pna_aggregate_fwd(a->msg, a->edge.in_ptr, a->edge.num_nodes, a->edge.f_out, a->aggregators, a->n_aggregators,
                              a->scalers, a->n_scalers, a->force_scalers, a->avg_d_log, const_cast<float*>(a->post.agg), stream));
    TRY(i3d_grouped_fc_bn_fwd(&a->post, stream));
    for (int i = 0; i < a->n_post_extra; ++i) TRY(i3d_fc_bn_fwd(&a->postx[i], stream));
    return I3D_OK;
}

// Two forks per layer (three measured 2.493 against 2.477 ms: the third only costs the host its two event calls).
extern "C" int i3d_pna_layer_bwd(const I3dPnaLayerArgs* a, void* stream) {
    I3D_CHECK_ARG(a != nullptr && a->n_pre_extra >= 0 && a->n_pre_extra <= I3D_MAX_EXTRA_FC && a->n_post_extra >= 0 &&
                      a->n_post_extra <= I3D_MAX_EXTRA_FC, "bad arguments");
    // The chain on the caller's stream; the weight gradients on the side stream behind three forks: right after the posttrans
    // blocks' chain part (their weight gradients are the largest, and the aggregation backward and the pretrans chain run
    // next to them), after the edge block's BatchNorm backward (the later pretrans blocks' and everything behind dQ need
    // grad_pre only), and when the edge block's dP exists.  Measured: one fork per block (4 per layer) costs the host
    // 0.07 ms more per step for nothing, forking everything late (after the pretrans chain) costs the GPU 2 %.  One join.
    Aux* x = aux_for((hipStream_t)stream);
    for (int i = a->n_post_extra - 1; i >= 0; --i) TRY(fc_bn_bwd_chain(&a->postx[i], stream));
    // residual layer whose caller passes ONE buffer as grad_out and post.grad_h: dh = dh_out + (layer's own terms), the
    // first of which is accumulated on top of dh_out instead of a separate add at the end.  Only without extra posttrans
    // blocks (then post.grad_y IS grad_out and its last reader, the BatchNorm backward, runs before that GEMM).
    const bool inplace = a->residual && a->post.grad_h == a->grad_out && a->n_post_extra == 0;
    I3D_CHECK_ARG(a->post.grad_h != a->grad_out || inplace, "grad_h may alias grad_out only for a residual layer without extra blocks");
    // round 3: every weight gradient of the layer from one launch behind ONE fork (after dP exists) when the layer's shape
    // allows it; otherwise the per-block launches behind two / three forks
    const bool multi = wgrad_multi_layer_ok(a);
    // merged h-products (I3dPnaLayerArgs.merge_h): dlin goes into the last columns of DL = [dP | dlin], the posttrans block's
    // own dL/dh GEMM is dropped and ONE GEMM  dL/dh (+)= DL [W_s ; W_d ; W_h]  closes the layer (needs the one-launch weight
    // gradients: they take dlin / dP with DL's row pitch)
    const bool merged = multi && merge_h_ok(a) && a->DL != nullptr;
    const long n = (long)a->edge.num_nodes * a->edge.f_h;
    if (merged) {
        const I3dGroupedFcArgs* g = &a->post;
        const int WLb = 2 * a->edge.f_out + g->f_out;
        float* dlin = a->DL + 2 * a->edge.f_out;
        TRY(i3d_bn_bwd_strided(g->grad_y, g->xact, g->pre_keep, g->num_nodes, g->f_out, g->tail.act, g->tail.post_act, g->tail.mean,
                               g->tail.invstd, g->tail.gamma, g->tail.beta, g->grad_gamma, g->grad_beta, dlin, WLb, g->grad_bias,
                               g->tail.workspace, g->tail.bias_partial, stream));
        TRY(i3d_gemm_f32_grouped(0, g->m_padded, g->agg_width, g->f_out, dlin, WLb, g->num_nodes, g->deg_rows, g->deg_tile_group, g->WD,
                                 g->agg_width, (long)g->f_out * g->agg_width, g->grad_agg, g->agg_width, 0, stream));
    } else {
        TRY(grouped_fc_bn_bwd_chain(&a->post, stream, inplace ? 1 : 0));
        // separate buffers: the residual's term right here, so that both forms add in the same order (dh_out + dlin W_h) + ...
        if (a->residual && !inplace) TRY(i3d_add_inplace(a->post.grad_h, a->grad_out, n, stream));
    }
    void* wst = stream;
    if (multi && a->wgrad_split) {
        wst = fork_wgrad(x, stream);
        const int took = pna_layer_wgrad_multi(a, wst, false, 1);
        if (took < 0) return took;
        I3D_CHECK_ARG(took == 1, "weight-gradient launch refused a layer it had accepted");
    }
    if (!multi) {
        wst = fork_wgrad(x, stream);
        TRY(grouped_fc_bn_bwd_wgrad(&a->post, wst));
        for (int i = a->n_post_extra - 1; i >= 0; --i) TRY(fc_bn_bwd_wgrad(&a->postx[i], wst));
    }
    // fused BatchNorm: a->msg is the last pretrans block's activation BEFORE its BatchNorm, applied on load (aff);
    // pre[i].x / the aggregation read raw activations, so the weight gradients of pre[i] are corrected with aff[i]
    const float* msg_aff = a->fused_bn ? a->aff[a->n_pre_extra] : nullptr;
    const int f_msg = a->n_pre_extra > 0 ? a->pre[a->n_pre_extra - 1].f_out : a->edge.f_out;
    const int msg16 = a->fused_bn && a->msg_bf16 && a->n_pre_extra > 0;
    TRY(i3d_pna_aggregate_bwd_ex(a->post.grad_agg, a->msg, msg16, msg_aff, a->edge.in_ptr, a->edge.num_nodes, f_msg, a->aggregators,
                                 a->n_aggregators, a->scalers, a->n_scalers, a->force_scalers, a->avg_d_log, a->grad_msg, stream));
    for (int i = a->n_pre_extra - 1; i >= 0; --i) TRY(fc_bn_bwd_chain(&a->pre[i], stream, msg16 && i == a->n_pre_extra - 1));
    const bool edge_fused = multi && merged && edge_bwd_fused_ok(a);
    if (edge_fused) {       // BatchNorm backward of the edge block with the data gradient formed inside the two segmented sums
        const I3dEdgeFcArgs* e = &a->edge;
        const int WLb = 2 * e->f_out + a->post.f_out;
        // round 6: where the one-launch BatchNorm backward takes the shape (bn.hip: bn_bwd_fused_kernel, 16-17 us at batch 512) it
        // writes g and the pair of segmented sums follows (8.5 us) - against reduction 15.5 us + i3d_bn_bwd_edge_sums 17-20 us.
        // (profiles/r06_ab_edge_onelaunch.txt: 1.951 -> 1.925 ms)
        if (i3d_bn_bwd_one_launch_supported(e->num_edges, e->f_out)) {
            TRY(i3d_bn_bwd_strided(e->grad_y, e->xact, nullptr, e->num_edges, e->f_out, e->tail.act, I3D_ACT_NONE, e->tail.mean, e->tail.invstd,
                                   e->tail.gamma, e->tail.beta, e->grad_gamma, e->grad_beta, e->grad_pre, e->f_out, nullptr,
                                   e->tail.workspace, nullptr, stream));
            TRY(i3d_segment_sum_pair(e->grad_pre, e->f_out, e->out_ptr, e->out_epos, a->DL, e->in_ptr, nullptr, a->DL + e->f_out,
                                     e->num_nodes, e->f_out, WLb, stream));
        } else
        TRY(i3d_bn_bwd_edge_sums(e->grad_y, e->xact, e->num_edges, e->f_out, e->tail.act, e->tail.mean, e->tail.invstd, e->tail.gamma,
                                 e->tail.beta, e->grad_gamma, e->grad_beta, e->grad_pre, e->in_ptr, e->out_ptr, e->out_epos, e->num_nodes,
                                 a->DL, a->DL + e->f_out, WLb, e->tail.workspace, stream));
    } else {
        TRY(edge_fc_bn_bwd_tail(&a->edge, stream));
    }
    if (multi) {
        if (edge_fused) {
        } else if (merged) {       // dP[src] | dP[dst] straight into the first 2 Fo columns of DL
            const I3dEdgeFcArgs* e = &a->edge;
            const int WLb = 2 * e->f_out + a->post.f_out;
            TRY(i3d_segment_sum_pair(e->grad_pre, e->f_out, e->out_ptr, e->out_epos, a->DL, e->in_ptr, nullptr, a->DL + e->f_out,
                                     e->num_nodes, e->f_out, WLb, stream));
        } else {
            TRY(edge_fc_bn_bwd_sums(&a->edge, stream));
        }
        wst = fork_wgrad(x, stream);
        const int took = pna_layer_wgrad_multi(a, wst, false, a->wgrad_split ? 2 : 0);
        if (took < 0) return took;
        I3D_CHECK_ARG(took == 1, "weight-gradient launch refused a layer it had accepted");
    } else {
        TRY(edge_fc_bn_bwd_sums(&a->edge, stream));
        wst = fork_wgrad(x, stream);
        for (int i = a->n_pre_extra - 1; i >= 0; --i) TRY(fc_bn_bwd_wgrad(&a->pre[i], wst, a->fused_bn ? a->aff[i] : nullptr));
        TRY(edge_fc_bn_bwd_wgrad(&a->edge, wst));
    }
    if (merged) {
        const int WLb = 2 * a->edge.f_out + a->post.f_out;
        if (panel_ok(a->Wcat_dgrad_panel, WLb, a->edge.f_h, 1024) && (((uintptr_t)a->DL | (uintptr_t)a->post.grad_h) & 15) == 0)
            TRY(i3d_panel_gemm(a->edge.num_nodes, a->edge.f_h, WLb, a->DL, WLb, a->Wcat_dgrad_panel, a->post.grad_h, a->edge.f_h, nullptr,
                               inplace ? 1 : 0, stream));      // (32-row slabs at batch 512: 37 -> 27 us)
        else
            TRY(i3d_gemm_f32(0, 0, a->edge.num_nodes, a->edge.f_h, WLb, a->DL, WLb, a->Wcat, a->edge.f_h, a->post.grad_h, a->edge.f_h,
                             nullptr, inplace ? 1 : 0, stream));
        if (a->residual && !inplace) TRY(i3d_add_inplace(a->post.grad_h, a->grad_out, n, stream));
    } else {
        TRY(edge_fc_bn_bwd_dgrad(&a->edge, stream, a->post.grad_h));       // post.grad_h += edge block's dh
    }
    if (edge_fused && edge_bias_on_chain(a)) {
        const int Fo = a->edge.f_out;
        TRY(i3d_colsum_strided(a->DL + Fo, 2 * Fo + a->post.f_out, a->edge.num_nodes, Fo, a->edge.grad_bias, a->edge_bias_partial, stream));
    }
    if (a->defer_join) return I3D_OK;
    return join_wgrad(x, stream);
}

static bool wgrad_multi_layer_ok(const I3dPnaLayerArgs* a) { return pna_layer_wgrad_multi(a, nullptr, true) == 1; }

extern "C" int i3d_wgrad_stream_join(void* stream) { return join_wgrad(aux_for((hipStream_t)stream), stream); }

// ---- timing events -------------------------------------------------------------------------------------
extern "C" int i3d_event_create(void** event) {
    I3D_CHECK_ARG(event != nullptr, "null");
    hipEvent_t e;
    // timing-only events: no system-scope fence (cache write-back + invalidate) when they are recorded - the HIP API
    // documents this flag for exactly that use ("can improve the accuracy of timing measurements")
    if (hipEventCreateWithFlags(&e, hipEventDisableSystemFence) != hipSuccess &&
        hipEventCreateWithFlags(&e, hipEventReleaseToDevice) != hipSuccess && hipEventCreate(&e) != hipSuccess) {
        i3d::set_error("hipEventCreate failed");
        return I3D_ERR_LAUNCH;
    }
    *event = (void*)e;
    return I3D_OK;
}

extern "C" int i3d_event_destroy(void* event) {
    if (event != nullptr) (void)hipEventDestroy((hipEvent_t)event);
    return I3D_OK;
}

extern "C" int i3d_event_record(void* event, void* stream) {
    I3D_CHECK_ARG(event != nullptr, "null");
    if (hipEventRecord((hipEvent_t)event, (hipStream_t)stream) != hipSuccess) {
        i3d::set_error("hipEventRecord failed");
        return I3D_ERR_LAUNCH;
    }
    return I3D_OK;
}

extern "C" int i3d_event_elapsed_ms(void* start, void* stop, float* ms) {
    I3D_CHECK_ARG(start != nullptr && stop != nullptr && ms != nullptr, "null");
    if (hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop) != hipSuccess) {
        i3d::set_error("hipEventElapsedTime failed (events not completed?)");
        return I3D_ERR_LAUNCH;
    }
    return I3D_OK;
}
