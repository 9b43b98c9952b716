// Whole-model host sequencing: the PNA forward / backward pass of a training step from ONE C call each.
//
// The step is ~270 kernel launches; measured on MI355X (tools/host_segments.py, tools/launch_floor.py) the launches
// themselves cost the host ~0.8 ms per step, the Python around them (per-layer pointer arithmetic and ctypes struct fills,
// the tape, ~20 small wrappers for encoders / readout / head, torch allocations) another ~1.8 ms - the step was bound by
// the host, not by the GPU.  This file moves all of that below the C ABI: Python hands over the model description
// (parameter / state / gradient pointers, built once per model), the batch description (sizes + index pointers) and two
// buffers (saved activations, backward scratch); the memory layout, the argument structs of the layer composites
// (composite.hip) and the launch order are computed here.  The kernels and their order are the ones the Python path
// issues (fused-BatchNorm layer form, degree-grouped posttrans, bond table), the results are the same bits.
//
// Reference call chain replaced: PNA.forward -> PNAGNN.forward -> [PNALayer.forward]* -> readout -> MLP head
// (models/pna.py:131-135, 161-166, 199-213, 127-129) and its autograd backward.
#include "common.h"


#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

using namespace i3d;

#define TRY(call)                 \
    do {                          \
        int rc_ = (call);         \
        if (rc_ != I3D_OK) return rc_; \
    } while (0)

namespace {

inline long al4(long n) { return (n + 3) & ~3L; }

struct Bump {           // bump allocator over a caller-provided float buffer; base == null: sizing pass
    float* base;
    long used = 0;
    explicit Bump(float* b) : base(b) {}
    float* take(long n) {
        float* p = base ? base + used : nullptr;
        used += al4(n);
        return p;
    }
};

struct HeadSaved {
    float* xact = nullptr;
    float* y = nullptr;
    float* mean = nullptr;
    float* invstd = nullptr;
};

struct PnaCtx {
    I3dPnaModel m;
    I3dPnaBatch b;
    std::vector<I3dPnaLayerArgs> layers;
    std::vector<float*> h;          // h[0] = atom embedding, h[l + 1] = output of layer l (h[L] = the caller's node_emb)
    float* bond_table = nullptr;    // [V, F]
    int* codes = nullptr;           // [E]
    float* onehot = nullptr;        // [E, v_pad]
    float* readout = nullptr;       // [B, n_readout * F]
    std::vector<HeadSaved> head;
    std::vector<I3dFcArgs> head_args;
    float* out = nullptr;           // [B, target]
    float coef[128];
    int n_scalers_cfg = 0;
    long saved_floats = 0;
    float* hot_atoms = nullptr;     // [N, va] / [n_comb, vb] multi-hot matrices of the encoders' weight gradients (functions of
    float* hot_bonds = nullptr;     // the batch's categorical features only: built on the side stream during the forward pass)
    bool hot_ready = false;
    int gh_cur = 0;                 // which of the two dL/dh buffers holds the gradient after the layers done so far (backward)
};

// (measured and left off: the bias gradients are finalised from row-chunk partials on the weight-gradient stream instead of
// inside the data-gradient pass (I3dBnTail.bias_partial).  Off by default: measured on one box (tools/ab.sh, 4 interleaved
// runs of 300 steps) 2.816 ms deferred against 2.787 ms in-launch - it takes ~10 us per BatchNorm off the main stream's
// chain but adds three small launches per layer to the side stream, whose join is what the next layer waits for: at batch
// 512 the backward pass is bound by the TOTAL work of the two streams, not by the chain.
// the [rows, v] multi-hot matrices of the atom / bond-combination features (columns: the concatenated vocabularies, padded
// to 32): embedding-table gradients are their transposes times dL/d(embedding)
int encoder_multihot(const PnaCtx& c, float* hot_atoms, float* hot_bonds, void* stream) {
    const I3dPnaModel& m = c.m;
    const I3dPnaBatch& b = c.b;
    int offs[16], o = 0;
    for (int k = 0; k < m.n_atom_tables; ++k) { offs[k] = o; o += m.atom_dims[k]; }
    TRY(i3d_multihot(b.atom_feat, nullptr, b.num_nodes, m.n_atom_tables, offs, (o + 31) / 32 * 32, hot_atoms, stream));
    o = 0;
    for (int k = 0; k < m.n_bond_tables; ++k) { offs[k] = o; o += m.bond_dims[k]; }
    return i3d_multihot(b.comb, nullptr, b.n_comb, m.n_bond_tables, offs, (o + 31) / 32 * 32, hot_bonds, stream);
}

bool hoist_weights() {
    return true;
}

bool defer_bias() {
    return false;
}

// I3D_WGRAD_JOIN=layer: the backward pass of every layer waits for its weight gradients (round-1 behaviour).  Default
// `model`: one join at the end of the model's backward - the weight-gradient stream turned out to be the critical path
// of a layer (~214 us of GEMMs per layer behind the first fork against ~150 us left on the chain), a join per layer
// makes the chain of the NEXT layer wait for it.  Costs scratch: what that stream reads or writes is kept per layer.
bool join_per_layer() {
    return false;      // (one join per model backward: per-layer joins measured slower, docs/history)
}

// The first layer's weight gradients (the last ones of a backward pass, and what the step waits for at its very end) as two
// launches, the posttrans products early, next to that layer's chain: only the pretrans / bond-table products are left when the
// chain ends.  Before the chain's kernels had wave priority this lost (2.289 against 2.253 ms: the early launch took the CUs from
// the chain it was meant to hide behind); with it: 2.141 against 2.152 ms (tools/ab.sh, 5 interleaved runs).  Every layer split
// the same way (every layer) loses 40 us: one more launch + reduction per layer for work that was hidden anyway.
int split_wgrad() {     // 0: no layer, 1: the first layer (last of the backward pass), 2: every layer
    return 1;
}

// bf16 matmul mode: a layer's messages ([E, F], the last pretrans block's activation) stored as bf16 once they are large enough to
// be bound by their bytes (I3D_MSG_BF16=0: never, =force: at every size; default: E * F * 4 >= 32 MB - the QMugs shape, not the QM9 one)
bool msg_bf16_storage(const I3dPnaModel& m, long E) {
    if (i3d_get_matmul_precision() == 0 || m.n_pre < 2) return false;
    const char* e = getenv("I3D_MSG_BF16");
    if (e != nullptr && e[0] == '0') return false;
    if (e != nullptr && e[0] == 'f') return true;
    return E * (long)m.hidden * 4 >= (32L << 20);
}

// I3D_MERGE_H=0: the products that read the node features (edge block's P, posttrans block's h-term, and their data gradients)
// as separate GEMMs (round 2) instead of one per direction
bool merge_h() {
    return true;
}

bool simple_act(int act) { return act == I3D_ACT_NONE || act == I3D_ACT_RELU || act == I3D_ACT_LEAKY_RELU; }

void fill_tail(I3dBnTail& t, const I3dFcParams& p, float* mean, float* invstd) {
    t.act = p.act;
    t.post_act = I3D_ACT_NONE;
    t.eps = p.eps;
    t.momentum = p.momentum;
    t.gamma = p.gamma;
    t.beta = p.beta;
    t.running_mean = p.running_mean;
    t.running_var = p.running_var;
    t.mean = mean;
    t.invstd = invstd;
    t.num_batches_tracked = p.num_batches_tracked;
    t.workspace = nullptr;
    t.gemm_workspace = nullptr;
    t.gemm_workspace_bytes = 0;
    t.bias_partial = nullptr;
}

void set_ws(I3dBnTail& t, void* bn_ws, void* gemm_ws, long gemm_ws_bytes) {
    t.workspace = bn_ws;
    t.gemm_workspace = gemm_ws;
    t.gemm_workspace_bytes = gemm_ws_bytes;
}

// scaler coefficient of an in-degree group, reference models/pna.py:57-68 (np.log in float64, cast to fp32); D = 0: the
// aggregate of a node without in-edges is a zero row, every coefficient 0
float scaler_coef(int scaler, int D, float avg) {
    if (D == 0) return 0.f;
    if (scaler == I3D_SCALE_AMPLIFICATION) return (float)(std::log((double)D + 1.0) / (double)avg);
    if (scaler == I3D_SCALE_ATTENUATION) return (float)((double)avg / std::log((double)D + 1.0));
    return 1.f;
}

// dW = dY^T X as the Python-sequenced path issues it (ops.gemm): through the split-K scratch from 1024 rows on
int wgrad(int M, int N, int K, const float* A, int lda, const float* B, int ldb, float* C, int ldc, void* ws, long ws_bytes,
          void* stream) {
    if (K >= 1024) return i3d_gemm_f32_ws(1, 0, M, N, K, A, lda, B, ldb, C, ldc, nullptr, 0, ws, ws_bytes, stream);
    return i3d_gemm_f32(1, 0, M, N, K, A, lda, B, ldb, C, ldc, nullptr, 0, stream);
}

// the head's backward buffers (read by its weight gradients on the side stream: kept apart from the layers' shared region),
// and a column-reduction scratch of that stream's own
long head_floats(const I3dPnaModel& m, long B) {
    long t = 0;
    for (int i = 0; i < m.n_head; ++i) t += al4(B * (long)m.head[i].f_in) + 2 * al4(B * (long)m.head[i].f_out);
    return t;
}
long head_colsum_floats(const I3dPnaModel& m, long B) {
    long t = 0;
    for (int i = 0; i < m.n_head; ++i) t = std::max(t, (i3d_colreduce_workspace_bytes((int)B, m.head[i].f_out) + 3) / 4);
    return al4(t);
}

// floats of the buffers of layer l's backward pass that the weight-gradient stream reads or writes
long side_floats(const I3dPnaModel& m, const I3dPnaBatch& b, int l) {
    const long N = b.num_nodes, E = b.num_edges, F = m.hidden;
    const long f_msg = m.pre[l][m.n_pre - 1].f_out, A = m.n_aggregators * f_msg, Fo0 = m.pre[l][0].f_out;
    long t = al4(N * F) + al4((long)b.n_groups * F * A) + al4(i3d_bn_bias_partial_floats((int)F));
    for (int i = 1; i < m.n_pre; ++i) t += al4(E * (long)m.pre[l][i].f_out) + al4(i3d_bn_bias_partial_floats(m.pre[l][i].f_out));
    t += al4(E * Fo0) + al4(N * 2 * Fo0) + al4((long)b.v_pad * Fo0) + al4(i3d_bn_bias_partial_floats((int)Fo0));
    t += al4(N * (2 * Fo0 + F));        // DL (merged h-products)
    return t;
}

int check_model(const I3dPnaModel* m, const I3dPnaBatch* b) {
    I3D_CHECK_ARG(m != nullptr && b != nullptr, "null");
    I3D_CHECK_ARG(m->n_layers >= 1 && m->n_layers <= I3D_MAX_LAYERS && m->n_pre >= 1 && m->n_pre <= I3D_MAX_EXTRA_FC + 1,
                  "1..16 layers, 1..4 pretrans blocks");
    I3D_CHECK_ARG(m->hidden > 0 && m->hidden % 4 == 0, "hidden_dim must be a multiple of 4");
    I3D_CHECK_ARG(m->n_head >= 1 && m->n_head <= I3D_MAX_HEAD_FC && m->n_readout >= 1 && m->n_readout <= 4, "bad head");
    I3D_CHECK_ARG(m->n_scalers >= 2 && m->n_scalers <= 4 && m->n_aggregators >= 1 && m->n_aggregators <= 8,
                  "degree-grouped posttrans needs 2..4 scalers");
    I3D_CHECK_ARG(b->num_nodes > 0 && b->num_edges > 0 && b->num_graphs > 0, "empty batch");
    I3D_CHECK_ARG(b->n_groups >= 1 && b->n_groups <= 32 && b->n_groups * m->n_scalers <= 128 && b->m_padded % 64 == 0,
                  "bad degree groups");
    I3D_CHECK_ARG(b->n_comb >= 1 && b->v_pad >= b->n_comb && b->v_pad % 4 == 0, "bad bond table");
    for (int l = 0; l < m->n_layers; ++l) {
        for (int i = 0; i < m->n_pre; ++i) {
            const I3dFcParams& p = m->pre[l][i];
            I3D_CHECK_ARG(p.gamma != nullptr && simple_act(p.act) && p.f_out % 4 == 0 && p.f_in % 4 == 0 && p.f_in <= 3 * 1024,
                          "pretrans block: BatchNorm, activation none / ReLU / LeakyReLU, widths multiples of 4");
        }
        const I3dFcParams& p = m->post[l];
        I3D_CHECK_ARG(p.gamma != nullptr && simple_act(p.act) && p.f_out == m->hidden, "posttrans block");
    }
    return I3D_OK;
}

// Everything a forward pass keeps for the backward pass, laid out in `saved` (null: sizing).  Fills ctx.
long plan_forward(PnaCtx& c, float* saved, float* node_emb, float* out) {
    const I3dPnaModel& m = c.m;
    const I3dPnaBatch& b = c.b;
    const int N = b.num_nodes, E = b.num_edges, B = b.num_graphs, F = m.hidden, L = m.n_layers;
    Bump ar(saved);
    c.h.assign(L + 1, nullptr);
    c.h[0] = ar.take((long)N * F);
    c.bond_table = ar.take((long)b.n_comb * F);
    c.codes = reinterpret_cast<int*>(ar.take(E));
    c.onehot = ar.take((long)E * b.v_pad);
    // per-degree scaler coefficients (shared by the layers)
    c.n_scalers_cfg = m.n_scalers;
    for (int g = 0; g < b.n_groups; ++g)
        for (int s = 0; s < m.n_scalers; ++s) c.coef[g * m.n_scalers + s] = scaler_coef(m.scalers[s], b.group_degree[g], m.avg_d_log);
    int f_max = F;
    for (int l = 0; l < L; ++l)
        for (int i = 0; i < m.n_pre; ++i) f_max = std::max(f_max, m.pre[l][i].f_out);
    float* stats_ws = ar.take(i3d_pna_layer_stats_floats(N, E, b.m_padded, f_max));      // scratch, shared by all layers
    c.layers.assign(L, I3dPnaLayerArgs());
    for (int l = 0; l < L; ++l) {
        I3dPnaLayerArgs& a = c.layers[l];
        std::memset(&a, 0, sizeof(a));
        a.fused_bn = 1;
        a.stats_ws = stats_ws;
        c.h[l + 1] = (l == L - 1) ? node_emb : ar.take((long)N * F);
        // ---- pretrans block 0: edge gather-combine
        I3dEdgeFcArgs& e = a.edge;
        const I3dFcParams& p0 = m.pre[l][0];
        const int Fo0 = p0.f_out;
        fill_tail(e.tail, p0, ar.take(Fo0), ar.take(Fo0));
        e.num_nodes = N; e.num_edges = E; e.f_h = F; e.f_q = F; e.f_out = Fo0; e.ldw = p0.f_in;
        e.q_rows = b.n_comb; e.v_pad = b.v_pad; e.q_code = c.codes; e.onehot = c.onehot;
        e.h = c.h[l]; e.q = c.bond_table; e.W = p0.W; e.bias = p0.bias;
        e.src_s = b.src_s; e.dst_s = b.dst_s; e.in_ptr = b.in_ptr; e.out_ptr = b.out_ptr; e.out_epos = b.out_epos;
        e.Q = ar.take((long)b.n_comb * Fo0);
        e.P = ar.take((long)N * 2 * Fo0);
        if (merge_h()) {
            a.merge_h = 1;
            a.Wcat = ar.take((long)(2 * Fo0 + F) * F);
            a.bcat = ar.take(2 * Fo0 + F);
            a.PL = ar.take((long)N * (2 * Fo0 + F));
            a.Wcat_panel = ar.take(i3d_panel_packed_bytes(2 * Fo0 + F, F) / 4);      // (csrc/panel.hip; packed where Wcat is)
            a.Wcat_dgrad_panel = ar.take(i3d_panel_packed_bytes(F, 2 * Fo0 + F) / 4);
        }
        e.xact = ar.take((long)E * Fo0);
        a.aff[0] = ar.take(3L * Fo0);
        const float* x = e.xact;
        int f_in = Fo0;
        a.n_pre_extra = m.n_pre - 1;
        for (int i = 1; i < m.n_pre; ++i) {
            const I3dFcParams& p = m.pre[l][i];
            I3dFcArgs& fc = a.pre[i - 1];
            fill_tail(fc.tail, p, ar.take(p.f_out), ar.take(p.f_out));
            fc.rows = E; fc.f_in = f_in; fc.f_out = p.f_out; fc.ldw = p.f_in;
            fc.x = x; fc.W = p.W; fc.bias = p.bias;
            fc.xact = ar.take((long)E * p.f_out);
            if (merge_h()) {
                fc.W_dgrad_panel = ar.take(i3d_panel_packed_bytes(f_in, p.f_out) / 4);
                fc.W_fwd_panel = ar.take(i3d_panel_packed_bytes(p.f_out, f_in) / 4);
            }
            a.aff[i] = ar.take(3L * p.f_out);
            x = fc.xact;
            f_in = p.f_out;
        }
        // ---- aggregation (identity block only: the scalers live in the per-degree weights)
        a.n_aggregators = m.n_aggregators;
        for (int i = 0; i < m.n_aggregators; ++i) a.aggregators[i] = m.aggregators[i];
        a.n_scalers = 1; a.scalers[0] = I3D_SCALE_IDENTITY; a.force_scalers = 0; a.avg_d_log = m.avg_d_log;
        a.msg = x;
        const int A = m.n_aggregators * f_in;
        // ---- posttrans: degree-grouped block
        I3dGroupedFcArgs& g = a.post;
        const I3dFcParams& pp = m.post[l];
        fill_tail(g.tail, pp, ar.take(F), ar.take(F));
        g.num_nodes = N; g.f_h = F; g.f_out = F; g.agg_width = A; g.ldw = pp.f_in;
        g.n_groups = b.n_groups; g.n_scalers = m.n_scalers; g.m_padded = b.m_padded;
        for (int k = 0; k < b.n_groups; ++k) { g.group_start[k] = b.group_start[k]; g.group_count[k] = b.group_count[k]; }
        for (int k = 0; k < b.n_groups * m.n_scalers; ++k) g.coef[k] = c.coef[k];
        g.h = c.h[l]; g.W = pp.W; g.bias = pp.bias;
        g.agg = ar.take((long)N * A);
        g.deg_rows = b.deg_rows; g.deg_tile_group = b.deg_tile_group;
        g.WD = ar.take((long)b.n_groups * F * A);
        g.xact = ar.take((long)N * F);
        g.y = c.h[l + 1];
        a.n_post_extra = 0;
        a.residual = m.residual ? 1 : 0;
        g.residual = m.residual ? c.h[l] : nullptr;
    }
    // ---- readout + head
    c.readout = ar.take((long)B * m.n_readout * F);
    c.head.assign(m.n_head, HeadSaved());
    c.head_args.assign(m.n_head, I3dFcArgs());
    const float* x = c.readout;
    for (int i = 0; i < m.n_head; ++i) {
        const I3dFcParams& p = m.head[i];
        I3dFcArgs& fc = c.head_args[i];
        std::memset(&fc, 0, sizeof(fc));
        const bool last = i == m.n_head - 1;
        float* y = last ? out : ar.take((long)B * p.f_out);
        fc.rows = B; fc.f_in = p.f_in; fc.f_out = p.f_out; fc.ldw = p.f_in;
        fc.x = x; fc.W = p.W; fc.bias = p.bias; fc.y = y;
        if (p.gamma != nullptr) {
            c.head[i].mean = ar.take(p.f_out);
            c.head[i].invstd = ar.take(p.f_out);
            fill_tail(fc.tail, p, c.head[i].mean, c.head[i].invstd);
            fc.xact = ar.take((long)B * p.f_out);
            if (!simple_act(p.act)) fc.pre_keep = ar.take((long)B * p.f_out);
        } else if (p.act != I3D_ACT_NONE) {
            fc.xact = ar.take((long)B * p.f_out);        // pre-activation, kept for the activation's derivative
        }
        c.head[i].xact = fc.xact;
        c.head[i].y = y;
        x = y;
    }
    c.out = out;
    {
        long oa = 0, ob = 0;
        for (int k = 0; k < m.n_atom_tables; ++k) oa += m.atom_dims[k];
        for (int k = 0; k < m.n_bond_tables; ++k) ob += m.bond_dims[k];
        c.hot_atoms = ar.take((long)N * ((oa + 31) / 32 * 32));
        c.hot_bonds = ar.take((long)b.n_comb * ((ob + 31) / 32 * 32));
    }
    c.saved_floats = ar.used;
    return ar.used;
}

}  // namespace

extern "C" long i3d_pna_model_saved_floats(const I3dPnaModel* m, const I3dPnaBatch* b) {
    if (check_model(m, b) != I3D_OK) return -1;
    PnaCtx c;
    c.m = *m;
    c.b = *b;
    return plan_forward(c, nullptr, nullptr, nullptr);
}

extern "C" long i3d_pna_model_scratch_floats(const I3dPnaModel* m, const I3dPnaBatch* b) {
    if (check_model(m, b) != I3D_OK) return -1;
    // mirrors the takes of i3d_pna_model_bwd
    const long N = b->num_nodes, E = b->num_edges, B = b->num_graphs, F = m->hidden;
    const long top = 2 * al4(N * F) + al4((long)b->n_comb * F);
    const long head = head_floats(*m, B) + head_colsum_floats(*m, B);
    long layer = 0;
    for (int l = 0; l < m->n_layers; ++l) {
        const long f_msg = m->pre[l][m->n_pre - 1].f_out, A = m->n_aggregators * f_msg, Fo0 = m->pre[l][0].f_out;
        long t = al4(N * F) + al4((long)b->n_groups * F * A) + al4(N * A) + al4(E * f_msg) + al4(i3d_bn_bias_partial_floats(F));
        for (int i = 1; i < m->n_pre; ++i)
            t += al4(E * (long)m->pre[l][i].f_out) + al4(E * (long)m->pre[l][i].f_in) + al4(i3d_bn_bias_partial_floats(m->pre[l][i].f_out));
        t += al4(E * Fo0) + al4(N * 2 * Fo0) + al4(N * F) + al4((long)b->v_pad * Fo0) + al4(i3d_bn_bias_partial_floats(Fo0));
        t += al4(N * (2 * Fo0 + F));
        layer = std::max(layer, t);
    }
    long oa = 0, ob = 0;
    for (int k = 0; k < m->n_atom_tables; ++k) oa += m->atom_dims[k];
    for (int k = 0; k < m->n_bond_tables; ++k) ob += m->bond_dims[k];
    const long emb = al4(N * ((oa + 31) / 32 * 32)) + al4((long)b->n_comb * ((ob + 31) / 32 * 32)) + al4(((oa + 3) / 4 * 4) * F);
    long side = 0;          // generous: the per-layer sets are taken in addition to the shared region
    for (int l = 0; l < m->n_layers; ++l) side += side_floats(*m, *b, l);
    return top + side + head + std::max(layer, emb);
}

extern "C" int i3d_pna_model_fwd(const I3dPnaModel* m, const I3dPnaBatch* b, float* saved, float* node_emb, float* edge_emb,
                                 float* out, void* bn_workspace, void* const* agg_events, void* stream, void** ctx_out) {
    TRY(check_model(m, b));
    I3D_CHECK_ARG(saved != nullptr && node_emb != nullptr && out != nullptr && bn_workspace != nullptr && ctx_out != nullptr,
                  "null buffer");
    PnaCtx* c = new (std::nothrow) PnaCtx();
    I3D_CHECK_ARG(c != nullptr, "out of host memory");
    c->m = *m;
    c->b = *b;
    plan_forward(*c, saved, node_emb, out);
    *ctx_out = c;
    const int N = b->num_nodes, E = b->num_edges, B = b->num_graphs, F = m->hidden, L = m->n_layers;
    // ---- encoders (reference commons/mol_encoder.py:34-42, 65-73; models/pna.py:162-163)
    TRY(i3d_embedding_sum_fwd(b->atom_feat, nullptr, N, m->n_atom_tables, m->atom_tables, F, c->h[0], stream));
    TRY(i3d_embedding_sum_fwd(b->comb, nullptr, b->n_comb, m->n_bond_tables, m->bond_tables, F, c->bond_table, stream));
    int strides[8], s = 1;
    for (int k = 0; k < m->n_bond_tables; ++k) { strides[k] = s; s *= m->bond_dims[k]; }
    TRY(i3d_edge_codes(b->bond_feat, b->perm, E, m->n_bond_tables, strides, b->v_pad, c->codes, c->onehot, stream));
    // ---- message passing layers.  Q = bond table x W_q^T and W_D = sum_s coef W_s of a layer depend on parameters (and the
    // bond table) only: those of the layers after the first go to the side stream now (idle in the forward pass) and are
    // awaited before layer 1 - 14 us of small launches per layer off the main chain
    bool hoisted = false;
    if (L > 1 && hoist_weights()) {
        void* side = nullptr;
        TRY(i3d_wgrad_stream_fork(stream, &side));
        if (side != stream) {
            for (int l = 1; l < L; ++l) TRY(i3d_pna_layer_weights_fwd(&c->layers[l], side));
            // reference side effect (models/pna.py:163): edata['feat'] becomes the float bond embedding, edge-id order - nothing
            // in the step reads it: a leaf, off the chain (it was one more launch in front of the readout)
            if (edge_emb != nullptr)
                TRY(i3d_embedding_sum_fwd(b->bond_feat, nullptr, E, m->n_bond_tables, m->bond_tables, F, edge_emb, side));
            TRY(encoder_multihot(*c, c->hot_atoms, c->hot_bonds, side));      // needed by the backward pass only
            c->hot_ready = true;
            hoisted = true;
        }
    }
    const int eval_mode = m->training ? 0 : 1;
    if (eval_mode) {      // every BatchNorm in front of a consumer that applies it on load: its affine vector from the running statistics
        I3dBnEvalAff ea[64];
        int n = 0;
        for (int l = 0; l < L; ++l)
            for (int i = 0; i < m->n_pre; ++i) {
                const I3dFcParams& p = m->pre[l][i];
                I3D_CHECK_ARG(n < 64, "eval mode: at most 64 pretrans BatchNorms");
                ea[n++] = I3dBnEvalAff{p.running_mean, p.running_var, p.gamma, p.beta, c->layers[l].aff[i], p.f_out, p.eps};
            }
        TRY(i3d_bn_eval_aff_multi(ea, n, stream));
    }
    for (int l = 0; l < L; ++l) {
        I3dPnaLayerArgs& a = c->layers[l];
        a.eval_mode = eval_mode;
        a.msg_bf16 = msg_bf16_storage(*m, E) ? 1 : 0;
        a.weights_ready = (hoisted && l >= 1) ? 1 : 0;
        if (hoisted && l == 1) TRY(i3d_wgrad_stream_join(stream));
        set_ws(a.edge.tail, bn_workspace, nullptr, 0);
        for (int i = 0; i < a.n_pre_extra; ++i) set_ws(a.pre[i].tail, bn_workspace, nullptr, 0);
        set_ws(a.post.tail, bn_workspace, nullptr, 0);
        a.agg_event_start = agg_events ? agg_events[2 * l] : nullptr;
        a.agg_event_stop = agg_events ? agg_events[2 * l + 1] : nullptr;
        TRY(i3d_pna_layer_fwd(&a, stream));
        a.agg_event_start = a.agg_event_stop = nullptr;
    }
    if (edge_emb != nullptr && !hoisted)
        TRY(i3d_embedding_sum_fwd(b->bond_feat, nullptr, E, m->n_bond_tables, m->bond_tables, F, edge_emb, stream));
    // ---- readout + head (models/pna.py:133-134, 127-129)
    TRY(i3d_segment_readout_fwd(c->h[L], b->graph_ptr, B, F, m->readout_ops, m->n_readout, c->readout, stream));
    for (int i = 0; i < m->n_head; ++i) {
        I3dFcArgs& fc = c->head_args[i];
        const I3dFcParams& p = m->head[i];
        if (p.gamma != nullptr && eval_mode) {
            float* lin = fc.pre_keep ? fc.pre_keep : fc.xact;
            TRY(i3d_gemm_f32(0, 1, B, p.f_out, p.f_in, fc.x, p.f_in, p.W, p.f_in, lin, p.f_out, p.bias, 0, stream));
            if (p.act != I3D_ACT_NONE) TRY(i3d_act_fwd(lin, (long)B * p.f_out, p.act, fc.xact, stream));
            TRY(i3d_bn_eval_fwd(fc.xact, B, p.f_out, p.running_mean, p.running_var, p.eps, p.gamma, p.beta, I3D_ACT_NONE, nullptr,
                                fc.y, stream));
        } else if (p.gamma != nullptr) {
            set_ws(fc.tail, bn_workspace, nullptr, 0);
            TRY(i3d_fc_bn_fwd(&fc, stream));
        } else if (p.act != I3D_ACT_NONE) {
            TRY(i3d_gemm_f32(0, 1, B, p.f_out, p.f_in, fc.x, p.f_in, p.W, p.f_in, fc.xact, p.f_out, p.bias, 0, stream));
            TRY(i3d_act_fwd(fc.xact, (long)B * p.f_out, p.act, fc.y, stream));
        } else {
            TRY(i3d_gemm_f32(0, 1, B, p.f_out, p.f_in, fc.x, p.f_in, p.W, p.f_in, fc.y, p.f_out, p.bias, 0, stream));
        }
    }
    return I3D_OK;
}

extern "C" int i3d_pna_model_bwd(void* ctx, const I3dPnaModel* grads_from, const float* grad_out, float* scratch,
                                 void* bn_workspace, void* gemm_workspace, long gemm_workspace_bytes, void* stream) {
    return i3d_pna_model_bwd_part(ctx, grads_from, grad_out, scratch, bn_workspace, gemm_workspace, gemm_workspace_bytes, 0, 0,
                                  stream);
}

// part 0: the whole backward pass.  Data parallel, to overlap the gradient all-reduce with the rest of the backward pass:
// part 1 = head + layers [split, L) and a join of the weight-gradient stream (their parameter gradients are final when the
// call returns, in stream order); part 2 = layers [0, split) + encoders (same ctx, same scratch: the state in between
// - dL/dh, the bond-table gradient - lives in the scratch).
extern "C" int i3d_pna_model_bwd_part(void* ctx, const I3dPnaModel* grads_from, const float* grad_out, float* scratch,
                                      void* bn_workspace, void* gemm_workspace, long gemm_workspace_bytes, int part, int split,
                                      void* stream) {
    I3D_CHECK_ARG(ctx != nullptr && grad_out != nullptr && scratch != nullptr && bn_workspace != nullptr, "null");
    PnaCtx* c = static_cast<PnaCtx*>(ctx);
    if (grads_from != nullptr) {       // the gradient buffers are chosen at backward time: take them (same model otherwise)
        I3D_CHECK_ARG(grads_from->n_layers == c->m.n_layers && grads_from->n_pre == c->m.n_pre &&
                          grads_from->n_head == c->m.n_head && grads_from->hidden == c->m.hidden, "different model");
        c->m = *grads_from;
    }
    const I3dPnaModel& m = c->m;
    I3D_CHECK_ARG(m.grad_atom_tables != nullptr && m.grad_bond_tables != nullptr && m.post[0].grad_W != nullptr,
                  "gradient buffers not set");
    const I3dPnaBatch& b = c->b;
    const int N = b.num_nodes, E = b.num_edges, B = b.num_graphs, F = m.hidden, L = m.n_layers;
    I3D_CHECK_ARG(part >= 0 && part <= 2 && (part == 0 || (split > 0 && split < L)), "bad part / split");
    const int l_hi = part == 2 ? split : L, l_lo = part == 1 ? split : 0;      // layers [l_lo, l_hi) in this call
    Bump top(scratch);
    float* gh[2] = {top.take((long)N * F), top.take((long)N * F)};      // dL/dh, ping-pong between layers
    float* grad_table = top.take((long)b.n_comb * F);                  // dL/d(bond table), summed over the layers
    // buffers the weight-gradient stream reads or writes: one set per layer unless every layer joins that stream
    const bool per_layer_join = join_per_layer();
    // the last 16 MB of the weight-gradient scratch belong to the launches the MAIN stream issues while the side stream is
    // busy (the atom tables' gradient at the end); the side stream's launches see the front part only
    const long tail_bytes = (!per_layer_join && gemm_workspace != nullptr && gemm_workspace_bytes >= (64L << 20)) ? (16L << 20) : 0;
    char* const tail_ws = tail_bytes > 0 ? (char*)gemm_workspace + (gemm_workspace_bytes - tail_bytes) / 256 * 256 : nullptr;
    const long side_ws_bytes = tail_bytes > 0 ? (long)(tail_ws - (char*)gemm_workspace) : gemm_workspace_bytes;
    std::vector<float*> side(L, nullptr);
    if (!per_layer_join)
        for (int l = 0; l < L; ++l) side[l] = top.take(side_floats(m, b, l));
    float* const head_buf = top.take(head_floats(m, B));
    float* const head_colsum_ws = top.take(head_colsum_floats(m, B));
    float* const rest = scratch + top.used;
    // the head's weight / bias gradients (leaves); only invoked inside this call: a plain lambda over the locals (a
    // std::function with by-value captures copied the whole model description and heap-allocated per backward call)
    const float* gpre_of[I3D_MAX_HEAD_FC] = {};
    auto head_leaves = [&](void* wst) -> int {
        for (int i = m.n_head - 1; i >= 0; --i) {
            I3dFcArgs& fc = c->head_args[i];
            const I3dFcParams& p = m.head[i];
            if (p.gamma != nullptr) {
                TRY(i3d_fc_bn_bwd_wgrad(&fc, wst));
            } else {
                TRY(wgrad(p.f_out, p.f_in, B, gpre_of[i], p.f_out, fc.x, p.f_in, p.grad_W, p.f_in, gemm_workspace, side_ws_bytes, wst));
                void* cws = bn_workspace;
                if (wst != stream) {      // a reduction scratch of the side stream's own; its arrival counters (the first bytes) start at 0
                    cws = head_colsum_ws;
                    if (hipMemsetAsync(cws, 0, 256, (hipStream_t)wst) != hipSuccess) {
                        i3d::set_error("i3d_pna_model_bwd: clearing the head's reduction scratch failed");
                        return I3D_ERR_LAUNCH;
                    }
                }
                TRY(i3d_colsum(gpre_of[i], nullptr, B, p.f_out, p.grad_bias, cws, wst));
            }
        }
        return I3D_OK;
    };
    bool leaves_pending = false;
    // ---- head, last block first: the chain (BatchNorm backward, data gradients) on the caller's stream, then ONE fork and the
    // head's weight and bias gradients - leaves - on the weight-gradient stream, next to the readout's and the last layer's
    // chain (round 2 ran them in line: three launches of latency on the critical path).  Their inputs live in `head_buf`,
    // outside the region the layers reuse.
    if (part != 2) {
        Bump ar(head_buf);
        const float* gy = grad_out;
        for (int i = m.n_head - 1; i >= 0; --i) {
            I3dFcArgs& fc = c->head_args[i];
            const I3dFcParams& p = m.head[i];
            float* gx = ar.take((long)B * p.f_in);
            if (p.gamma != nullptr) {
                set_ws(fc.tail, bn_workspace, gemm_workspace, side_ws_bytes);
                fc.grad_y = gy; fc.grad_pre = ar.take((long)B * p.f_out); fc.grad_x = gx;
                fc.grad_W = p.grad_W; fc.grad_bias = p.grad_bias; fc.grad_gamma = p.grad_gamma; fc.grad_beta = p.grad_beta;
                TRY(i3d_fc_bn_bwd_chain(&fc, stream));
            } else {
                const float* gpre = gy;
                if (p.act != I3D_ACT_NONE) {
                    float* t = ar.take((long)B * p.f_out);
                    TRY(i3d_act_bwd(gy, fc.xact, (long)B * p.f_out, p.act, t, stream));
                    gpre = t;
                }
                TRY(i3d_gemm_f32(0, 0, B, p.f_in, p.f_out, gpre, p.f_out, p.W, p.f_in, gx, p.f_in, nullptr, 0, stream));
                gpre_of[i] = gpre;
            }
            gy = gx;
        }
        constexpr bool leaves_aside = true;
        // the leaves go to the weight-gradient stream behind the fork the LAST layer's backward makes anyway (a fork of their own
        // was one more event on the chain's stream: ~7 us); without a side stream for them: here, in line
        leaves_pending = !per_layer_join && leaves_aside && l_hi == L && l_hi > l_lo;
        if (!leaves_pending) TRY(head_leaves(stream));
        // readout backward -> dL/dh_L
        TRY(i3d_segment_readout_bwd(gy, c->h[L], b.graph_ptr, B, F, m.readout_ops, m.n_readout, gh[L & 1], stream));
        c->gh_cur = L & 1;
    }
    // ---- layers, last first
    for (int l = l_hi - 1; l >= l_lo; --l) {
        I3dPnaLayerArgs& a = c->layers[l];
        Bump ar(rest);
        Bump own(side[l]);
        Bump& sd = per_layer_join ? ar : own;       // where the side stream's buffers of this layer live
        a.defer_join = per_layer_join ? 0 : 1;
        a.wgrad_split = (split_wgrad() == 2 || (l == 0 && split_wgrad() == 1)) ? 1 : 0;
        // a residual layer accumulates dL/dh_in on top of the incoming gradient IN PLACE (dh_in = dh_out + ...: the separate
        // add pass over [N, F] is gone, composite.hip: i3d_pna_layer_bwd); others ping-pong between the two buffers
        const int cur = c->gh_cur, nxt = a.residual ? cur : cur ^ 1;
        const float* grad_in = gh[cur];
        a.grad_out = grad_in;
        I3dGroupedFcArgs& g = a.post;
        const I3dFcParams& pp = m.post[l];
        set_ws(g.tail, bn_workspace, gemm_workspace, side_ws_bytes);
        g.grad_W = pp.grad_W; g.grad_bias = pp.grad_bias; g.grad_gamma = pp.grad_gamma; g.grad_beta = pp.grad_beta;
        g.grad_y = grad_in;
        g.grad_pre = sd.take((long)N * F);
        g.grad_WD = sd.take((long)b.n_groups * F * g.agg_width);
        g.grad_h = gh[nxt];
        g.grad_agg = ar.take((long)N * g.agg_width);
        g.tail.bias_partial = defer_bias() ? sd.take(i3d_bn_bias_partial_floats(F)) : nullptr;
        const int f_msg = a.n_pre_extra > 0 ? a.pre[a.n_pre_extra - 1].f_out : a.edge.f_out;
        a.grad_msg = ar.take((long)E * f_msg);
        const float* gy = a.grad_msg;
        for (int i = a.n_pre_extra - 1; i >= 0; --i) {
            I3dFcArgs& fc = a.pre[i];
            const I3dFcParams& p = m.pre[l][i + 1];
            set_ws(fc.tail, bn_workspace, gemm_workspace, side_ws_bytes);
            fc.grad_W = p.grad_W; fc.grad_bias = p.grad_bias; fc.grad_gamma = p.grad_gamma; fc.grad_beta = p.grad_beta;
            fc.grad_y = gy;
            fc.grad_pre = sd.take((long)E * fc.f_out);
            fc.grad_x = ar.take((long)E * fc.f_in);
            fc.tail.bias_partial = defer_bias() ? sd.take(i3d_bn_bias_partial_floats(fc.f_out)) : nullptr;
            gy = fc.grad_x;
        }
        I3dEdgeFcArgs& e = a.edge;
        const I3dFcParams& p0 = m.pre[l][0];
        set_ws(e.tail, bn_workspace, gemm_workspace, side_ws_bytes);
        e.grad_W = p0.grad_W; e.grad_bias = p0.grad_bias; e.grad_gamma = p0.grad_gamma; e.grad_beta = p0.grad_beta;
        e.grad_y = gy;
        e.grad_pre = sd.take((long)E * e.f_out);
        e.grad_P = sd.take((long)N * 2 * e.f_out);
        a.DL = a.merge_h ? sd.take((long)N * (2 * e.f_out + F)) : nullptr;
        e.grad_h = ar.take((long)N * F);
        e.grad_Q = sd.take((long)b.v_pad * e.f_out);
        e.tail.bias_partial = defer_bias() ? sd.take(i3d_bn_bias_partial_floats(e.f_out)) : nullptr;
        // (the buffer side_floats() counts for the edge block's bias partials: taken by exactly one of the two)
        a.edge_bias_partial = (a.merge_h && !defer_bias()) ? sd.take(i3d_bn_bias_partial_floats(e.f_out)) : nullptr;
        e.grad_q = grad_table;
        e.grad_q_accumulate = (l == L - 1) ? 0 : 1;        // the bond table feeds every layer: its gradient is their sum
        TRY(i3d_pna_layer_bwd(&a, stream));
        c->gh_cur = nxt;
        if (leaves_pending) {
            void* wst = stream;
            TRY(i3d_wgrad_stream_peek(stream, &wst));
            TRY(head_leaves(wst));
            leaves_pending = false;
        }
    }
    if (part == 1) {
        if (!per_layer_join) TRY(i3d_wgrad_stream_join(stream));
        return I3D_OK;
    }
    // ---- encoders: embedding-table gradients as multi-hot^T dY (deterministic, csrc/edge.hip: multihot_kernel).  The atom
    // tables' needs dL/dh_0 only - the chain's own result - so it runs BEFORE the join, next to the first layer's weight
    // gradients (round 2: after it, 30 us at the very end of the step); its split-K slices go through a slice of the scratch
    // of its own (the side stream's launches own the front of it).  The bond tables' needs the side stream's grad_table.
    {
        Bump ar(rest);
        int oa = 0, ob = 0;
        for (int k = 0; k < m.n_atom_tables; ++k) oa += m.atom_dims[k];
        for (int k = 0; k < m.n_bond_tables; ++k) ob += m.bond_dims[k];
        const int va = (oa + 31) / 32 * 32, vb = (ob + 31) / 32 * 32;
        float* hot = c->hot_atoms;
        float* hotb = c->hot_bonds;
        if (!c->hot_ready) {            // not built during the forward pass (no side stream): here, into scratch
            hot = ar.take((long)N * va);
            hotb = ar.take((long)b.n_comb * vb);
            TRY(encoder_multihot(*c, hot, hotb, stream));
        }
        const bool own_ws = tail_bytes > 0;
        // 173 atom categories: not a multiple of 4, and a product whose M is not takes the GEMM's 4-byte-load variant (66 us
        // against 12 us).  The multi-hot matrix has zero columns up to va, so the product is taken over oa rounded up to 4 into
        // scratch and the real rows are copied out (same tiles, same split, same summation order: same bits)
        const int oa4 = (oa + 3) / 4 * 4;
        float* atoms_out = oa4 == oa ? m.grad_atom_tables : ar.take((long)oa4 * F);
        auto atom_tables = [&](void* ws, long wsb) -> int {
            TRY(wgrad(oa4, F, N, hot, va, gh[c->gh_cur], F, atoms_out, F, ws, wsb, stream));
            if (atoms_out != m.grad_atom_tables &&
                hipMemcpyAsync(m.grad_atom_tables, atoms_out, (size_t)oa * F * sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream) != hipSuccess) {
                i3d::set_error("i3d_pna_model_bwd: copy of the atom-table gradient failed");
                return I3D_ERR_LAUNCH;
            }
            return I3D_OK;
        };
        if (own_ws) TRY(atom_tables(tail_ws, tail_bytes - 256));
        // the bond tables' product reads the sum the weight-gradient stream has just finished: on THAT stream, in front of the
        // join (behind it, it was one more cross-stream hop - ~10 us - between the last panel reduction and Adam)
        void* bst = stream;
        constexpr bool bond_aside = true;
        if (bond_aside && !per_layer_join && own_ws && c->hot_ready) TRY(i3d_wgrad_stream_peek(stream, &bst));
        if (bst != stream)
            TRY(wgrad(ob, F, b.n_comb, hotb, vb, grad_table, F, m.grad_bond_tables, F, gemm_workspace, side_ws_bytes, bst));
        if (!per_layer_join) TRY(i3d_wgrad_stream_join(stream));
        if (!own_ws) TRY(atom_tables(gemm_workspace, gemm_workspace_bytes));
        if (bst == stream)
            TRY(wgrad(ob, F, b.n_comb, hotb, vb, grad_table, F, m.grad_bond_tables, F, gemm_workspace, gemm_workspace_bytes, stream));
    }
    return I3D_OK;
}

// test entry: the messages of layer `layer` as its aggregation kernels read them (saved activation of the last pretrans
// block with its BatchNorm applied on load), [E, f_msg] in destination-sorted order
extern "C" int i3d_pna_model_debug_messages(void* ctx, int layer, float* out, void* stream) {
    I3D_CHECK_ARG(ctx != nullptr && out != nullptr, "null");
    PnaCtx* c = static_cast<PnaCtx*>(ctx);
    I3D_CHECK_ARG(layer >= 0 && layer < c->m.n_layers, "layer out of range");
    const I3dPnaLayerArgs& a = c->layers[layer];
    const int f_msg = a.n_pre_extra > 0 ? a.pre[a.n_pre_extra - 1].f_out : a.edge.f_out;
    return i3d_pna_messages_normalized_ex(a.msg, a.fused_bn && a.msg_bf16 && a.n_pre_extra > 0, a.aff[a.n_pre_extra], a.edge.num_edges, f_msg, out, stream);
}

extern "C" int i3d_pna_model_ctx_free(void* ctx) {
    delete static_cast<PnaCtx*>(ctx);
    return I3D_OK;
}
