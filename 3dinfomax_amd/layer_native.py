"""One PNA layer through the native layer composite (csrc/composite.hip: i3d_pna_layer_fwd / i3d_pna_layer_bwd).

pna.PNALayerFn runs a layer as four block Functions back to back from Python (~20 us of Python and 5-7 allocations per
block and direction).  Here the whole layer is ONE C call per direction: Python sizes one scratch buffer, fills one
argument struct (pointers into that buffer, chained block to block) and allocates only what autograd hands out (the
layer output, dL/dh, dL/dq and the parameter gradients).  Same kernels, same order, same bits as the block path - it is
eligible when every block is a BatchNorm block in training mode with local statistics (the pre-training configuration)
and the posttrans is degree-grouped; anything else stays on the block path.
"""
import ctypes

import torch

import os

from . import _lib, ops, tape
from . import layers as _layers
from .layers import _keeps_pre, _set_workspaces

# I3D_NATIVE_LAYER=0: the layer as four block composites sequenced from Python (pna.PNALayerFn)
NATIVE_LAYER = True
# I3D_FUSED_BN=0: round-1 form of the layer (statistics pass + apply pass per block); default: BatchNorm statistics in the
# producers' epilogues, BatchNorm-apply in the consumers' loads (csrc/fused_bn.hip).  Same arithmetic up to summation order.
FUSED_BN = True
# I3D_MERGE_H=0: the products that read the node features as separate GEMMs (csrc/model.hip reads the same switch)
MERGE_H = True
_SIMPLE_ACTS = (None, 'relu', 'leakyrelu')
KEEP_LAST_ARGS = None


def fused_bn_ok(plan, params, n_pre, n_post):
    if not FUSED_BN or n_post != 1:
        return False
    for spec in plan.pre_specs + plan.post_specs[:1]:
        if spec.act not in _SIMPLE_ACTS:
            return False
    for spec in plan.pre_specs:
        if spec.post_act is not None:
            return False
    for i in range(1, n_pre):          # BatchNorm prologue of the GEMM: K <= 1024
        if params[4 * i].shape[1] > 1024:
            return False
    return True


def _al(n):
    return (n + 3) & ~3


class _Arena:
    """bump allocator over one float32 tensor: take(n) -> device pointer (16-byte aligned)"""

    def __init__(self, nfloats, device):
        self.buf = torch.empty(nfloats, dtype=torch.float32, device=device)
        self.base = self.buf.data_ptr()
        self.cap = nfloats
        self.used = 0

    def take(self, n):
        p = self.base + 4 * self.used
        self.used += _al(n)
        assert self.used <= self.cap
        return p


def eligible(h, q, index, qmap, plan, params):
    if not (NATIVE_LAYER and plan.grouped and h.is_cuda and index.num_edges > 0):
        return False
    n_pre, n_post = len(plan.pre_specs), len(plan.post_specs)
    if n_pre - 1 > 3 or n_post - 1 > 3 or h.shape[1] % 4:
        return False
    for spec in plan.pre_specs + plan.post_specs:
        if not _layers._composite_ok(spec, h):
            return False
    for i in range(n_pre + n_post):
        W = params[4 * i]
        if not W.is_contiguous() or W.shape[0] % 4 or W.shape[1] % 4:
            return False
    groups = index.degree_groups()[2]
    if len(groups) > 32 or len(groups) * len(plan.coef[0]) > 128:
        return False
    if q is not None and (not q.is_contiguous() or (qmap is not None and q.shape[0] != qmap.rows)):
        return False
    return True


def _tail(tail, spec, gamma, beta, mean_ptr, invstd_ptr, feat, device):
    bn = spec.bn
    tail.act, tail.post_act = ops.ACT[spec.act], ops.ACT[spec.post_act]
    tail.eps, tail.momentum = bn.eps, bn.momentum
    tail.gamma, tail.beta = gamma.data_ptr(), beta.data_ptr()
    tail.running_mean, tail.running_var = bn.running_mean.data_ptr(), bn.running_var.data_ptr()
    tail.mean, tail.invstd = mean_ptr, invstd_ptr
    nbt = bn.num_batches_tracked           # bumped by the statistics kernel
    tail.num_batches_tracked = nbt.data_ptr() if nbt is not None else None
    _set_workspaces(tail, feat, device)


def forward(ctx, h, q, index, qmap, plan, params):
    h = h.contiguous()
    dev = h.device
    N, Fh = h.shape
    E = index.num_edges
    n_pre, n_post = len(plan.pre_specs), len(plan.post_specs)
    pre_p = [params[4 * i:4 * i + 4] for i in range(n_pre)]
    post_p = [params[4 * (n_pre + i):4 * (n_pre + i) + 4] for i in range(n_post)]
    rows_d, tiles_d, groups = index.degree_groups()
    nG, nS = len(groups), len(plan.coef[0])
    Fo0 = pre_p[0][0].shape[0]
    Fq = q.shape[1] if q is not None else 0
    q_rows = qmap.rows if (qmap is not None and q is not None) else 0
    Fmsg = pre_p[-1][0].shape[0]
    A = len(plan.aggregators) * Fmsg
    Fp0 = post_p[0][0].shape[0]

    fused = fused_bn_ok(plan, params, n_pre, n_post)
    L = _lib.load()
    # ---- size of the saved/scratch buffer
    total = _al(N * 2 * Fo0) + (_al((q_rows or E) * Fo0) if q is not None else 0)
    for i, spec in enumerate(plan.pre_specs):
        Fo = pre_p[i][0].shape[0]
        total += _al(E * Fo) * ((1 if fused else 2) + (1 if _keeps_pre(spec) else 0)) + 2 * _al(Fo) + (_al(3 * Fo) if fused else 0)
    n_stats = 0
    if fused:
        f_max = max([pp[0].shape[0] for pp in pre_p] + [Fp0])
        n_stats = int(L.i3d_pna_layer_stats_floats(N, E, rows_d.shape[0], f_max))
        total += _al(n_stats)
    total += _al(N * A) + _al(nG * Fp0 * A)
    merged = fused and n_post == 1 and MERGE_H and Fp0 % 4 == 0          # csrc/composite.hip: merge_h_ok
    WL = 2 * Fo0 + Fp0
    if merged:
        total += _al(WL * Fh) + _al(WL) + _al(N * WL)
        # the packed weights of the row-panel products (csrc/panel.hip): Wcat forward, the later pretrans blocks' data gradients
        total += _al(L.i3d_panel_packed_bytes(WL, Fh) // 4) + _al(L.i3d_panel_packed_bytes(Fh, WL) // 4)
        f_prev = Fo0
        for i in range(1, n_pre):
            total += _al(L.i3d_panel_packed_bytes(f_prev, pre_p[i][0].shape[0]) // 4)
            total += _al(L.i3d_panel_packed_bytes(pre_p[i][0].shape[0], f_prev) // 4)
            f_prev = pre_p[i][0].shape[0]
    for i, spec in enumerate(plan.post_specs):
        Fo = post_p[i][0].shape[0]
        total += _al(N * Fo) * (2 + (1 if _keeps_pre(spec) else 0)) + 2 * _al(Fo)
    ar = _Arena(total, dev)
    Fout = post_p[-1][0].shape[0]
    y_out = torch.empty(N, Fout, dtype=torch.float32, device=dev)

    a = _lib.PnaLayerArgs()
    if fused:
        a.fused_bn = 1
        a.stats_ws = ar.take(n_stats)
    # ---- pretrans block 0: edge gather-combine
    e = a.edge
    W, b, ga, be = pre_p[0]
    spec = plan.pre_specs[0]
    _tail(e.tail, spec, ga, be, ar.take(Fo0), ar.take(Fo0), Fo0, dev)
    e.num_nodes, e.num_edges, e.f_h, e.f_q, e.f_out, e.ldw = N, E, Fh, Fq, Fo0, W.stride(0)
    e.h, e.W, e.bias = h.data_ptr(), W.data_ptr(), b.data_ptr()
    if q is not None:
        e.q = q.data_ptr()
        if q_rows:
            e.q_rows, e.v_pad, e.q_code, e.onehot = q_rows, qmap.v_pad, qmap.codes.data_ptr(), qmap.onehot.data_ptr()
        e.Q = ar.take((q_rows or E) * Fo0)
    e.src_s, e.dst_s, e.in_ptr = index.src_s.data_ptr(), index.dst_s.data_ptr(), index.in_ptr.data_ptr()
    e.out_ptr, e.out_epos = index.out_ptr.data_ptr(), index.out_epos.data_ptr()
    e.P = ar.take(N * 2 * Fo0)
    if merged:          # the products that read h as one GEMM per direction (include/infomax3d_hip.h: I3dPnaLayerArgs.merge_h)
        a.merge_h = 1
        a.Wcat, a.bcat, a.PL = ar.take(WL * Fh), ar.take(WL), ar.take(N * WL)
        a.Wcat_panel = ar.take(L.i3d_panel_packed_bytes(WL, Fh) // 4)
        a.Wcat_dgrad_panel = ar.take(L.i3d_panel_packed_bytes(Fh, WL) // 4)
    e.xact = ar.take(E * Fo0)
    if _keeps_pre(spec):
        e.pre_keep = ar.take(E * Fo0)
    if fused:              # the normalised activation is never written: the consumer applies aff while it loads xact
        a.aff[0] = ar.take(3 * Fo0)
        x_ptr, f_in = e.xact, Fo0
    else:
        e.y = ar.take(E * Fo0)
        x_ptr, f_in = e.y, Fo0
    # ---- further pretrans blocks
    a.n_pre_extra = n_pre - 1
    for i in range(1, n_pre):
        W, b, ga, be = pre_p[i]
        spec = plan.pre_specs[i]
        Fo = W.shape[0]
        c = a.pre[i - 1]
        _tail(c.tail, spec, ga, be, ar.take(Fo), ar.take(Fo), Fo, dev)
        c.rows, c.f_in, c.f_out, c.ldw = E, f_in, Fo, W.stride(0)
        c.x, c.W, c.bias = x_ptr, W.data_ptr(), b.data_ptr()
        c.xact = ar.take(E * Fo)
        if merged:
            c.W_dgrad_panel = ar.take(L.i3d_panel_packed_bytes(f_in, Fo) // 4)
            c.W_fwd_panel = ar.take(L.i3d_panel_packed_bytes(Fo, f_in) // 4)
        if _keeps_pre(spec):
            c.pre_keep = ar.take(E * Fo)
        if fused:
            a.aff[i] = ar.take(3 * Fo)
            x_ptr, f_in = c.xact, Fo
        else:
            c.y = ar.take(E * Fo)
            x_ptr, f_in = c.y, Fo
    # ---- aggregation
    a.n_aggregators, a.n_scalers, a.force_scalers, a.avg_d_log = len(plan.aggregators), len(plan.agg_scalers), 0, plan.avg
    for i, v in enumerate(plan.aggregators):
        a.aggregators[i] = v
    for i, v in enumerate(plan.agg_scalers):
        a.scalers[i] = v
    a.msg = x_ptr
    # ---- posttrans block 0: degree-grouped
    p = a.post
    W, b, ga, be = post_p[0]
    spec = plan.post_specs[0]
    _tail(p.tail, spec, ga, be, ar.take(Fp0), ar.take(Fp0), Fp0, dev)
    p.num_nodes, p.f_h, p.f_out, p.agg_width, p.ldw = N, Fh, Fp0, A, W.stride(0)
    p.n_groups, p.n_scalers, p.m_padded = nG, nS, rows_d.shape[0]
    for gi, (_, start, count) in enumerate(groups):
        p.group_start[gi], p.group_count[gi] = start, count
    k = 0
    for g_ in plan.coef:
        for v in g_:
            p.coef[k] = v
            k += 1
    p.h, p.W, p.bias = h.data_ptr(), W.data_ptr(), b.data_ptr()
    p.agg = ar.take(N * A)
    p.deg_rows, p.deg_tile_group = rows_d.data_ptr(), tiles_d.data_ptr()
    p.WD = ar.take(nG * Fp0 * A)
    p.xact = ar.take(N * Fp0)
    if _keeps_pre(spec):
        p.pre_keep = ar.take(N * Fp0)
    a.n_post_extra = n_post - 1
    a.residual = 1 if plan.residual else 0
    if n_post == 1:
        p.y = y_out.data_ptr()
        if plan.residual:
            p.residual = h.data_ptr()
    else:
        p.y = ar.take(N * Fp0)
    x_ptr, f_in = p.y, Fp0
    for i in range(1, n_post):
        W, b, ga, be = post_p[i]
        spec = plan.post_specs[i]
        Fo = W.shape[0]
        c = a.postx[i - 1]
        _tail(c.tail, spec, ga, be, ar.take(Fo), ar.take(Fo), Fo, dev)
        c.rows, c.f_in, c.f_out, c.ldw = N, f_in, Fo, W.stride(0)
        c.x, c.W, c.bias = x_ptr, W.data_ptr(), b.data_ptr()
        c.xact = ar.take(N * Fo)
        if _keeps_pre(spec):
            c.pre_keep = ar.take(N * Fo)
        last = i == n_post - 1
        c.y = y_out.data_ptr() if last else ar.take(N * Fo)
        if last and plan.residual:
            c.residual = h.data_ptr()
        x_ptr, f_in = c.y, Fo
    if ops.KERNEL_TIMERS is not None:   # bench.py: HIP events on the launch stream around the roofline kernel
        t0, t1 = ops.RawEvent(), ops.RawEvent()
        a.agg_event_start, a.agg_event_stop = t0.handle, t1.handle
        ops.KERNEL_TIMERS.setdefault('pna_aggregate_fwd', []).append(
            (t0, t1, N, E, Fmsg, len(plan.aggregators) * len(plan.agg_scalers) * Fmsg))
    _lib.check(L.i3d_pna_layer_fwd(ctypes.byref(a), ops._stream()), 'i3d_pna_layer_fwd')
    a.agg_event_start = a.agg_event_stop = None
    ctx.native = (a, ar, h, q, qmap, index, plan, params, (N, E, Fh, Fq, A, nG))
    if KEEP_LAST_ARGS is not None:      # tools/launch_floor.py: replay the C call without the Python around it
        KEEP_LAST_ARGS.append((a, ctx.native))
    return y_out


def backward(ctx, grad):
    a, ar_fwd, h, q, qmap, index, plan, params, (N, E, Fh, Fq, A, nG) = ctx.native
    dev = h.device
    grad = grad.contiguous()
    n_pre, n_post = len(plan.pre_specs), len(plan.post_specs)
    dims_pre = [params[4 * i].shape for i in range(n_pre)]
    dims_post = [params[4 * (n_pre + i)].shape for i in range(n_post)]
    Fo0, Fmsg, Fp0 = dims_pre[0][0], dims_pre[-1][0], dims_post[0][0]
    need_q = q is not None and ctx.needs_input_grad[1]
    total = 0
    for (Fo, Fi) in dims_post[1:]:
        total += _al(N * Fo) + _al(N * Fi)
    total += _al(N * Fp0) + _al(nG * Fp0 * A) + _al(N * A) + _al(E * Fmsg)
    for (Fo, Fi) in dims_pre[1:]:
        total += _al(E * Fo) + _al(E * Fi)
    total += _al(E * Fo0) + _al(N * 2 * Fo0) + _al(N * Fh) + (_al(qmap.v_pad * Fo0) if (qmap is not None and q is not None) else 0)
    if a.merge_h:
        total += _al(N * (2 * Fo0 + Fp0)) + _al(_lib.load().i3d_bn_bias_partial_floats(Fo0))
    ar = _Arena(total, dev)
    grads = []          # (gW, gbias, ggamma, gbeta) per block, forward order

    def param_grads(k):       # (W, bias, gamma, beta) of block k: the model's persistent gradient buffers when it has them
        return tuple(tape.grad_like(params[4 * k + j]) for j in range(4))

    def set_param_grads(c, g4):
        c.grad_W, c.grad_bias, c.grad_gamma, c.grad_beta = (t.data_ptr() for t in g4)

    a.grad_out = grad.data_ptr()
    # ---- posttrans, last block first
    gy = grad.data_ptr()
    post_g = [None] * n_post
    for i in range(n_post - 1, 0, -1):
        c = a.postx[i - 1]
        Fo, Fi = dims_post[i]
        _set_workspaces(c.tail, Fo, dev)
        g4 = param_grads(n_pre + i)
        post_g[i] = g4
        set_param_grads(c, g4)
        c.grad_y, c.grad_pre, c.grad_x = gy, ar.take(N * Fo), ar.take(N * Fi)
        gy = c.grad_x
    p = a.post
    _set_workspaces(p.tail, Fp0, dev)
    g4 = param_grads(n_pre)
    post_g[0] = g4
    set_param_grads(p, g4)
    gh = torch.empty(N, Fh, dtype=torch.float32, device=dev)
    p.grad_y, p.grad_pre, p.grad_WD = gy, ar.take(N * Fp0), ar.take(nG * Fp0 * A)
    p.grad_h, p.grad_agg = gh.data_ptr(), ar.take(N * A)
    a.grad_msg = ar.take(E * Fmsg)
    # ---- pretrans, last block first
    gy = a.grad_msg
    pre_g = [None] * n_pre
    for i in range(n_pre - 1, 0, -1):
        c = a.pre[i - 1]
        Fo, Fi = dims_pre[i]
        _set_workspaces(c.tail, Fo, dev)
        g4 = param_grads(i)
        pre_g[i] = g4
        set_param_grads(c, g4)
        c.grad_y, c.grad_pre, c.grad_x = gy, ar.take(E * Fo), ar.take(E * Fi)
        gy = c.grad_x
    e = a.edge
    _set_workspaces(e.tail, Fo0, dev)
    g4 = param_grads(0)
    pre_g[0] = g4
    set_param_grads(e, g4)
    gq = torch.empty_like(q) if need_q else None
    e.grad_y, e.grad_pre, e.grad_P, e.grad_h = gy, ar.take(E * Fo0), ar.take(N * 2 * Fo0), ar.take(N * Fh)
    if a.merge_h:
        a.DL = ar.take(N * (2 * Fo0 + Fp0))
        # the edge block's BatchNorm backward fused with its segmented sums (csrc/bn.hip: i3d_bn_bwd_edge_sums): partials of the
        # bias gradient's column sum on the weight-gradient stream
        a.edge_bias_partial = ar.take(_lib.load().i3d_bn_bias_partial_floats(Fo0))
    e.grad_q = gq.data_ptr() if gq is not None else None
    if qmap is not None and q is not None:
        e.grad_Q = ar.take(qmap.v_pad * Fo0)
    L = _lib.load()
    _lib.check(L.i3d_pna_layer_bwd(ctypes.byref(a), ops._stream()), 'i3d_pna_layer_bwd')
    flat = [t for g4_ in pre_g + post_g for t in g4_]
    return (gh, gq, None, None, None) + tuple(flat)
