"""Net3D forward / backward as ONE node of the model tape, sequenced over raw device pointers.

reference models/net3d.py:57-125 (Net3D.forward, Net3DLayer).  Same kernels and block composites as the per-block path
of net3d.py (`I3D_NATIVE_NET3D=0` selects that one; the tests check the two give the same bits): what changes is the
host side.  The per-block path is ~9 tape nodes per direction, each an autograd-Function body with its own tensor
allocations, context and argument marshalling (~40-50 us of Python per node on this stack, more than the ~20 launches
of Net3D cost the GPU).  Here one scratch buffer is sized per direction, the argument structs of the block composites
are filled with pointers into it, and the C entry points (include/infomax3d_hip.h) are called back to back; only what
leaves the network is a tensor (the output, the node embeddings and distances stored on the graph, parameter gradients).

Eligible: training mode with local batch statistics on every BatchNorm, `use_node_features=False` (the reference's
pre-training configs), no activation on the last output layer.
"""
import ctypes
import os

import torch

from . import _lib, ops, tape
from .layer_native import _Arena, _al, _tail
from .layers import _composite_ok, _keeps_pre, _set_workspaces

NATIVE_NET3D = True
# bf16 matmul mode: the edge stage's own [E3, H] activations (x_msg, msg) stored as bf16 once they are large enough to be bound
# by their bytes (QMugs shape, 3.9 M edges: 313 MB per tensor, step 6.71 -> 6.49 ms; at the QM9 shape's 140 k edges they sit in
# the Infinity Cache and the extra conversion work costs 1 %).  (the tests set BF16_STORE_MIN_EDGES = 0: at every size)
BF16_STORE = True
BF16_STORE_MIN_EDGES = 1 << 20
FUSED_EDGE = True      # the edge stage in one lane per edge (csrc/net3d_edge.hip)
_F32 = torch.float32


def _chk(rc, name):
    if rc != 0:
        _lib.check(rc, name)


def _blocks(model):
    """the FC layers of the network in forward order, grouped (cached on the module: the structure never changes)"""
    b = model.__dict__.get('_i3d_blocks')
    if b is None:
        layers = [(list(l.message_network.fully_connected), l.soft_edge_network, list(l.update_network.fully_connected), l.reduce_mean)
                  for l in model.mp_layers]
        nw = list(model.node_wise_output_network.fully_connected) if model.node_wise_output_layers > 0 else []
        b = model.__dict__['_i3d_blocks'] = (model.edge_input.fully_connected[0], layers, nw, list(model.output.fully_connected))
    return b


def eligible(model, g):
    if not (NATIVE_NET3D and model.training and not model.use_node_features):      # (called under an active tape only)
        return False
    idx = g.index()
    if g.device.type != 'cuda' or idx.num_edges == 0 or idx.num_graphs == 0:
        return False
    edge_in, layers, nw, out = _blocks(model)
    fcs = [edge_in] + [fc for m, _, u, _ in layers for fc in m + u] + nw + out[:-1]
    for fc in fcs:
        h = fc.hot()
        if not (_composite_ok(h[4]) and h[0].is_cuda and h[0].is_contiguous() and h[1] is not None and fc.bias):
            return False
    last = out[-1].hot()
    if last[4].bn is not None:                       # BatchNorm on the last output layer: an ordinary block
        if not (_composite_ok(last[4]) and last[0].is_contiguous()):
            return False
    elif last[4].act is not None or not last[0].is_contiguous() or last[1] is None:
        return False
    return len(layers) >= 1 and all(len(m) >= 1 and len(u) >= 1 for m, _, u, _ in layers)


def fused_edge_ok(model):
    """the structure net3d_edge.hip is written for: one propagation layer with one message block on the broadcast node
    embedding, SiLU activations, a hidden width / Fourier order the kernels are built for (cached on the module)"""
    ok = model.__dict__.get('_i3d_fused_edge')
    if ok is None:
        edge_in, layers, nw, out = _blocks(model)
        ok = False
        if len(layers) == 1 and len(layers[0][0]) == 1:
            H = model.node_embedding.shape[0]
            w_in, spec_in = edge_in.hot('silu')[0], edge_in.hot('silu')[4]
            w_m, spec_m = layers[0][0][0].hot()[0], layers[0][0][0].hot()[4]
            n_enc = model.fourier_encodings
            din = 2 * n_enc + 1 if n_enc > 0 else 1
            ok = (tuple(w_in.shape) == (H, din) and tuple(w_m.shape) == (H, 3 * H) and spec_in.act == 'silu'
                  and spec_m.act == 'silu' and spec_m.post_act is None
                  and bool(_lib.load().i3d_net3d_edge_supported(H, n_enc)))
        model.__dict__['_i3d_fused_edge'] = ok
    return ok and FUSED_EDGE


def _edge_stage_fwd(L, stream, ar, dev, model, g, idx, d_raw, edge_in, msg_fc, se, reduce_mean, N, E, H):
    """the fused edge stage: -> (argument struct, m_sum pointer, d_out tensor)"""
    a = _lib.Net3dEdgeArgs()
    W_in, b_in, g_in, be_in, spec_in = edge_in.hot('silu')[:5]
    W_m, b_m, g_m, be_m, spec_m = msg_fc.hot()[:5]
    _tail(a.tail_in, spec_in, g_in, be_in, ar.take(H), ar.take(H), H, dev)
    _tail(a.tail_msg, spec_m, g_m, be_m, ar.take(H), ar.take(H), H, dev)
    a.num_nodes, a.num_edges, a.hidden, a.n_enc, a.reduce_mean = N, E, H, model.fourier_encodings, int(reduce_mean)
    a.ld_w_in, a.ld_w_msg = W_in.stride(0), W_m.stride(0)
    a.d_raw, a.perm, a.dst_s, a.in_ptr = d_raw.data_ptr(), idx.perm.data_ptr(), idx.dst_s.data_ptr(), idx.in_ptr.data_ptr()
    a.emb = model.node_embedding.data_ptr()
    a.W_in, a.b_in, a.W_msg, a.b_msg = W_in.data_ptr(), b_in.data_ptr(), W_m.data_ptr(), b_m.data_ptr()
    a.w_gate, a.b_gate = se.weight.data_ptr(), se.bias.data_ptr()
    a.stats = ar.take(int(L.i3d_net3d_edge_stats_floats(E, H)))
    a.aff_in, a.aff_msg = ar.take(3 * H), ar.take(3 * H)
    a.x_msg, a.msg, a.m_sum = ar.take(E * H), ar.take(E * H), ar.take(N * H)
    d_out = torch.empty(E, H, dtype=_F32, device=dev)
    a.d_out = d_out.data_ptr()
    # the storage form of a pass is fixed here (the backward call gets the same struct)
    a.store_bf16 = int(BF16_STORE and E >= BF16_STORE_MIN_EDGES and L.i3d_get_matmul_precision() != 0)
    a.x_center = ar.take(H)
    _chk(L.i3d_net3d_edge_fwd(ctypes.byref(a), stream), 'i3d_net3d_edge_fwd')
    return a, a.m_sum, d_out, (W_in, b_in, g_in, be_in, W_m, b_m, g_m, be_m, se.weight, se.bias)


class _Rec:
    """one step of the forward pass, with what its backward needs"""
    __slots__ = ('kind', 'a', 'fc', 'rows', 'fin', 'fout', 'x', 'y', 'extra')

    def __init__(self, kind, **kw):
        self.a = self.fc = self.extra = None
        self.rows = self.fin = self.fout = self.x = self.y = 0
        self.kind = kind
        for k, v in kw.items():
            setattr(self, k, v)


def _fc_floats(rows, fout, spec):
    """scratch of one FC block: xact, (pre_keep), y, mean, invstd"""
    return _al(rows * fout) * (3 if _keeps_pre(spec) else 2) + 2 * _al(fout)


def _fc_fwd(L, stream, ar, dev, fc, rows, x_ptr, residual=None, post_act=None, y_ptr=None):
    W, b, gamma, beta, spec = fc.hot(post_act)[:5]
    fout, fin = W.shape
    a = _lib.FcArgs()
    _tail(a.tail, spec, gamma, beta, ar.take(fout), ar.take(fout), fout, dev)
    a.rows, a.f_in, a.f_out, a.ldw = rows, fin, fout, W.stride(0)
    a.x, a.W, a.bias, a.residual = x_ptr, W.data_ptr(), b.data_ptr(), residual
    a.xact = ar.take(rows * fout)
    a.pre_keep = ar.take(rows * fout) if _keeps_pre(spec) else None
    a.y = y_ptr if y_ptr is not None else ar.take(rows * fout)
    _chk(L.i3d_fc_bn_fwd(ctypes.byref(a), stream), 'i3d_fc_bn_fwd')
    return _Rec('fc', a=a, fc=fc, rows=rows, fin=fin, fout=fout, x=x_ptr, y=a.y, extra=(W, b, gamma, beta, residual is not None))


def forward(ctx, model, g, params):
    idx = g.index()
    dev = g.device
    N, E, B = idx.num_nodes, idx.num_edges, idx.num_graphs
    edge_in, layers, nw, out = _blocks(model)
    H = model.node_embedding.shape[0]
    n_enc = model.fourier_encodings
    enc_dim = 2 * n_enc + 1 if n_enc > 0 else 1
    codes = model._readout_codes
    R = len(codes)
    L = _lib.load()
    stream = ops._stream()

    fused = fused_edge_ok(model)
    # ---- scratch size
    total = _al(N * H)
    if fused:
        total += 5 * _al(H) + 2 * _al(3 * H) + _al(int(L.i3d_net3d_edge_stats_floats(E, H))) + 2 * _al(E * H) + 2 * _al(N * H)
    else:
        total += _al(E) + (_al(E * enc_dim) if n_enc > 0 else 0) + _fc_floats(E, H, edge_in.hot('silu')[4])
    for msg, se, upd, _ in layers:
        if not fused:
            Fo = msg[0].hot()[0].shape[0]
            total += _al(N * 2 * Fo) + _al(E * Fo) + _fc_floats(E, Fo, msg[0].hot()[4])
            for fc in msg[1:]:
                total += _fc_floats(E, fc.hot()[0].shape[0], fc.hot()[4])
            total += _al(E * H) * 2 + _al(E) + _al(N * H) * 2           # d_next, gated message, gate, m_sum, u
        for fc in upd:
            total += _fc_floats(N, fc.hot()[0].shape[0], fc.hot()[4])
    for fc in nw:
        total += _fc_floats(N, fc.hot()[0].shape[0], fc.hot()[4])
    total += _al(B * R * H)
    for fc in out:
        total += _fc_floats(B, fc.hot()[0].shape[0], fc.hot()[4]) if fc.hot()[4].bn is not None else 0
    ar = _Arena(total + 64, dev)
    tr = {'layers': [], 'nw': [], 'out': []}

    # ---- inputs: broadcast node vector, distances in destination-sorted order, Fourier features (no gradient)
    d_raw = g.edata['d']
    if d_raw.dtype != _F32 or not d_raw.is_contiguous():
        d_raw = d_raw.contiguous().float()
    emb = model.node_embedding
    h = ar.take(N * H)
    _chk(L.i3d_broadcast_row(emb.data_ptr(), N, H, h, stream), 'i3d_broadcast_row')
    d_out = d = None
    if not fused:
        dperm = ar.take(E)
        _chk(L.i3d_gather_rows(d_raw.data_ptr(), idx.perm.data_ptr(), E, 1, dperm, stream), 'i3d_gather_rows')
        enc = dperm
        if n_enc > 0:
            enc = ar.take(E * enc_dim)
            _chk(L.i3d_fourier_encode(dperm, E, n_enc, enc, stream), 'i3d_fourier_encode')
        tr['edge_in'] = _fc_fwd(L, stream, ar, dev, edge_in, E, enc, post_act='silu')   # reference :80-81: d = silu(edge_input(d))
        d = tr['edge_in'].y

    # the node embeddings the network leaves on the graph are a tensor: the block that produces them writes into it
    last_h_fc = nw[-1] if nw else layers[-1][2][-1]
    h_out = torch.empty(N, last_h_fc.hot()[0].shape[0], dtype=_F32, device=dev)
    n_layers = len(layers)
    for li, (msg, se, upd, reduce_mean) in enumerate(layers):
        last_layer = li + 1 == n_layers
        lay = {'msg': [], 'upd': []}
        if fused:
            # edge input block, message block, gate and reduce in net3d_edge.hip; the graph's distance embedding comes out
            # in edge-id order
            lay['fused'], m_sum, d_out, ctx.fused_params = _edge_stage_fwd(L, stream, ar, dev, model, g, idx, d_raw, edge_in,
                                                                           msg[0], se, reduce_mean, N, E, H)
            u = ar.take(N * H)
            _chk(L.i3d_add(m_sum, h, N * H, u, stream), 'i3d_add')
            x = u
            for k, fc in enumerate(upd):
                final = k + 1 == len(upd)
                y_ptr = h_out.data_ptr() if (final and not nw) else None
                r = _fc_fwd(L, stream, ar, dev, fc, N, x, residual=h if final else None, y_ptr=y_ptr)     # reference :120-125
                lay['upd'].append(r)
                x = r.y
            tr['layers'].append(lay)
            h = x
            break
        # message network: first layer on [h_src | h_dst | d] through the node-level products (edge.hip), then plain blocks
        W, b, gamma, beta, spec = msg[0].hot()[:5]
        Fo = W.shape[0]
        a = _lib.EdgeFcArgs()
        _tail(a.tail, spec, gamma, beta, ar.take(Fo), ar.take(Fo), Fo, dev)
        a.num_nodes, a.num_edges, a.f_h, a.f_q, a.f_out, a.ldw = N, E, H, H, Fo, W.stride(0)
        a.h, a.q, a.W, a.bias = h, d, W.data_ptr(), b.data_ptr()
        a.src_s, a.dst_s, a.in_ptr = idx.src_s.data_ptr(), idx.dst_s.data_ptr(), idx.in_ptr.data_ptr()
        a.out_ptr, a.out_epos = idx.out_ptr.data_ptr(), idx.out_epos.data_ptr()
        a.P, a.Q = ar.take(N * 2 * Fo), ar.take(E * Fo)
        a.xact = ar.take(E * Fo)
        a.pre_keep = ar.take(E * Fo) if _keeps_pre(spec) else None
        a.y = ar.take(E * Fo)
        _chk(L.i3d_edge_fc_bn_fwd(ctypes.byref(a), stream), 'i3d_edge_fc_bn_fwd')
        lay['edge'] = _Rec('edge', a=a, fc=msg[0], fout=Fo, x=h, y=a.y, extra=(W, b, gamma, beta))
        m = a.y
        for fc in msg[1:]:
            r = _fc_fwd(L, stream, ar, dev, fc, E, m)
            lay['msg'].append(r)
            m = r.y
        d_next = d
        if not last_layer:                                   # reference :116 (dead for the last layer)
            d_next = ar.take(E * H)
            _chk(L.i3d_add(d, m, E * H, d_next, stream), 'i3d_add')
        # soft edge gate, mean / sum over the in-edges, + h
        msg_w, gate, m_sum, u = ar.take(E * H), ar.take(E), ar.take(N * H), ar.take(N * H)
        sw, sb = se.weight, se.bias
        _chk(L.i3d_soft_edge_fwd(m, sw.data_ptr(), sb.data_ptr(), E, H, msg_w, gate, stream), 'i3d_soft_edge_fwd')
        _chk(L.i3d_segment_sum(msg_w, H, idx.in_ptr.data_ptr(), None, N, H, int(reduce_mean), m_sum, H, stream), 'i3d_segment_sum')
        _chk(L.i3d_add(m_sum, h, N * H, u, stream), 'i3d_add')
        lay['gate'] = _Rec('gate', x=m, y=gate, extra=(sw, sb, reduce_mean))
        x = u
        for k, fc in enumerate(upd):
            final = k + 1 == len(upd)
            y_ptr = h_out.data_ptr() if (final and last_layer and not nw) else None
            r = _fc_fwd(L, stream, ar, dev, fc, N, x, residual=h if final else None, y_ptr=y_ptr)     # reference :120-125
            lay['upd'].append(r)
            x = r.y
        tr['layers'].append(lay)
        h, d = x, d_next
    for k, fc in enumerate(nw):
        r = _fc_fwd(L, stream, ar, dev, fc, N, h, y_ptr=h_out.data_ptr() if k + 1 == len(nw) else None)
        tr['nw'].append(r)
        h = r.y
    # side effects of the reference forward: final node embeddings and (edge-id order) distance embeddings on the graph
    g.ndata['feat'] = h_out
    if d_out is None:
        d_out = torch.empty(E, H, dtype=_F32, device=dev)
        _chk(L.i3d_gather_rows(d, idx.inv_perm.data_ptr(), E, H, d_out.data_ptr(), stream), 'i3d_gather_rows')
    g.edata['d'] = d_out

    # ---- readout and output network
    ro = ar.take(B * R * H)
    _chk(L.i3d_segment_readout_fwd(h, idx.graph_ptr.data_ptr(), B, H, _lib.int_array(codes), R, ro, stream), 'i3d_segment_readout_fwd')
    tr['readout'] = _Rec('readout', x=h, y=ro)
    x, fin = ro, R * H
    z = None
    for k, fc in enumerate(out):
        W, b, gamma, beta, spec = fc.hot()[:5]
        final = k + 1 == len(out)
        if final:
            z = torch.empty(B, W.shape[0], dtype=_F32, device=dev)
        if spec.bn is not None:
            r = _fc_fwd(L, stream, ar, dev, fc, B, x, y_ptr=z.data_ptr() if final else None)
            x, fin = r.y, W.shape[0]
        else:                                                # last layer: Linear only
            _chk(L.i3d_gemm_f32(0, 1, B, W.shape[0], fin, x, fin, W.data_ptr(), W.stride(0), z.data_ptr(), W.shape[0], b.data_ptr(), 0,
                                stream), 'i3d_gemm_f32')
            r = _Rec('linear', rows=B, fin=fin, fout=W.shape[0], x=x, extra=(W, b))
        tr['out'].append(r)
    # (the fused backward re-reads the raw distances and the distance embedding: both kept alive here)
    ctx.native = (ar, tr, idx, emb, (N, E, B, H), codes, h_out, (d_raw, d_out) if fused else None)
    ctx.params = params
    return z


def _fc_bwd(L, stream, ar, dev, r, grad_y, grads, need_x=True):
    """backward of an 'fc' record; returns the pointer of the input gradient (or None)"""
    a = r.a
    W, b, gamma, beta, _ = r.extra
    a.grad_y = grad_y
    a.grad_pre = ar.take(r.rows * r.fout)
    gW, gb, gg, gbe = tape.grad_like(W), tape.grad_like(b), tape.grad_like(gamma), tape.grad_like(beta)
    grads[id(W)], grads[id(b)], grads[id(gamma)], grads[id(beta)] = gW, gb, gg, gbe
    a.grad_W, a.grad_bias, a.grad_gamma, a.grad_beta = gW.data_ptr(), gb.data_ptr(), gg.data_ptr(), gbe.data_ptr()
    a.grad_x = ar.take(r.rows * r.fin) if need_x else None
    _set_workspaces(a.tail, r.fout, dev)
    _chk(L.i3d_fc_bn_bwd(ctypes.byref(a), stream), 'i3d_fc_bn_bwd')
    return a.grad_x


def _edge_stage_bwd(L, stream, ar, dev, a, gu, gh_l, emb, grads, ctx, N, E, H, ws_h):
    """backward of the fused edge stage: fills the gradients of the edge-input block, the message block, the gate and the
    node embedding (node-level part: column sums of gh_l; edge part added by the kernels)"""
    model_params = ctx.fused_params           # (W_in, b_in, g_in, be_in, W_m, b_m, g_m, be_m, sw, sb)
    bufs = [tape.grad_like(t) for t in model_params]
    for t, b in zip(model_params, bufs):
        grads[id(t)] = b
    (a.grad_W_in, a.grad_b_in, a.grad_gamma_in, a.grad_beta_in, a.grad_W_msg, a.grad_b_msg, a.grad_gamma_msg, a.grad_beta_msg,
     a.grad_w_gate, a.grad_b_gate) = [b.data_ptr() for b in bufs]
    gemb = tape.grad_like(emb)
    grads[id(emb)] = gemb
    _chk(L.i3d_colsum(gh_l, None, N, H, gemb.data_ptr(), ws_h, stream), 'i3d_colsum')
    a.grad_emb = gemb.data_ptr()
    a.grad_m_sum = gu
    a.grad_ya, a.grad_lin = ar.take(E * H), None        # (grad_lin: the fused backward pass keeps it in registers)
    a.partial = ar.take(int(L.i3d_net3d_edge_bwd_floats(E, H, a.n_enc)))
    _chk(L.i3d_net3d_edge_bwd(ctypes.byref(a), stream), 'i3d_net3d_edge_bwd')


def _bwd_floats(tr, N, E, H, L):
    def fc(r):
        return _al(r.rows * r.fout) + _al(r.rows * r.fin)
    total = 64 + _al(N * H)
    if 'edge_in' in tr:
        total += fc(tr['edge_in'])
    for r in tr['out']:
        total += fc(r) if r.kind == 'fc' else _al(r.rows * r.fin)
    for r in tr['nw']:
        total += fc(r)
    for lay in tr['layers']:
        total += sum(fc(r) for r in lay['upd'] + lay['msg'])
        if 'fused' in lay:
            a = lay['fused']
            total += _al(N * H) + _al(E * H) + _al(int(L.i3d_net3d_edge_bwd_floats(E, H, a.n_enc)))
            continue
        Fo = lay['edge'].fout
        total += 2 * _al(N * H) + 3 * _al(E * H) + _al(E) + _al(E * Fo) + _al(N * 2 * Fo)
    return total


def backward(ctx, grad_z):
    ar_f, tr, idx, emb, (N, E, B, H), codes, h_out, _edge_keep = ctx.native
    dev = grad_z.device
    grad_z = grad_z.contiguous()
    L = _lib.load()
    stream = ops._stream()
    grads = {}
    ar = _Arena(_bwd_floats(tr, N, E, H, L), dev)
    ws_h = ops._workspace(H, dev).data_ptr()

    # ---- output network
    g = grad_z.data_ptr()
    for r in reversed(tr['out']):
        if r.kind == 'linear':
            W, b = r.extra
            gW, gb = tape.grad_like(W), tape.grad_like(b)
            grads[id(W)], grads[id(b)] = gW, gb
            _chk(L.i3d_gemm_f32_ws(1, 0, r.fout, r.fin, r.rows, g, r.fout, r.x, r.fin, gW.data_ptr(), W.stride(0), None, 0,
                                   ops._gemm_workspace(dev).data_ptr(), ops.GEMM_WORKSPACE_BYTES, stream), 'i3d_gemm_f32_ws')
            _chk(L.i3d_colsum(g, None, r.rows, r.fout, gb.data_ptr(), ops._workspace(r.fout, dev).data_ptr(), stream), 'i3d_colsum')
            gx = ar.take(r.rows * r.fin)
            _chk(L.i3d_gemm_f32(0, 0, r.rows, r.fin, r.fout, g, r.fout, W.data_ptr(), W.stride(0), gx, r.fin, None, 0, stream),
                 'i3d_gemm_f32')
            g = gx
        else:
            g = _fc_bwd(L, stream, ar, dev, r, g, grads)
    # ---- readout, node-wise output blocks
    gh = ar.take(N * H)
    _chk(L.i3d_segment_readout_bwd(g, tr['readout'].x, idx.graph_ptr.data_ptr(), B, H, _lib.int_array(codes), len(codes), gh, stream),
         'i3d_segment_readout_bwd')
    for r in reversed(tr['nw']):
        gh = _fc_bwd(L, stream, ar, dev, r, gh, grads)
    # ---- layers, last first.  gd: gradient w.r.t. the distance embedding a layer handed to the next one (None: unused there)
    gd = None
    for lay in reversed(tr['layers']):
        gu = gh
        for r in reversed(lay['upd']):
            gu = _fc_bwd(L, stream, ar, dev, r, gu, grads)
        if 'fused' in lay:
            gh_l = ar.take(N * H)
            _chk(L.i3d_add(gh, gu, N * H, gh_l, stream), 'i3d_add')
            _edge_stage_bwd(L, stream, ar, dev, lay['fused'], gu, gh_l, emb, grads, ctx, N, E, H, ws_h)
            return tuple(grads.get(id(p)) for p in ctx.params)
        gate = lay['gate']
        sw, sb, reduce_mean = gate.extra
        # h enters the layer three times: the residual of the last update block (gradient = gh), the sum m_sum + h
        # (gradient = gu) and the edge block (below)
        gh_l = ar.take(N * H)
        _chk(L.i3d_add(gh, gu, N * H, gh_l, stream), 'i3d_add')
        gmsg = ar.take(E * H)
        _chk(L.i3d_segment_bcast(gu, idx.in_ptr.data_ptr(), idx.dst_s.data_ptr(), E, H, int(reduce_mean), gmsg, stream),
             'i3d_segment_bcast')
        gm, gg = ar.take(E * H), ar.take(E)
        _chk(L.i3d_soft_edge_bwd(gmsg, gate.x, gate.y, sw.data_ptr(), E, H, gm, gg, stream), 'i3d_soft_edge_bwd')
        gsw, gsb = tape.grad_like(sw), tape.grad_like(sb)
        grads[id(sw)], grads[id(sb)] = gsw, gsb
        _chk(L.i3d_colsum(gate.x, gg, E, H, gsw.data_ptr(), ws_h, stream), 'i3d_colsum')
        _chk(L.i3d_colsum(gg, None, E, 1, gsb.data_ptr(), ops._workspace(1, dev).data_ptr(), stream), 'i3d_colsum')
        if gd is not None:                           # d_next = d + m fed the next layer: its gradient reaches m and d
            _chk(L.i3d_add_inplace(gm, gd, E * H, stream), 'i3d_add_inplace')
        for r in reversed(lay['msg']):
            gm = _fc_bwd(L, stream, ar, dev, r, gm, grads)
        e = lay['edge']
        a = e.a
        W, b, gamma, beta = e.extra
        Fo = e.fout
        a.grad_y, a.grad_pre, a.grad_P = gm, ar.take(E * Fo), ar.take(N * 2 * Fo)
        gW, gb, gga, gbe = tape.grad_like(W), tape.grad_like(b), tape.grad_like(gamma), tape.grad_like(beta)
        grads[id(W)], grads[id(b)], grads[id(gamma)], grads[id(beta)] = gW, gb, gga, gbe
        a.grad_W, a.grad_bias, a.grad_gamma, a.grad_beta = gW.data_ptr(), gb.data_ptr(), gga.data_ptr(), gbe.data_ptr()
        a.grad_h, a.grad_q = ar.take(N * H), ar.take(E * H)
        _set_workspaces(a.tail, Fo, dev)
        _chk(L.i3d_edge_fc_bn_bwd(ctypes.byref(a), stream), 'i3d_edge_fc_bn_bwd')
        _chk(L.i3d_add_inplace(gh_l, a.grad_h, N * H, stream), 'i3d_add_inplace')
        if gd is not None:
            _chk(L.i3d_add_inplace(a.grad_q, gd, E * H, stream), 'i3d_add_inplace')
        gd, gh = a.grad_q, gh_l
    # ---- edge input block (its input - the Fourier features - has no gradient) and the node embedding
    _fc_bwd(L, stream, ar, dev, tr['edge_in'], gd, grads, need_x=False)
    gemb = tape.grad_like(emb)
    grads[id(emb)] = gemb
    _chk(L.i3d_colsum(gh, None, N, H, gemb.data_ptr(), ws_h, stream), 'i3d_colsum')
    return tuple(grads.get(id(p)) for p in ctx.params)
