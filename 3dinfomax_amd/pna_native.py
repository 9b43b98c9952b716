"""PNA forward / backward of a training step through the whole-model C sequencer (csrc/model.hip).

Why: the step is bound by the host, not by the GPU (tools/host_segments.py: ~0.8 ms of launches, ~1.8 ms of Python /
torch around them per 2.7 ms step).  For the pre-training configuration the whole PNA forward is ONE C call and the whole
backward another: Python describes the model once (`_describe`: parameter, BatchNorm-state and gradient pointers -
rebuilt only when a parameter object or the gradient buffers change) and the batch per call (sizes and index pointers),
allocates the saved-activation buffer, the backward scratch and the three tensors the caller sees (output, node
embeddings, bond embeddings); the memory layout and the launch order live in C++.  Same kernels in the same order as the
Python-sequenced path (pna.PNALayerFn / layer_native with fused BatchNorm): the same bits.

Eligible: training mode with gradients on, every block Linear -> {none, ReLU, LeakyReLU} -> BatchNorm1d with local
statistics (no sync group), >= 2 degree scalers, categorical bond features, one posttrans layer, all widths multiples
of 4.  Anything else runs the Python-sequenced path (I3D_NATIVE_MODEL=0 forces it).
"""
import ctypes
import os

import torch

from . import _lib, layer_native, ops, streams, tape

NATIVE_MODEL = os.environ.get('I3D_NATIVE_MODEL', '1') != '0'
_SIMPLE = (None, 'relu', 'leakyrelu')


class _Desc:
    """ctypes description of one PNA module + what it was built from (to notice when it is stale)"""
    __slots__ = ('struct', 'params', 'param_ids', 'keep', 'grad_key', 'fcs', 'bn_feat', 'target_dim', 'emb_params')


def _fc_ok(fc, need_bn):
    if not fc.bias:                     # FCLayer(bias=False): the block Functions only
        return False
    if fc.dropout is not None:          # dropout > 0: the per-kernel path (layers._Tail)
        return False
    if fc.batch_norm is None:
        return not need_bn and fc.activation in _SIMPLE
    return (fc.activation in _SIMPLE and fc.sync_group is None and fc.batch_norm.affine
            and fc.batch_norm.track_running_stats and fc.batch_norm.momentum is not None)


def eligible(module, g):
    """cheap per-call checks; the structural ones are cached on the module"""
    if not (NATIVE_MODEL and layer_native.FUSED_BN and layer_native.NATIVE_LAYER):
        return False
    # training mode (with or without autograd: the reference's inference.py runs train-mode BatchNorm under no_grad), or
    # eval mode without autograd - the validation pass of trainer/trainer.py:72-78 (BatchNorm with running statistics,
    # forward only: csrc/model.hip I3dPnaModel.training = 0)
    if not module.training and torch.is_grad_enabled():
        return False
    if tape.active() is not None or not tape.FUSED_MODEL:
        return False
    from . import layers as _layers
    from . import pna as P
    if not (P.GROUPED_POSTTRANS and P.FUSED_LAYER and P.EDGE_TABLE and _layers.COMPOSITE):      # A/B switches of the tests
        return False
    plist = tape._param_list(module)           # a new list object whenever a sub-module / parameter was replaced
    ent = module.__dict__.get('_i3d_native_ok')
    if ent is None or ent[0] is not plist:
        from .layers import FCLayer
        ent = module.__dict__['_i3d_native_ok'] = (plist, _structure_ok(module),
                                                   [m for m in module.modules() if isinstance(m, FCLayer)])
    if not ent[1]:
        return False
    for fc in ent[2]:                            # dist.setup(sync_bn=True) may attach a group after the first forward
        if fc.sync_group is not None:
            return False
    for fc in ent[2]:                            # every block in the model's mode (a frozen BatchNorm inside a training model:
        bn = fc.batch_norm                       # the per-block path)
        if fc.training != module.training or (bn is not None and bn.training != module.training):
            return False
    if not _layers.COMPOSITE or (module.training and not _layers._composite_ok(ent[2][0].hot()[4])):
        return False                             # the gate of the block composites (local statistics)
    feat = g.ndata.get('feat')
    ef = g.edata.get('feat')
    if feat is None or ef is None or not feat.is_cuda or feat.dtype != torch.int64 or ef.dtype != torch.int64:
        return False
    idx = g.index()
    if idx.num_edges == 0 or idx.num_nodes == 0:
        return False
    groups = idx.degree_groups()[2]
    n_sc = len(module.node_gnn.mp_layers[0].scalers)
    return len(groups) <= 32 and len(groups) * n_sc <= 128


def _structure_ok(module):
    gnn = module.node_gnn
    if len(gnn.mp_layers) < 1 or len(gnn.mp_layers) > 16:
        return False
    first = gnn.mp_layers[0]
    hidden = gnn.atom_encoder.atom_embedding_list[0].embedding_dim
    if hidden % 4 or len(first.scalers) < 2 or not first.edge_features:
        return False
    if getattr(gnn.atom_encoder, 'padding', False) or getattr(gnn.bond_encoder, 'padding', False):
        return False     # padding=True encoders (extra row 0, lookup at x + 1, no gradient for row 0): layers.EmbeddingSumPadFn only
    n_comb = 1
    for d in gnn.bond_encoder.dims:
        n_comb *= d
    if n_comb > 256 or len(gnn.bond_encoder.dims) > 8 or len(gnn.atom_encoder.dims) > 16:
        return False
    for layer in gnn.mp_layers:
        if layer.pairwise_distances:        # a per-edge input column: the block Functions (pna.PNAGNN.forward)
            return False
        pre, post = list(layer.pretrans.fully_connected), list(layer.posttrans.fully_connected)
        if len(post) != 1 or not 1 <= len(pre) <= 4:
            return False
        if layer.aggregators != first.aggregators or layer.scalers != first.scalers or layer.residual != first.residual:
            return False
        if float(layer.avg_d['log']) != float(first.avg_d['log']):
            return False
        for fc in pre + post:
            if not _fc_ok(fc, True) or fc.out_dim % 4 or fc.in_dim % 4:
                return False
        if pre[0].in_dim != 3 * hidden or post[0].out_dim != hidden:
            return False
        for i in range(1, len(pre)):
            if pre[i].in_dim > 1024:
                return False
    head = list(module.output.fully_connected)
    if not 1 <= len(head) <= 4:
        return False
    for fc in head:
        if not _fc_ok(fc, False):
            return False
    return True


def _set_fc(dst, fc, grads):
    """I3dFcParams of one FCLayer; grads: {id(param): tensor} or None (forward only)"""
    W, b = fc.linear.weight, fc.linear.bias
    dst.W, dst.bias = W.data_ptr(), b.data_ptr()
    dst.f_in, dst.f_out, dst.act = fc.in_dim, fc.out_dim, ops.ACT[fc.activation]
    keep = [W, b]
    bn = fc.batch_norm
    if bn is not None:
        dst.gamma, dst.beta = bn.weight.data_ptr(), bn.bias.data_ptr()
        dst.running_mean, dst.running_var = bn.running_mean.data_ptr(), bn.running_var.data_ptr()
        dst.num_batches_tracked = bn.num_batches_tracked.data_ptr()
        dst.eps, dst.momentum = bn.eps, bn.momentum
        keep += [bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.num_batches_tracked]
    if grads is not None:
        dst.grad_W, dst.grad_bias = grads[id(W)].data_ptr(), grads[id(b)].data_ptr()
        if bn is not None:
            dst.grad_gamma, dst.grad_beta = grads[id(bn.weight)].data_ptr(), grads[id(bn.bias)].data_ptr()
    return keep


def _describe(module, grads=None):
    """(re)build the I3dPnaModel struct; `grads` {id(param): gradient buffer} for the backward pass"""
    gnn = module.node_gnn
    d = _Desc()
    m = _lib.PnaModel()
    first = gnn.mp_layers[0]
    hidden = gnn.atom_encoder.atom_embedding_list[0].embedding_dim
    m.training = 1
    m.n_layers, m.hidden, m.residual = len(gnn.mp_layers), hidden, 1 if first.residual else 0
    m.n_pre = len(first.pretrans.fully_connected)
    m.n_aggregators = len(first.aggregators)
    for i, v in enumerate(first.aggregators):
        m.aggregators[i] = v
    m.n_scalers = len(first.scalers)
    for i, v in enumerate(first.scalers):
        m.scalers[i] = v
    m.avg_d_log = float(first.avg_d['log'])
    keep, bn_feat = [], hidden
    for l, layer in enumerate(gnn.mp_layers):
        for i, fc in enumerate(layer.pretrans.fully_connected):
            keep += _set_fc(m.pre[l][i], fc, grads)
            bn_feat = max(bn_feat, fc.out_dim)
        keep += _set_fc(m.post[l], layer.posttrans.fully_connected[0], grads)
    at, bt = [e.weight for e in gnn.atom_encoder.atom_embedding_list], [e.weight for e in gnn.bond_encoder.bond_embedding_list]
    m.n_atom_tables, m.n_bond_tables = len(at), len(bt)
    for i, t in enumerate(at):
        m.atom_dims[i], m.atom_tables[i] = t.shape[0], t.data_ptr()
    for i, t in enumerate(bt):
        m.bond_dims[i], m.bond_tables[i] = t.shape[0], t.data_ptr()
    keep += at + bt
    d.emb_params = (at, bt)
    if grads is not None:
        m.grad_atom_tables, m.grad_bond_tables = grads[id(at[0])].data_ptr(), grads[id(bt[0])].data_ptr()
    m.n_readout = len(module._readout_codes)
    for i, v in enumerate(module._readout_codes):
        m.readout_ops[i] = v
    head = list(module.output.fully_connected)
    m.n_head = len(head)
    for i, fc in enumerate(head):
        keep += _set_fc(m.head[i], fc, grads)
        if fc.batch_norm is not None:
            bn_feat = max(bn_feat, fc.out_dim)
    d.struct, d.keep, d.bn_feat, d.target_dim = m, keep, bn_feat, head[-1].out_dim
    return d


def _contiguous_views(tables, views):
    """are the gradient buffers of these tables one contiguous [sum rows, F] block, in table order?"""
    p = None
    for t in tables:
        v = views.get(id(t))
        if v is None or not v.is_contiguous() or (p is not None and v.data_ptr() != p):
            return False
        p = v.data_ptr() + 4 * v.numel()
    return True


def _batch_struct(g, idx, gnn):
    st = idx.__dict__.get('_i3d_pna_batch')
    if st is None:
        b = _lib.PnaBatch()
        rows_d, tiles_d, groups = idx.degree_groups()
        b.num_nodes, b.num_edges, b.num_graphs = idx.num_nodes, idx.num_edges, idx.num_graphs
        b.in_ptr, b.perm, b.src_s, b.dst_s = idx.in_ptr.data_ptr(), idx.perm.data_ptr(), idx.src_s.data_ptr(), idx.dst_s.data_ptr()
        b.out_ptr, b.out_epos, b.graph_ptr = idx.out_ptr.data_ptr(), idx.out_epos.data_ptr(), idx.graph_ptr.data_ptr()
        b.deg_rows, b.deg_tile_group = rows_d.data_ptr(), tiles_d.data_ptr()
        b.m_padded, b.n_groups = rows_d.shape[0], len(groups)
        for k, (D, start, count) in enumerate(groups):
            b.group_degree[k], b.group_start[k], b.group_count[k] = D, start, count
        st = idx.__dict__['_i3d_pna_batch'] = b
    dims = gnn.bond_encoder.dims
    n_comb = 1
    for d in dims:
        n_comb *= d
    comb = gnn._combinations(dims, idx.in_ptr.device)
    st.comb, st.n_comb, st.v_pad = comb.data_ptr(), n_comb, (n_comb + 31) // 32 * 32
    return st, comb


class _CtxHandle:
    """owns the host-side context of one forward pass (csrc/model.hip: PnaCtx)"""

    def __init__(self, ptr):
        self.ptr = ptr

    def __del__(self):
        try:
            if self.ptr:
                _lib.load().i3d_pna_model_ctx_free(self.ptr)
        except Exception:
            pass


class PNAModelFn(torch.autograd.Function):
    """The whole PNA model as one autograd node whose forward and backward are one C call each."""

    @staticmethod
    def forward(ctx, module, g, state, *params):
        gnn = module.node_gnn
        idx = g.index()
        dev = idx.in_ptr.device
        desc = module.__dict__.get('_i3d_desc_fwd')
        ids = tuple(id(p) for p in params)
        if desc is None or desc.param_ids != ids:
            desc = _describe(module)
            desc.param_ids = ids
            module.__dict__['_i3d_desc_fwd'] = desc
        desc.struct.training = 1 if module.training else 0
        atom_feat, bond_feat = g.ndata['feat'].contiguous(), g.edata['feat'].contiguous()
        b, comb = _batch_struct(g, idx, gnn)
        b.atom_feat, b.bond_feat = atom_feat.data_ptr(), bond_feat.data_ptr()
        L = _lib.load()
        n_saved = L.i3d_pna_model_saved_floats(ctypes.byref(desc.struct), ctypes.byref(b))
        if n_saved < 0:
            _lib.check(1, 'i3d_pna_model_saved_floats')
        N, E, B, F = idx.num_nodes, idx.num_edges, idx.num_graphs, desc.struct.hidden
        saved = torch.empty(n_saved, dtype=torch.float32, device=dev)
        node_emb = torch.empty(N, F, dtype=torch.float32, device=dev)
        edge_emb = torch.empty(E, F, dtype=torch.float32, device=dev)
        out = torch.empty(B, desc.target_dim, dtype=torch.float32, device=dev)
        events = None
        if ops.KERNEL_TIMERS is not None:      # bench.py: HIP events around the roofline kernel of every layer
            nl = desc.struct.n_layers
            evs = [ops.RawEvent() for _ in range(2 * nl)]
            events = (ctypes.c_void_p * (2 * nl))(*[e.handle for e in evs])
            f_msg = gnn.mp_layers[0].pretrans.fully_connected[-1].out_dim
            for l in range(nl):
                ops.KERNEL_TIMERS.setdefault('pna_aggregate_fwd', []).append(
                    (evs[2 * l], evs[2 * l + 1], N, E, f_msg, desc.struct.n_aggregators * f_msg))
        handle = ctypes.c_void_p()
        _lib.check(L.i3d_pna_model_fwd(ctypes.byref(desc.struct), ctypes.byref(b), saved.data_ptr(), node_emb.data_ptr(),
                                       edge_emb.data_ptr(), out.data_ptr(), ops._workspace(desc.bn_feat, dev).data_ptr(), events,
                                       ops._stream(), ctypes.byref(handle)), 'i3d_pna_model_fwd')
        ctx.handle = _CtxHandle(handle.value)
        # everything the C context points into stays alive with the autograd node
        ctx.keep = (saved, node_emb, atom_feat, bond_feat, comb, idx, desc)
        ctx.module, ctx.state, ctx.params, ctx.batch = module, state, params, b
        ctx.mark_non_differentiable(node_emb, edge_emb)
        # without this the engine hands backward() freshly zero-filled [N,F] and [E,F] tensors for the two embeddings
        # (two fill kernels of 6.7 + 13.3 MB on the critical path in front of the backward pass)
        ctx.set_materialize_grads(False)
        return out, node_emb, edge_emb

    @staticmethod
    def backward(ctx, grad, _g_node, _g_edge):
        if grad is None:                     # (materialize_grads is off: the output itself did not reach the loss)
            return (None,) * (3 + len(ctx.params))
        streams.invalidate_step()
        module, params = ctx.module, ctx.params
        dev = grad.device
        direct = tape.DIRECT_PARAM_GRADS and tape._plain_leaves(params)
        if direct and tape.PERSISTENT_GRADS:
            pool = ctx.state.pool_for(params)
            views, key = pool.view_of, ('pool', id(pool))
        else:
            sizes = [p.numel() for p in params]
            flat = torch.empty(sum(sizes), dtype=torch.float32, device=dev)
            views, key = {id(p): v.view_as(p) for p, v in zip(params, flat.split(sizes))}, None
        at, bt = ctx.keep[6].emb_params
        if not (_contiguous_views(at, views) and _contiguous_views(bt, views)):
            # gradient buffers of the embedding tables are written as one [sum rows, F] block each
            views = dict(views)
            for tabs in (at, bt):
                blk = torch.empty(sum(t.shape[0] for t in tabs), tabs[0].shape[1], dtype=torch.float32, device=dev)
                o = 0
                for t in tabs:
                    views[id(t)] = blk[o:o + t.shape[0]]
                    o += t.shape[0]
            key = None
        desc = module.__dict__.get('_i3d_desc_bwd') if key is not None else None
        ids = tuple(id(p) for p in params)
        if desc is None or desc.grad_key != key or desc.param_ids != ids:
            desc = _describe(module, views)
            desc.grad_key, desc.param_ids = key, ids
            if key is not None:
                module.__dict__['_i3d_desc_bwd'] = desc
        L = _lib.load()
        n_scratch = L.i3d_pna_model_scratch_floats(ctypes.byref(desc.struct), ctypes.byref(ctx.batch))
        scratch = torch.empty(n_scratch, dtype=torch.float32, device=dev)
        grad = grad.contiguous()
        bn_ws, gemm_ws = ops._workspace(desc.bn_feat, dev).data_ptr(), ops._gemm_workspace(dev).data_ptr()

        def part(k, split):
            _lib.check(L.i3d_pna_model_bwd_part(ctx.handle.ptr, ctypes.byref(desc.struct), grad.data_ptr(), scratch.data_ptr(), bn_ws,
                                                gemm_ws, ops.GEMM_WORKSPACE_BYTES, k, split, ops._stream()), 'i3d_pna_model_bwd_part')

        red = ctx.state.reducer
        n_layers = desc.struct.n_layers
        if (red is not None and red._agreed and n_layers >= 2 and key is not None
                and ctx.state.sink_views is views and (_world(red.group) > 1 or _FORCE_EARLY)):
            # data parallel: the gradients of the head and of the upper half of the layers are final after part 1 - their
            # all-reduce (RCCL, its own stream) runs next to the lower half of the backward pass
            split = n_layers // 2
            part(1, split)
            red.launch_async(module)         # the slices of layers [split, L) + head (GradReducer._plan_early: the same split)
            part(2, split)
        else:
            part(0, 0)
        out = [views[id(p)] for p in params]
        sink = ctx.state.sink
        if sink is not None:         # data parallel: the gradients go (or already are) in the all-reduce buffer
            out = sink(params, out)
        ctx.keep = ctx.handle = None
        if direct:
            for p, gr in zip(params, out):
                p.grad = gr
            return (None,) * (3 + len(params))
        return (None, None, None) + tuple(out)


# test hook (tests/test_gpu_dist.py): the split backward pass + early all-reduce at world 1, on the one-GPU box
_FORCE_EARLY = _lib.TEST_HOOKS['force_early_allreduce']


def _world(group):
    import torch.distributed as dist
    return dist.get_world_size(group) if dist.is_initialized() else 1


def run(module, g):
    """PNA.forward body for an eligible call: -> output [B, target_dim]; sets the graph's feature side effects"""
    params = [p for p in tape._param_list(module) if p.requires_grad]
    if len(params) != len(tape._param_list(module)):
        return None                      # frozen parameters: the Python-sequenced path handles them
    out, node_emb, edge_emb = PNAModelFn.apply(module, g, tape.model_state(module), *params)
    g.ndata['feat'] = node_emb           # reference models/pna.py:213
    g.edata['feat'] = edge_emb           # reference models/pna.py:163
    return out


def debug_messages(out, layer, num_edges, feat):
    """test hook: the messages of `layer` as the aggregation kernels of the forward pass that produced `out` read them,
    [E, feat] in destination-sorted order (csrc/model.hip: i3d_pna_model_debug_messages)"""
    node = out.grad_fn
    handle = getattr(node, 'handle', None)
    if handle is None:
        raise RuntimeError('not the output of the native PNA sequencer (or its backward pass has already run)')
    msg = torch.empty(num_edges, feat, dtype=torch.float32, device=out.device)
    _lib.check(_lib.load().i3d_pna_model_debug_messages(handle.ptr, layer, msg.data_ptr(), ops._stream()),
               'i3d_pna_model_debug_messages')
    return msg
