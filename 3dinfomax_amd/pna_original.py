"""Original-PNA variants on the MI355X kernels - drop-in for `PNAOriginal` / `PNAOriginalSimple` (and their
`PNAGNNOriginal`, `PNAGNNSimple`, `PNATower`, `PNALayer`, `PNASimpleLayer`, `MLPReadout` parts) of reference
models/pna_original.py + models/base_layers.py:149-164.  Same constructor kwargs, sub-module names and
state_dict keys.  They reuse the hot path's kernels: the segmented aggregation (K4) with the REAL `avg_d` and
always-applied scalers (reference :231-236 - no single-scaler quirk here), the edge gather-combine for the tower
pretrans, the two-segment posttrans GEMMs, `h * snorm_n` as a row-scale kernel, LeakyReLU mixing.

dropout / in_feat_dropout > 0 (configs/pna_original_simple.yml: 0.3) draw torch's own mask (layers.DropoutFn) and keep the
towers on the per-tower path, as do use_3d and gru_enable.  Refused by name: the moment aggregators.
"""
import ctypes
import os

import torch
import torch.nn as nn

from . import _lib, layers as _layers, ops, tape
from .graph import as_batched_graph
from .layers import (MLP, AggregateFn, BNSpec, Concat2FCFn, EdgeFCFn, FCFn, FCSpec, GroupedConcat2FCFn, ReadoutFn,
                     dropout as _dropout)
from .mol_encoder import AtomEncoder, BondEncoder
from .pna import _AppendSqDistFn, _codes, _GatherRowsFn, _scaler_coef

# I3D_TOWER_STACK=0: the towers of a layer one after the other (one autograd node per block and tower: the first version)
TOWER_STACK = True
# I3D_TOWER_PAD=0: the stacked layers at the model's own widths (hidden_dim 90 / edge_hidden_dim 70 of the yml: every kernel of the
# layer in its unaligned form - 4.8 ms per step at batch 512 against the padded form's)
PAD_WIDTHS = True
# I3D_TOWER_BLOCKS=0: the posttrans products of a stacked layer as ONE dense product on the zero-padded stacked weight instead of
# `towers` diagonal blocks
TOWER_BLOCKS = True
# I3D_TOWER_FOLD=0: the aggregation of a stacked layer with all its scaler blocks ([N, 12 F]) instead of the scalers folded into
# per-degree posttrans weights as in the 2D network (the aggregation writes its identity blocks only, K of the products on it is
# n_scalers times shorter, the aggregated tensor n_scalers times smaller; with I3D_TOWER_BLOCKS the per-degree weights' diagonal
# blocks are multiplied, without it one dense grouped product)
TOWER_FOLD = True


class _RowScaleFn(torch.autograd.Function):
    """h * snorm_n  (reference models/pna_original.py:257-258); snorm_n [N,1] is data (no gradient)."""

    @staticmethod
    def forward(ctx, h, s):
        s = s.reshape(-1).contiguous().float()
        ctx.save_for_backward(s)
        return ops.row_scale(h.contiguous(), s)

    @staticmethod
    def backward(ctx, g):
        (s,) = ctx.saved_tensors
        return ops.row_scale(g.contiguous(), s), None


class _CopyColsFn(torch.autograd.Function):
    """x [rows, c] -> [rows, cols]: zeros in the new columns (cols > c) or the first cols columns (csrc/pack.hip: i3d_copy_cols);
    the gradient goes the other way round"""

    @staticmethod
    def forward(ctx, x, cols):
        x = x.contiguous()
        ctx.cols_in = x.shape[1]
        out = torch.empty(x.shape[0], cols, dtype=torch.float32, device=x.device)
        _lib.check(_lib.load().i3d_copy_cols(x.data_ptr(), x.shape[0], x.shape[1], out.data_ptr(), cols, ops._stream()), 'i3d_copy_cols')
        return out

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous()
        out = torch.empty(g.shape[0], ctx.cols_in, dtype=torch.float32, device=g.device)
        _lib.check(_lib.load().i3d_copy_cols(g.data_ptr(), g.shape[0], g.shape[1], out.data_ptr(), ctx.cols_in, ops._stream()),
                   'i3d_copy_cols')
        return out, None


class _GatherSrcFn(torch.autograd.Function):
    """DGL fn.copy_u('feat','m'): message of edge j = features of its source node (destination-sorted edge order);
    backward = sum over each node's out-edges (segmented, no atomics)."""

    @staticmethod
    def forward(ctx, h, index):
        ctx.index = index
        return ops.gather_rows(h.contiguous(), index.src_s)

    @staticmethod
    def backward(ctx, g):
        idx = ctx.index
        return ops.segment_sum(g.contiguous(), idx.out_ptr, idx.out_epos, idx.num_nodes), None


class _AddFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        return ops.add_inplace(a.clone(), b.contiguous())

    @staticmethod
    def backward(ctx, g):
        return g, g


class MLPReadout(nn.Module):
    """reference models/base_layers.py:149-164: L x (Linear halving the width + ReLU), then Linear."""

    def __init__(self, input_dim, output_dim, L=2):
        super().__init__()
        layers = [nn.Linear(input_dim // 2 ** l, input_dim // 2 ** (l + 1), bias=True) for l in range(L)]
        layers.append(nn.Linear(input_dim // 2 ** L, output_dim, bias=True))
        self.FC_layers = nn.ModuleList(layers)
        self.L = L

    def forward(self, x):
        y = x
        for l in range(self.L):
            y = tape.apply(FCFn, y, self.FC_layers[l].weight, self.FC_layers[l].bias, None, None, None, FCSpec('relu', None))
        return tape.apply(FCFn, y, self.FC_layers[self.L].weight, self.FC_layers[self.L].bias, None, None, None,
                          FCSpec(None, None))


class _GRUCellFn(torch.autograd.Function):
    """One GRU step h' = GRU(x, h0) (torch's nn.GRU arithmetic, gate order r | z | n): two products of the library's GEMM and the
    gate kernels of csrc/gru.hip; backward: the gate kernel, then four products and two column sums."""

    @staticmethod
    def forward(ctx, x, h0, W_ih, W_hh, b_ih, b_hh):
        x, h0 = x.contiguous(), h0.contiguous()
        N, H = h0.shape
        GI = ops.gemm(x, W_ih, trans_b=True, bias=b_ih)
        GH = ops.gemm(h0, W_hh, trans_b=True, bias=b_hh)
        out = torch.empty_like(h0)
        saved = torch.empty(N, 3 * H, dtype=torch.float32, device=h0.device)
        _lib.check(_lib.load().i3d_gru_gates_fwd(GI.data_ptr(), GH.data_ptr(), h0.data_ptr(), N, H, out.data_ptr(), saved.data_ptr(),
                                                 ops._stream()), 'i3d_gru_gates_fwd')
        ctx.save_for_backward(x, h0, W_ih, W_hh, GH, saved)
        return out

    @staticmethod
    def backward(ctx, g):
        x, h0, W_ih, W_hh, GH, saved = ctx.saved_tensors
        g = g.contiguous()
        N, H = h0.shape
        dGI, dGH, dh0 = torch.empty_like(GH), torch.empty_like(GH), torch.empty_like(h0)
        _lib.check(_lib.load().i3d_gru_gates_bwd(g.data_ptr(), saved.data_ptr(), GH.data_ptr(), h0.data_ptr(), N, H, dGI.data_ptr(),
                                                 dGH.data_ptr(), dh0.data_ptr(), ops._stream()), 'i3d_gru_gates_bwd')
        dx = ops.gemm(dGI, W_ih)
        ops.gemm(dGH, W_hh, out=dh0, accumulate=True)
        return (dx, dh0, ops.gemm(dGI, x, trans_a=True), ops.gemm(dGH, h0, trans_a=True), ops.colsum(dGI), ops.colsum(dGH))


class GRU(nn.Module):
    """reference models/pna_original.py:64-84: a one-step nn.GRU (x = the layer's input, hidden state = the layer's output).  The
    nn.GRU module holds the parameters (the reference's state_dict keys gru.weight_ih_l0 ...); the step itself is _GRUCellFn."""

    def __init__(self, input_size, hidden_size, device):
        super().__init__()
        self.input_size, self.hidden_size = input_size, hidden_size
        self.gru = nn.GRU(input_size=input_size, hidden_size=hidden_size).to(device)

    def forward(self, x, y):
        assert x.shape[-1] == self.input_size and y.shape[-1] == self.hidden_size      # (the reference pads narrower inputs: not needed here)
        m = self.gru
        return tape.apply(_GRUCellFn, x, y, m.weight_ih_l0, m.weight_hh_l0, m.bias_ih_l0, m.bias_hh_l0)


def _check_unsupported(dropout=0.0, in_feat_dropout=0.0, gru_enable=False, use_3d=False):
    pass        # (every option of the variant is offered: dropout, use_3d and gru_enable keep the towers on the per-tower path)


class PNAOriginal(nn.Module):
    """reference models/pna_original.py:119-146."""

    def __init__(self, hidden_dim, last_layer_dim, target_dim, in_feat_dropout, dropout, last_batch_norm,
                 mid_batch_norm, propagation_depth, readout_aggregators, readout_hidden_dim, readout_layers,
                 aggregators, scalers, avg_d, residual, posttrans_layers, pretrans_layers, device, edge_hidden_dim,
                 graph_norm, use_3d=False, gru_enable=False, divide_input_last=True, divide_input_first=True,
                 edge_feat=True, towers=1, **kwargs):
        super().__init__()
        self.node_gnn = PNAGNNOriginal(hidden_dim=hidden_dim, last_layer_dim=last_layer_dim,
                                       last_batch_norm=last_batch_norm, mid_batch_norm=mid_batch_norm,
                                       in_feat_dropout=in_feat_dropout, dropout=dropout, aggregators=aggregators,
                                       scalers=scalers, residual=residual, avg_d=avg_d,
                                       propagation_depth=propagation_depth, posttrans_layers=posttrans_layers,
                                       device=device, pretrans_layers=pretrans_layers, gru_enable=gru_enable,
                                       use_3d=use_3d, edge_hidden_dim=edge_hidden_dim,
                                       divide_input_first=divide_input_first, divide_input_last=divide_input_last,
                                       edge_feat=edge_feat, graph_norm=graph_norm, towers=towers)
        self.readout_aggregators = readout_aggregators
        self._readout_codes = [ops.AGG[a] for a in readout_aggregators]
        self.output = MLPReadout(last_layer_dim * len(self.readout_aggregators), target_dim)

    def forward(self, g, snorm_n):
        g = as_batched_graph(g)
        # (under an outer tape the stacked copies would not be leaves of THAT tape: the towers' parameters would silently get no
        # gradient - the per-tower path runs on the parameters themselves)
        stacks = _stacks_for(self) if (TOWER_STACK and g.ndata['feat'].is_cuda and
                                       not (tape.active() is not None and torch.is_grad_enabled())) else None
        if stacks is None:          # per-tower path (any structure; also the cross-check of the stacked path in the tests)
            g, h = self.node_gnn(g, g.ndata['feat'], g.edata['feat'], snorm_n)
            readout = ReadoutFn.apply(h, g.index(), self._readout_codes)
            return self.output(readout)
        params = [p for p in tape._param_list(self) if p.requires_grad]
        run = lambda: self._forward_stacked(g, snorm_n, stacks)        # noqa: E731
        if not torch.is_grad_enabled() or not params or tape.active() is not None:
            stacks.pack()
            out = run()
            if self.training:
                stacks.unpack_stats()
            return out
        return _StackedModelFn.apply(self, run, stacks, *params)

    def _forward_stacked(self, g, snorm_n, stacks):
        """The model with every layer's towers as ONE wide layer (_TowerStacks): per layer an edge block
        [h_src | h_dst | e] -> [E, towers * F_t], one aggregation, one posttrans block on [h | agg] -> [N, towers * F_out]
        (BatchNorm over the concatenated columns = the towers' BatchNorms side by side), graph norm, mixing network."""
        gnn = self.node_gnn
        idx = g.index()
        h = gnn.embedding_h(g.ndata['feat'])
        h = _dropout(h, gnn.in_feat_dropout.p, gnn.training)
        e_sorted = gnn.embedding_e(g.edata['feat'], perm=idx.perm) if gnn.edge_feat else None     # destination-sorted
        # widths rounded up to 4 floats (_LayerStack): zeros in the extra columns from here to the last layer's output
        st0 = stacks.layers[0]
        if h.shape[1] != st0.Dp:
            h = tape.apply(_CopyColsFn, h, st0.Dp)
        if e_sorted is not None and e_sorted.shape[1] != st0.Fep:
            e_sorted = tape.apply(_CopyColsFn, e_sorted, st0.Fep)
        snorm = snorm_n.to(h.device)
        snorm_flat = None
        for layer, st in zip(gnn.layers, stacks.layers):
            tw = layer.towers[0]
            assert h.shape[1] == st.Dp
            # (an eval-mode layer under a tape may still be differentiated: that backward is sequenced by the block path only)
            native = (TOWER_NATIVE and (layer.training or tape.active() is None) and ops.GEMM_WORKSPACE_BYTES > 0
                      and idx.num_edges > 0       # (a batch without bonds: the block path handles E = 0)
                      and (not st.fold or _fold_fits(idx, tw)))      # (> 32 distinct in-degrees: the block path has no such table)
            if native:
                if tw.graph_norm and snorm_flat is None:
                    snorm_flat = snorm.reshape(-1).contiguous().float()
                h = tape.apply(_TowerLayerFn, h, e_sorted if tw.edge_features else None, snorm_flat if tw.graph_norm else None,
                               st.Wp, st.bp, st.Wq, st.bq, st.gamma, st.beta, st.Wm, st.bm, idx, st, layer, layer.training)
                if layer.training:
                    for c in st.counters:
                        _layers._bump(c)
                continue
            msg = tape.apply(EdgeFCFn, h, e_sorted if tw.edge_features else None, st.Wp, st.bp, None, None, idx, st.pre_spec, None)
            if st.fold:       # identity blocks only, the scalers in per-degree weights (the columns of st.Wq are laid out for it)
                agg = tape.apply(AggregateFn, msg, idx, tw.aggregators, [ops.SCALER['identity']], float(tw.avg_d), False,
                                 st.Fip if st.tower_major else 0)
                coef = [[_scaler_coef(sc, deg, float(tw.avg_d)) for sc in tw.scalers] for deg, _, _ in idx.degree_groups()[2]]
                x = tape.apply(GroupedConcat2FCFn, h, agg, st.Wq, st.bq, st.gamma, st.beta, None, idx, coef, st.post_spec(layer.training))
            else:
                agg = tape.apply(AggregateFn, msg, idx, tw.aggregators, tw.scalers, float(tw.avg_d), True, st.Fip if st.tower_major else 0)
                x = tape.apply(Concat2FCFn, h, agg, st.Wq, st.bq, st.gamma, st.beta, None, st.post_spec(layer.training))
            if layer.training:
                for c in st.counters:
                    _layers._bump(c)
            if tw.graph_norm:
                x = tape.apply(_RowScaleFn, x, snorm)
            h = tape.apply(FCFn, x, st.Wm, st.bm, None, None, h if layer.residual else None, FCSpec('leakyrelu', None))
        if h.shape[1] != stacks.layers[-1].Mix:
            h = tape.apply(_CopyColsFn, h, stacks.layers[-1].Mix)
        g.ndata['feat'] = h
        readout = tape.apply(ReadoutFn, h, idx, self._readout_codes)
        return self.output(readout)


# ---- all towers of a layer as one wide layer -----------------------------------------------------------------------------
class _LayerStack:
    """Stacked buffers of ONE PNALayer (reference models/pna_original.py:264-319) and the block list that fills them.

    T towers, F_i inputs / F_o outputs per tower, D = the layer's input width, F_e edge features, B = aggregators x scalers.
    Every width is rounded up to a multiple of 4 floats (the yml's hidden_dim 90, edge_hidden_dim 70, 18 columns per tower are
    none) so that every kernel of the layer takes its 16-byte form: Dp = pad(D) (the activations between the layers carry
    zeros in the extra columns), Fep = pad(F_e), Fip = pad(F_i), Fop = pad(F_o) PER TOWER (a tower's columns never share a
    float4 with its neighbour's), Mixp = pad(out_dim).  Zeros in the extra rows / columns of every stacked tensor: the extra
    message / output columns are exactly zero in both directions and nothing is read back from them.
      Wp [T Fip, 2 Dp + Fep]   rows of tower t = its pretrans Linear; columns: [h_src | h_dst | e] - with divide_input a tower reads
                               only columns t F_i .. of h (zeros elsewhere)
      Wq [T Fop, Dp + B T Fip] rows of tower t = its posttrans Linear; the aggregation of the stacked messages is
                               [block (scaler, aggregator)][tower][feature], a tower's B column blocks are scattered accordingly
      Wm [Mixp, T Fop]         the mixing network (models/pna_original.py:291), its input columns at the towers' padded positions
      bp, bq, gamma, beta, running_mean, running_var, bm: the vectors side by side.
    `values` / `grads`: one flat buffer each, the tensors above are views."""

    def __init__(self, layer, device):
        towers = list(layer.towers)
        T, Fi, Fo, D = len(towers), layer.input_tower, layer.output_tower, layer.in_dim
        fc_pre, fc_post = towers[0].pretrans.fully_connected[0], towers[0].posttrans.fully_connected[0]
        Fe = fc_pre.in_dim - 2 * Fi
        B = len(towers[0].aggregators) * len(towers[0].scalers)
        assert fc_post.in_dim == (B + 1) * Fi and fc_pre.out_dim == Fi and fc_post.out_dim == Fo
        self.has_bn = fc_post.batch_norm is not None
        pad = (lambda n: (n + 3) & ~3) if PAD_WIDTHS else (lambda n: n)          # noqa: E731
        pad4 = lambda n: (n + 3) & ~3                                            # noqa: E731
        Dp, Fep, Fip, Fop, Mix = pad(D), pad(Fe), pad(Fi), pad(Fo), layer.out_dim
        Mixp = pad(Mix)
        self.D, self.Dp, self.Fe, self.Fep, self.Mix, self.Mixp = D, Dp, Fe, Fep, Mix, Mixp
        # the aggregated columns tower-major ([tower][block][feature]): a tower's B blocks are one K range, the posttrans products on
        # them run as T diagonal blocks (csrc/tower.hip, i3d_gemm_f32_batched) - needs the per-tower widths padded
        self.T, self.Fip = T, Fip
        # (csrc/grouped.hip builds the per-degree weights with 16-byte accesses)
        self.fold = TOWER_FOLD and Dp % 4 == 0 and (T * Fip) % 4 == 0
        self.tower_major = TOWER_BLOCKS and T > 1 and Fip % 4 == 0 and Fop % 4 == 0
        Mp, Kp, Mq, Kq = T * Fip, 2 * Dp + Fep, T * Fop, Dp + B * T * Fip
        ldp, ldq, ldm = pad4(Kp), pad4(Kq), Mq        # (csrc/tower.hip takes the mixing weights contiguous)
        sizes = [Mp * ldp, pad4(Mp), Mq * ldq, pad4(Mq), pad4(Mq), pad4(Mq), pad4(Mixp * ldm), pad4(Mixp)]
        offs = [0]
        for n in sizes:
            offs.append(offs[-1] + n)
        self.values = torch.zeros(offs[-1], dtype=torch.float32, device=device)
        self.grads = torch.zeros(offs[-1], dtype=torch.float32, device=device)
        self.stats = torch.zeros(2 * pad4(Mq), dtype=torch.float32, device=device)

        def views(buf):
            Wp = buf[offs[0]:offs[1]].view(Mp, ldp)[:, :Kp]
            bp = buf[offs[1]:offs[1] + Mp]
            Wq = buf[offs[2]:offs[3]].view(Mq, ldq)[:, :Kq]
            bq = buf[offs[3]:offs[3] + Mq]
            gamma = buf[offs[4]:offs[4] + Mq] if self.has_bn else None
            beta = buf[offs[5]:offs[5] + Mq] if self.has_bn else None
            Wm = buf[offs[6]:offs[7]].view(Mixp, ldm)[:, :Mq]
            bm = buf[offs[7]:offs[7] + Mixp]
            return Wp, bp, Wq, bq, gamma, beta, Wm, bm
        self.Wp, self.bp, self.Wq, self.bq, self.gamma, self.beta, self.Wm, self.bm = views(self.values)
        self.g_views = views(self.grads)
        self.rmean, self.rvar = self.stats[:Mq], self.stats[pad4(Mq):pad4(Mq) + Mq]
        self.leaves = [v for v in (self.Wp, self.bp, self.Wq, self.bq, self.gamma, self.beta, self.Wm, self.bm) if v is not None]
        self.leaf_grads = [v for v in self.g_views if v is not None]
        self.bias_of = {id(self.Wp): self.bp, id(self.Wq): self.bq, id(self.Wm): self.bm}
        self.pre_spec = FCSpec(fc_pre.activation, None)
        self._post_act = fc_post.activation
        bn = fc_post.batch_norm
        self._bn = (bn.momentum, bn.eps) if bn is not None else None
        self._specs = {}
        self.counters = [t.posttrans.fully_connected[0].batch_norm.num_batches_tracked for t in towers] if self.has_bn else []
        # ---- block list: (parameter, element offset in it, rows, cols, its row pitch, destination view, dest row, dest col)
        self.param_blocks, self.stat_blocks, self.params = [], [], []
        for t, tw in enumerate(towers):
            pre, post = tw.pretrans.fully_connected[0], tw.posttrans.fully_connected[0]
            W, b = pre.linear.weight, pre.linear.bias
            c0 = t * Fi if layer.divide_input else 0
            self.param_blocks += [(W, 0, Fi, Fi, Kp_t(W), 'Wp', t * Fip, c0), (W, Fi, Fi, Fi, Kp_t(W), 'Wp', t * Fip, Dp + c0)]
            if Fe:
                self.param_blocks.append((W, 2 * Fi, Fi, Fe, Kp_t(W), 'Wp', t * Fip, 2 * Dp))
            self.param_blocks.append((b, 0, 1, Fi, Fi, 'bp', 0, t * Fip))
            W2, b2 = post.linear.weight, post.linear.bias
            self.param_blocks.append((W2, 0, Fo, Fi, Kp_t(W2), 'Wq', t * Fop, c0))
            for k in range(B):
                nA = len(towers[0].aggregators)
                if self.tower_major and self.fold:      # [scaler][tower][aggregator][feature]
                    col = Dp + (k // nA) * (T * nA * Fip) + t * (nA * Fip) + (k % nA) * Fip
                elif self.tower_major:                  # [tower][block][feature]
                    col = Dp + t * B * Fip + k * Fip
                else:                                   # [block][tower][feature]
                    col = Dp + k * T * Fip + t * Fip
                self.param_blocks.append((W2, Fi + k * Fi, Fo, Fi, Kp_t(W2), 'Wq', t * Fop, col))
            self.param_blocks.append((b2, 0, 1, Fo, Fo, 'bq', 0, t * Fop))
            self.params += [W, b, W2, b2]
            if self.has_bn:
                m = post.batch_norm
                self.param_blocks += [(m.weight, 0, 1, Fo, Fo, 'gamma', 0, t * Fop), (m.bias, 0, 1, Fo, Fo, 'beta', 0, t * Fop)]
                self.stat_blocks += [(m.running_mean, 0, 1, Fo, Fo, 'rmean', 0, t * Fop), (m.running_var, 0, 1, Fo, Fo, 'rvar', 0, t * Fop)]
                self.params += [m.weight, m.bias]
        # the mixing network reads the towers' outputs at their padded positions
        Wm, bm = layer.mixing_network.weight, layer.mixing_network.bias
        assert tuple(Wm.shape) == (Mix, T * Fo)
        for t in range(T):
            self.param_blocks.append((Wm, t * Fo, Mix, Fo, Kp_t(Wm), 'Wm', 0, t * Fop))
        self.param_blocks.append((bm, 0, 1, Mix, Mix, 'bm', 0, 0))
        self.params += [Wm, bm]

    def post_spec(self, training):
        sp = self._specs.get(training)
        if sp is None:
            bn = BNSpec(self.rmean, self.rvar, None, self._bn[0], self._bn[1], training) if self._bn is not None else None
            sp = self._specs[training] = FCSpec(self._post_act, bn)
        return sp

    def dest(self, name, grads=False):
        if name in ('rmean', 'rvar'):
            return getattr(self, name)
        i = ('Wp', 'bp', 'Wq', 'bq', 'gamma', 'beta', 'Wm', 'bm').index(name)
        return (self.g_views if grads else (self.Wp, self.bp, self.Wq, self.bq, self.gamma, self.beta, self.Wm, self.bm))[i]


def Kp_t(W):
    return W.stride(0) if W.dim() == 2 else W.shape[0]


class _TowerStacks:
    """The stacked layers of one PNAOriginal and the three copy tables (device memory, built once): parameters -> stacked
    values, running statistics <-> stacked statistics, stacked gradients -> per-parameter gradient buffers."""

    def __reduce__(self):          # copy.deepcopy(model) / pickling: the copy starts without stacks (_stacks_for rebuilds them
        return (type(None), ())    # from ITS parameters: the copy tables hold device pointers)

    def __init__(self, model, device):
        self.layers = [_LayerStack(layer, device) for layer in model.node_gnn.layers]
        self.params = [p for st in self.layers for p in st.params]
        self.param_ids = {id(p) for p in self.params}
        self.key = tuple((id(p), p.data_ptr()) for p in self.params) + tuple(
            (id(b[0]), b[0].data_ptr()) for st in self.layers for b in st.stat_blocks)
        # persistent gradient buffers of the towers' parameters: views of one flat tensor
        sizes = [p.numel() for p in self.params]
        self.pgrad_flat = torch.zeros(sum(sizes), dtype=torch.float32, device=device)
        self.pgrad = {id(p): v.view_as(p) for p, v in zip(self.params, self.pgrad_flat.split(sizes))}
        self.leaves = [v for st in self.layers for v in st.leaves]
        self.leaf_grads = [v for st in self.layers for v in st.leaf_grads]

        def table(entries):
            arr = (_lib.CopyBlock * max(len(entries), 1))()
            for i, (src_ptr, dst_ptr, rows, cols, lds, ldd) in enumerate(entries):
                arr[i].src, arr[i].dst, arr[i].rows, arr[i].cols, arr[i].ld_src, arr[i].ld_dst = src_ptr, dst_ptr, rows, cols, lds, ldd
            raw = bytes(arr)[:ctypes.sizeof(_lib.CopyBlock) * len(entries)]
            dev = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(device) if entries else None
            return dev, len(entries)

        def entry(block, st, src_of, grads):
            t, off, rows, cols, ld, name, r0, c0 = block
            d = st.dest(name, grads)
            ldd = d.stride(0) if d.dim() == 2 else d.shape[0]
            dptr = d.data_ptr() + 4 * (r0 * ldd + c0)
            return (src_of(t).data_ptr() + 4 * off, dptr, rows, cols, ld, ldd)
        self.t_pack = table([entry(b, st, lambda t: t, False) for st in self.layers for b in st.param_blocks])
        self.t_stats = table([entry(b, st, lambda t: t, False) for st in self.layers for b in st.stat_blocks])
        self.t_grads = table([entry(b, st, lambda t: self.pgrad[id(t)], True) for st in self.layers for b in st.param_blocks])
        self._pool = None
        self.other_grad = {}         # id(parameter outside the towers) -> its persistent gradient buffer

    def _copy(self, tab, reverse):
        dev, n = tab
        if n:
            _lib.check(_lib.load().i3d_block_copy(dev.data_ptr(), n, int(reverse), ops._stream()), 'i3d_block_copy')

    def pack(self):
        self._copy(self.t_pack, False)
        self._copy(self.t_stats, False)

    def unpack_stats(self):
        self._copy(self.t_stats, True)

    def unpack_grads(self):
        self._copy(self.t_grads, True)

    def grad_pool(self, others=(), persistent_others=True):
        """what tape.grad_like / grad_for_bias_of consult during the backward pass: the stacked leaves' gradients are written
        straight into the stacked gradient buffer, the other parameters' (`others`: encoders, mixing networks, head) into
        persistent buffers of their own - the optimizer's one-launch kernel keeps its pointer table from step to step.
        persistent_others=False (a `.grad` may still alias those buffers): only the stacked leaves are pooled, every other
        gradient is a new tensor."""
        if not persistent_others:
            pool = tape._GradPool.__new__(tape._GradPool)
            pool.key = ()
            pool.view_of = {id(v): g for v, g in zip(self.leaves, self.leaf_grads)}
            pool.bias_of = {k: b for st in self.layers for k, b in st.bias_of.items()}
            pool.used = set()
            return pool
        pool = self._pool
        if pool is None or any(id(p) not in self.other_grad for p in others):
            for p in others:
                if id(p) not in self.other_grad:
                    self.other_grad[id(p)] = torch.zeros_like(p)
            pool = self._pool = tape._GradPool.__new__(tape._GradPool)
            pool.key = ()
            pool.view_of = {id(v): g for v, g in zip(self.leaves, self.leaf_grads)}
            pool.view_of.update(self.other_grad)
            pool.bias_of = {k: b for st in self.layers for k, b in st.bias_of.items()}
            pool.used = set()
        pool.used.clear()
        return pool


def _stackable(model):
    gnn = model.node_gnn
    if getattr(gnn, 'gru_enable', False):
        return False
    if any(tw.use_3d for layer in gnn.layers for tw in layer.towers):
        return False          # (a per-edge input column: the per-tower path)
    if any(tw.dropout.p > 0 for layer in gnn.layers for tw in layer.towers):
        return False          # (a mask per tower, drawn in the towers' order: the per-tower path keeps the reference's random stream)
    for layer in gnn.layers:
        tws = list(layer.towers)
        for tw in tws:
            if len(tw.pretrans.fully_connected) != 1 or len(tw.posttrans.fully_connected) != 1:
                return False
            if tw.pretrans.fully_connected[0].batch_norm is not None:
                return False
            if (tw.aggregators, tw.scalers, tw.avg_d, tw.graph_norm, tw.edge_features) != (
                    tws[0].aggregators, tws[0].scalers, tws[0].avg_d, tws[0].graph_norm, tws[0].edge_features):
                return False
    return True


def _stacks_for(model):
    """the model's _TowerStacks (built on first use; rebuilt when a parameter / buffer object or its storage has changed: .to(),
    load of a checkpoint into new tensors, a re-assigned Parameter), or None when the structure is not covered"""
    st = model.__dict__.get('_i3d_stacks')
    if st is False:
        return None
    if st is not None:
        cur = tuple((id(p), p.data_ptr()) for p in st.params) + tuple(
            (id(b[0]), b[0].data_ptr()) for ls in st.layers for b in ls.stat_blocks)
        owners_ok = all(own.get(name) is obj for own, name, obj in st.owners)
        if cur == st.key and owners_ok:
            return st
    if not _stackable(model):
        model.__dict__['_i3d_stacks'] = False
        return None
    dev = next(model.parameters()).device
    st = _TowerStacks(model, dev)
    st.owners = []
    for m in model.node_gnn.modules():
        for name, p in m._parameters.items():
            if p is not None and id(p) in st.param_ids:
                st.owners.append((m._parameters, name, p))
        for name, b in m._buffers.items():
            if b is not None and name in ('running_mean', 'running_var'):
                st.owners.append((m._buffers, name, b))
    model.__dict__['_i3d_stacks'] = st
    return st


# I3D_TOWER_NATIVE=0: the stacked layer as five block Functions sequenced from Python instead of one C call per direction
TOWER_NATIVE = True


def _fold_fits(idx, tw):
    """the per-degree tables of I3dTowerLayerArgs hold 32 in-degree groups / 128 (group, scaler) coefficients"""
    n = len(idx.degree_groups()[2])
    return n <= 32 and n * len(tw.scalers) <= 128


class _TowerLayerFn(torch.autograd.Function):
    """A stacked PNALayer from ONE C call per direction (csrc/tower.hip: i3d_tower_layer_fwd / _bwd) - the kernels and the
    operands of the five blocks _forward_stacked sequences from Python (edge block, aggregation, concat block with BatchNorm,
    row scale, mixing block), with the layout of saved activations and scratch computed in C."""

    @staticmethod
    def forward(ctx, h, e, snorm, Wp, bp, Wq, bq, gamma, beta, Wm, bm, idx, st, layer, training):
        h = h.contiguous()
        dev = h.device
        tw = layer.towers[0]
        a = _lib.TowerLayerArgs()
        a.num_nodes, a.num_edges, a.f_in = h.shape[0], idx.num_edges, h.shape[1]
        a.f_edge = e.shape[1] if e is not None else 0
        a.f_msg, a.f_out, a.f_mix = Wp.shape[0], Wq.shape[0], Wm.shape[0]
        a.ldp, a.ldq = Wp.stride(0), Wq.stride(0)
        a.n_aggregators, a.n_scalers = len(tw.aggregators), len(tw.scalers)
        for i, v in enumerate(tw.aggregators):
            a.aggregators[i] = v
        for i, v in enumerate(tw.scalers):
            a.scalers[i] = v
        a.avg_d_log = float(tw.avg_d)
        a.residual, a.training = int(layer.residual), int(training)
        a.n_towers = st.T if st.tower_major else 0
        if st.fold:
            rows, tiles, groups = idx.degree_groups()
            nS = len(tw.scalers)
            assert _fold_fits(idx, tw), 'the caller (_forward_stacked) sends such a batch through the block path'
            a.n_deg_groups, a.m_padded = len(groups), rows.shape[0]
            for gi, (deg, start, count) in enumerate(groups):
                a.group_start[gi], a.group_count[gi] = start, count
                for si, sc in enumerate(tw.scalers):
                    a.coef[gi * nS + si] = _scaler_coef(sc, deg, float(tw.avg_d))
                a.deg_rows, a.deg_tile_group = rows.data_ptr(), tiles.data_ptr()
        a.h, a.e = h.data_ptr(), (e.data_ptr() if e is not None else None)
        a.snorm = snorm.data_ptr() if snorm is not None else None
        a.Wp, a.bp, a.Wq, a.bq = Wp.data_ptr(), bp.data_ptr(), Wq.data_ptr(), bq.data_ptr()
        if gamma is not None:
            a.gamma, a.beta = gamma.data_ptr(), beta.data_ptr()
            a.running_mean, a.running_var = st.rmean.data_ptr(), st.rvar.data_ptr()
            a.momentum, a.eps = st._bn
        a.Wm, a.bm = Wm.data_ptr(), bm.data_ptr()
        a.src_s, a.dst_s, a.in_ptr = idx.src_s.data_ptr(), idx.dst_s.data_ptr(), idx.in_ptr.data_ptr()
        a.out_ptr, a.out_epos = idx.out_ptr.data_ptr(), idx.out_epos.data_ptr()
        L = _lib.load()
        saved = torch.empty(L.i3d_tower_layer_saved_floats(ctypes.byref(a)), dtype=torch.float32, device=dev)
        scratch = torch.empty(L.i3d_tower_layer_scratch_floats(ctypes.byref(a)), dtype=torch.float32, device=dev)
        out = torch.empty(h.shape[0], Wm.shape[0], dtype=torch.float32, device=dev)
        a.saved, a.scratch, a.out = saved.data_ptr(), scratch.data_ptr(), out.data_ptr()
        a.workspace = ops._workspace(max(a.f_msg, a.f_out), dev).data_ptr()
        _lib.check(L.i3d_tower_layer_fwd(ctypes.byref(a), ops._stream()), 'i3d_tower_layer_fwd')
        ctx.args, ctx.keep = a, (h, e, snorm, Wp, bp, Wq, bq, gamma, beta, Wm, bm, saved)
        ctx.idx = idx                   # (its degree tables are read by the backward pass)
        ctx.has_e_grad = e is not None and ctx.needs_input_grad[1]
        return out

    @staticmethod
    def backward(ctx, grad_out):
        a = ctx.args
        h, e, snorm, Wp, bp, Wq, bq, gamma, beta, Wm, bm, saved = ctx.keep
        dev = h.device
        grad_out = grad_out.contiguous()
        L = _lib.load()
        scratch = torch.empty(L.i3d_tower_layer_scratch_floats(ctypes.byref(a)), dtype=torch.float32, device=dev)
        gh = torch.empty_like(h)
        ge = torch.empty_like(e) if ctx.has_e_grad else None
        gWp, gbp, gWq, gbq = tape.grad_like(Wp), tape.grad_like(bp), tape.grad_like(Wq), tape.grad_like(bq)
        gg = tape.grad_like(gamma) if gamma is not None else None
        gb = tape.grad_like(beta) if beta is not None else None
        gWm, gbm = tape.grad_like(Wm), tape.grad_like(bm)
        a.scratch, a.grad_out, a.grad_h = scratch.data_ptr(), grad_out.data_ptr(), gh.data_ptr()
        a.grad_e, a.grad_e_accumulate = (ge.data_ptr() if ge is not None else None), 0
        a.grad_Wp, a.grad_bp, a.grad_Wq, a.grad_bq = gWp.data_ptr(), gbp.data_ptr(), gWq.data_ptr(), gbq.data_ptr()
        a.ldgp, a.ldgq = gWp.stride(0), gWq.stride(0)
        a.grad_gamma = gg.data_ptr() if gg is not None else None
        a.grad_beta = gb.data_ptr() if gb is not None else None
        a.grad_Wm, a.grad_bm = gWm.data_ptr(), gbm.data_ptr()
        a.workspace = ops._workspace(max(a.f_msg, a.f_out), dev).data_ptr()        # per thread and stream (autograd's thread here)
        a.gemm_workspace, a.gemm_workspace_bytes = ops._gemm_workspace(dev).data_ptr(), ops.GEMM_WORKSPACE_BYTES
        _lib.check(L.i3d_tower_layer_bwd(ctypes.byref(a), ops._stream()), 'i3d_tower_layer_bwd')
        return gh, ge, None, gWp, gbp, gWq, gbq, gg, gb, gWm, gbm, None, None, None, None


class _StackedModelFn(torch.autograd.Function):
    """PNAOriginal with stacked towers as ONE autograd node (tape.ModelFn with the stacked views as additional leaves): pack the
    towers' parameters, run the blocks under a tape; backward: walk the tape (weight gradients land in the stacked gradient
    buffer), scatter them to the towers' own gradient buffers with one launch."""

    @staticmethod
    def forward(ctx, module, run, stacks, *params):
        stacks.pack()
        leaves = [p for p in params if id(p) not in stacks.param_ids] + stacks.leaves
        tp = tape.Tape(leaves)
        prev = getattr(tape._tls, 'tape', None)
        tape._tls.tape = tp
        try:
            with _layers.bn_counter_scope():
                out = run()
        finally:
            tape._tls.tape = prev
        tp.release(out)
        if module.training:
            stacks.unpack_stats()
        ctx.tape, ctx.out_id, ctx.params, ctx.stacks = tp, id(out), params, stacks
        return out

    @staticmethod
    def backward(ctx, grad):
        stacks = ctx.stacks
        others = [p for p in ctx.params if id(p) not in stacks.param_ids]
        # The persistent per-parameter buffers (stacks.pgrad, stacks.other_grad) may be handed out only when nothing can still be
        # looking at them: every `.grad` empty, no hooks (tape.ModelFn's rule).  After such a step `p.grad` IS the buffer; with
        # zero_grad(set_to_none=False), gradient accumulation or two forwards before one backward the next pass must not write
        # into it - autograd would then add the new gradient to a buffer that already holds it (2 g_new instead of g_old + g_new).
        direct = tape.DIRECT_PARAM_GRADS and tape._plain_leaves(ctx.params)
        tape._tls.pool = stacks.grad_pool(others if direct else (), persistent_others=direct)
        try:
            grads = ctx.tape.backward(ctx.out_id, grad.contiguous())
        finally:
            tape._tls.pool = None
        for v, gv in zip(stacks.leaves, stacks.leaf_grads):
            ent = grads.get(id(v))
            if ent is None:
                gv.zero_()
            elif ent[0].data_ptr() != gv.data_ptr():
                gv.copy_(ent[0])
        if direct:
            stacks.unpack_grads()
            pgrad = stacks.pgrad
        else:
            # the copy table's destination is the persistent flat buffer, which a `.grad` may alias: scatter into it, take a copy,
            # put back what it held
            held = stacks.pgrad_flat.clone()
            stacks.unpack_grads()
            fresh = stacks.pgrad_flat.clone()
            stacks.pgrad_flat.copy_(held)
            sizes = [p.numel() for p in stacks.params]
            pgrad = {id(p): v.view_as(p) for p, v in zip(stacks.params, fresh.split(sizes))}
        out, dsts, srcs = [], [], []
        for p in ctx.params:
            if id(p) in stacks.param_ids:
                out.append(pgrad[id(p)])
            else:
                ent = grads.get(id(p))
                if ent is None:
                    out.append(None)
                    continue
                if not direct:
                    out.append(ent[0])
                    continue
                keep = stacks.other_grad[id(p)]
                if ent[0].data_ptr() != keep.data_ptr():       # a gradient that was not written through tape.grad_like (the encoders' tables)
                    dsts.append(keep)
                    srcs.append(ent[0])
                out.append(keep)
        if dsts:
            torch._foreach_copy_(dsts, srcs)
        if direct:
            for p, g in zip(ctx.params, out):
                if g is not None:
                    p.grad = g
            return (None,) * (3 + len(ctx.params))
        return (None, None, None) + tuple(out)


class PNAGNNOriginal(nn.Module):
    """reference models/pna_original.py:149-194."""

    def __init__(self, hidden_dim, last_layer_dim, in_feat_dropout, dropout, propagation_depth, graph_norm,
                 mid_batch_norm, last_batch_norm, residual, aggregators, scalers, avg_d, use_3d, towers,
                 divide_input_first, divide_input_last, edge_feat, edge_hidden_dim, pretrans_layers, posttrans_layers,
                 gru_enable, device):
        super().__init__()
        _check_unsupported(dropout, in_feat_dropout, gru_enable, use_3d)
        self.gru_enable = gru_enable
        self.edge_feat = edge_feat
        self.in_feat_dropout = nn.Dropout(in_feat_dropout)
        self.embedding_h = AtomEncoder(hidden_dim)
        if self.edge_feat:
            self.embedding_e = BondEncoder(edge_hidden_dim)
        common = dict(dropout=dropout, graph_norm=graph_norm, mid_batch_norm=mid_batch_norm,
                      last_batch_norm=last_batch_norm, use_3d=use_3d, residual=residual, aggregators=aggregators,
                      scalers=scalers, avg_d=avg_d, towers=towers, edge_features=edge_feat,
                      edge_hidden_dim=edge_hidden_dim, pretrans_layers=pretrans_layers,
                      posttrans_layers=posttrans_layers)
        self.layers = nn.ModuleList([PNALayer(in_dim=hidden_dim, out_dim=hidden_dim, divide_input=divide_input_first,
                                              **common) for _ in range(propagation_depth - 1)])
        self.layers.append(PNALayer(in_dim=hidden_dim, out_dim=last_layer_dim, divide_input=divide_input_last, **common))
        if self.gru_enable:                              # reference :179-180
            self.gru = GRU(hidden_dim, hidden_dim, device)
        self.MLP_layer = MLPReadout(hidden_dim, 1)      # unused by forward, kept for state_dict parity (:179)

    def forward(self, g, h, e, snorm_n):
        g = as_batched_graph(g)
        idx = g.index()
        h = self.embedding_h(h)
        h = _dropout(h, self.in_feat_dropout.p, self.training)                        # :187
        e_sorted = self.embedding_e(e, perm=idx.perm) if self.edge_feat else None     # destination-sorted
        snorm = snorm_n.to(h.device)
        for i, conv in enumerate(self.layers):
            h_t = conv(g, h, e_sorted, snorm, edges_sorted=True)
            if self.gru_enable and i != len(self.layers) - 1:            # reference :190-193
                h_t = self.gru(h, h_t)
            h = h_t
        g.ndata['feat'] = h
        return g, h


class PNATower(nn.Module):
    """reference models/pna_original.py:197-261."""

    def __init__(self, in_dim, out_dim, dropout, graph_norm, mid_batch_norm, last_batch_norm, aggregators, scalers,
                 avg_d, use_3d, pretrans_layers, posttrans_layers, edge_features, edge_hidden_dim):
        super().__init__()
        _check_unsupported(dropout, 0.0, False, use_3d)
        self.dropout = nn.Dropout(dropout)
        self.graph_norm = graph_norm
        self.edge_features = edge_features
        self.use_3d = use_3d           # reference :215-216, 224-226: the end points' distance (ndata['x']) as one more pretrans input
        self.aggregators = _codes(aggregators, ops.AGG, 'aggregator')
        self.scalers = _codes(scalers, ops.SCALER, 'scaler')
        self.pretrans = MLP(in_dim=2 * in_dim + (edge_hidden_dim if edge_features else 0) + (1 if use_3d else 0), hidden_size=in_dim,
                            out_dim=in_dim, layers=pretrans_layers, mid_activation='relu', last_activation='none')
        self.posttrans = MLP(in_dim=(len(aggregators) * len(scalers) + 1) * in_dim, hidden_size=out_dim,
                             mid_batch_norm=mid_batch_norm, last_batch_norm=last_batch_norm, out_dim=out_dim,
                             layers=posttrans_layers, mid_activation='relu', last_activation='none')
        self.avg_d = avg_d

    def forward(self, g, h, e_sorted, snorm_n):
        idx = as_batched_graph(g).index()
        q = e_sorted if (self.edge_features or self.use_3d) else None      # (use_3d: PNALayer.forward appended the distance column)
        msg = self.pretrans.forward_edge(h, q, idx)                                               # :246
        agg = AggregateFn.apply(msg, idx, self.aggregators, self.scalers, float(self.avg_d), True)   # :249
        h = self.posttrans.forward_concat2(h, agg)                                                # :250-253
        if self.graph_norm:
            h = _RowScaleFn.apply(h, snorm_n)                                                     # :256-258
        return _dropout(h, self.dropout.p, self.training)                                         # :260


class PNALayer(nn.Module):
    """reference models/pna_original.py:264-319 (the ORIGINAL multi-tower layer; models/pna.py has its own PNALayer)."""

    def __init__(self, in_dim, out_dim, aggregators, scalers, avg_d, dropout, graph_norm, mid_batch_norm, use_3d,
                 last_batch_norm, towers=1, pretrans_layers=1, posttrans_layers=1, divide_input=True, residual=False,
                 edge_features=False, edge_hidden_dim=0):
        super().__init__()
        assert (not divide_input) or in_dim % towers == 0, \
            "if divide_input is set the number of towers has to divide in_dim"
        assert out_dim % towers == 0, "the number of towers has to divide the last_layer_dim"
        assert avg_d is not None
        self.divide_input = divide_input
        self.input_tower = in_dim // towers if divide_input else in_dim
        self.output_tower = out_dim // towers
        self.in_dim, self.out_dim = in_dim, out_dim
        self.edge_features = edge_features
        self.residual = residual and in_dim == out_dim
        self.towers = nn.ModuleList()
        for _ in range(towers):
            self.towers.append(PNATower(in_dim=self.input_tower, out_dim=self.output_tower, aggregators=aggregators,
                                        scalers=scalers, avg_d=avg_d, pretrans_layers=pretrans_layers,
                                        posttrans_layers=posttrans_layers, mid_batch_norm=mid_batch_norm,
                                        last_batch_norm=last_batch_norm, dropout=dropout, use_3d=use_3d,
                                        graph_norm=graph_norm, edge_features=edge_features,
                                        edge_hidden_dim=edge_hidden_dim))
        self.mixing_network = nn.Linear(out_dim, out_dim)
        self.mixing_act = nn.LeakyReLU()

    def forward(self, g, h, e, snorm_n, edges_sorted=False):
        g = as_batched_graph(g)
        if e is not None and not edges_sorted:          # stand-alone use with edge-id-ordered float features
            idx = g.index()
            e = _GatherRowsFn.apply(e, idx.perm, idx.inv_perm)
        snorm_n = snorm_n.to(h.device)
        it = self.input_tower
        if self.towers[0].use_3d:       # [e | |x_src - x_dst|], built once for the layer's towers
            e = _AppendSqDistFn.apply(e if self.edge_features else None, g.ndata['x'], g.index(), True)
        outs = [tower(g, h[:, t * it:(t + 1) * it].contiguous() if self.divide_input else h, e, snorm_n)
                for t, tower in enumerate(self.towers)]
        h_cat = torch.cat(outs, dim=1)
        h_out = FCFn.apply(h_cat, self.mixing_network.weight, self.mixing_network.bias, None, None,
                           h if self.residual else None, FCSpec('leakyrelu', None))               # :308-311
        return h_out

    def __repr__(self):
        return '{}(in_channels={}, out_channels={})'.format(self.__class__.__name__, self.in_dim, self.out_dim)


class PNAOriginalSimple(nn.Module):
    """reference models/pna_original.py:325-348."""

    def __init__(self, hidden_dim, last_layer_dim, target_dim, in_feat_dropout, dropout, last_batch_norm,
                 mid_batch_norm, propagation_depth, readout_aggregators, readout_hidden_dim, readout_layers,
                 aggregators, scalers, avg_d, residual, posttrans_layers, readout_batchnorm, batch_norm_momentum,
                 **kwargs):
        super().__init__()
        self.node_gnn = PNAGNNSimple(hidden_dim=hidden_dim, last_layer_dim=last_layer_dim,
                                     last_batch_norm=last_batch_norm, mid_batch_norm=mid_batch_norm,
                                     in_feat_dropout=in_feat_dropout, dropout=dropout, aggregators=aggregators,
                                     scalers=scalers, residual=residual, avg_d=avg_d,
                                     propagation_depth=propagation_depth, posttrans_layers=posttrans_layers)
        self.readout_aggregators = readout_aggregators
        self._readout_codes = [ops.AGG[a] for a in readout_aggregators]
        self.output = MLP(in_dim=hidden_dim * len(self.readout_aggregators), hidden_size=readout_hidden_dim,
                          mid_batch_norm=readout_batchnorm, out_dim=target_dim, layers=readout_layers,
                          batch_norm_momentum=batch_norm_momentum)

    def forward(self, g, *unused):
        g = as_batched_graph(g)
        g, h = self.node_gnn(g, g.ndata['feat'])
        readout = ReadoutFn.apply(h, g.index(), self._readout_codes)
        return self.output(readout)


class PNAGNNSimple(nn.Module):
    """reference models/pna_original.py:351-382."""

    def __init__(self, hidden_dim, last_layer_dim, in_feat_dropout, dropout, residual, aggregators, scalers, avg_d,
                 last_batch_norm, mid_batch_norm, propagation_depth, posttrans_layers):
        super().__init__()
        self.in_feat_dropout = nn.Dropout(in_feat_dropout)
        self.embedding_h = AtomEncoder(emb_dim=hidden_dim)
        common = dict(dropout=dropout, last_batch_norm=last_batch_norm, mid_batch_norm=mid_batch_norm,
                      residual=residual, aggregators=aggregators, scalers=scalers, avg_d=avg_d,
                      posttrans_layers=posttrans_layers)
        self.layers = nn.ModuleList([PNASimpleLayer(in_dim=hidden_dim, out_dim=hidden_dim, **common)
                                     for _ in range(propagation_depth - 1)])
        self.layers.append(PNASimpleLayer(in_dim=hidden_dim, out_dim=last_layer_dim, **common))
        self.output = MLPReadout(last_layer_dim, 1)     # unused by forward, kept for state_dict parity (:372)

    def forward(self, g, h):
        g = as_batched_graph(g)
        h = self.embedding_h(h)
        h = _dropout(h, self.in_feat_dropout.p, self.training)                        # :375
        for conv in self.layers:
            h = conv(g, h)
        g.ndata['feat'] = h
        return g, h


class PNASimpleLayer(nn.Module):
    """reference models/pna_original.py:384-431: copy_u messages (no edge MLP), posttrans on the aggregate only,
    ReLU after it, residual."""

    def __init__(self, in_dim, out_dim, aggregators, scalers, avg_d, dropout, last_batch_norm, mid_batch_norm, residual,
                 posttrans_layers=1):
        super().__init__()
        self.dropout = nn.Dropout(p=dropout)
        self.aggregators = _codes(aggregators, ops.AGG, 'aggregator')
        self.scalers = _codes(scalers, ops.SCALER, 'scaler')
        self.in_dim, self.out_dim = in_dim, out_dim
        self.residual = residual
        self.posttrans = MLP(in_dim=(len(aggregators) * len(scalers)) * in_dim, hidden_size=out_dim,
                             last_batch_norm=last_batch_norm, mid_batch_norm=mid_batch_norm, out_dim=out_dim,
                             layers=posttrans_layers, mid_activation='relu', last_activation='none')
        self.avg_d = avg_d

    def forward(self, g, h):
        idx = as_batched_graph(g).index()
        m = _GatherSrcFn.apply(h, idx)                              # copy_u: message = source features, dst-sorted
        agg = AggregateFn.apply(m, idx, self.aggregators, self.scalers, float(self.avg_d), True)
        # posttrans -> ReLU -> (+ h_in): the ReLU rides on the last BatchNorm, the residual is added after it
        res = h if self.residual else None       # NB the reference adds the residual even when in_dim != out_dim would fail
        return _dropout(self.posttrans(agg, residual=res, post_act='relu'), self.dropout.p, self.training)       # :428

    def __repr__(self):
        return '{}(in_channels={}, out_channels={})'.format(self.__class__.__name__, self.in_dim, self.out_dim)
