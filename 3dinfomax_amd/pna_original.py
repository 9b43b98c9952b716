"""Original-PNA variants on the MI355X kernels - drop-in for `PNAOriginal` / `PNAOriginalSimple` (and their
`PNAGNNOriginal`, `PNAGNNSimple`, `PNATower`, `PNALayer`, `PNASimpleLayer`, `MLPReadout` parts) of reference
models/pna_original.py + models/base_layers.py:149-164.  Same constructor kwargs, sub-module names and
state_dict keys.  They reuse the hot path's kernels: the segmented aggregation (K4) with the REAL `avg_d` and
always-applied scalers (reference :231-236 - no single-scaler quirk here), the edge gather-combine for the tower
pretrans, the two-segment posttrans GEMMs, `h * snorm_n` as a row-scale kernel, LeakyReLU mixing.

Not on the accelerated path (raise NotImplementedError; configs/pna_original.yml uses none of them):
gru_enable, use_3d, dropout > 0, moment aggregators.
"""
import torch
import torch.nn as nn

from . import ops
from .graph import as_batched_graph
from .layers import MLP, AggregateFn, FCFn, FCSpec, ReadoutFn
from .mol_encoder import AtomEncoder, BondEncoder
from .pna import _codes, _GatherRowsFn


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
            y = FCFn.apply(y, self.FC_layers[l].weight, self.FC_layers[l].bias, None, None, None, FCSpec('relu', None))
        return FCFn.apply(y, self.FC_layers[self.L].weight, self.FC_layers[self.L].bias, None, None, None,
                          FCSpec(None, None))


def _check_unsupported(dropout=0.0, in_feat_dropout=0.0, gru_enable=False, use_3d=False):
    if dropout or in_feat_dropout:
        raise NotImplementedError('dropout > 0 is not on the accelerated path')
    if gru_enable:
        raise NotImplementedError('gru_enable=True is not on the accelerated path')
    if use_3d:
        raise NotImplementedError('use_3d=True is not on the accelerated path')


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
        g, h = self.node_gnn(g, g.ndata['feat'], g.edata['feat'], snorm_n)
        readout = ReadoutFn.apply(h, g.index(), self._readout_codes)
        return self.output(readout)


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
        self.MLP_layer = MLPReadout(hidden_dim, 1)      # unused by forward, kept for state_dict parity (:179)

    def forward(self, g, h, e, snorm_n):
        g = as_batched_graph(g)
        idx = g.index()
        h = self.embedding_h(h)
        e_sorted = self.embedding_e(e, perm=idx.perm) if self.edge_feat else None     # destination-sorted
        snorm = snorm_n.to(h.device)
        for conv in self.layers:
            h = conv(g, h, e_sorted, snorm, edges_sorted=True)
        g.ndata['feat'] = h
        return g, h


class PNATower(nn.Module):
    """reference models/pna_original.py:197-261."""

    def __init__(self, in_dim, out_dim, dropout, graph_norm, mid_batch_norm, last_batch_norm, aggregators, scalers,
                 avg_d, use_3d, pretrans_layers, posttrans_layers, edge_features, edge_hidden_dim):
        super().__init__()
        _check_unsupported(dropout, 0.0, False, use_3d)
        self.graph_norm = graph_norm
        self.edge_features = edge_features
        self.aggregators = _codes(aggregators, ops.AGG, 'aggregator')
        self.scalers = _codes(scalers, ops.SCALER, 'scaler')
        self.pretrans = MLP(in_dim=2 * in_dim + (edge_hidden_dim if edge_features else 0), hidden_size=in_dim,
                            out_dim=in_dim, layers=pretrans_layers, mid_activation='relu', last_activation='none')
        self.posttrans = MLP(in_dim=(len(aggregators) * len(scalers) + 1) * in_dim, hidden_size=out_dim,
                             mid_batch_norm=mid_batch_norm, last_batch_norm=last_batch_norm, out_dim=out_dim,
                             layers=posttrans_layers, mid_activation='relu', last_activation='none')
        self.avg_d = avg_d

    def forward(self, g, h, e_sorted, snorm_n):
        idx = as_batched_graph(g).index()
        msg = self.pretrans.forward_edge(h, e_sorted if self.edge_features else None, idx)        # :246
        agg = AggregateFn.apply(msg, idx, self.aggregators, self.scalers, float(self.avg_d), True)   # :249
        h = self.posttrans.forward_concat2(h, agg)                                                # :250-253
        if self.graph_norm:
            h = _RowScaleFn.apply(h, snorm_n)                                                     # :256-258
        return h


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
        _check_unsupported(dropout, in_feat_dropout)
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
        _check_unsupported(dropout)
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
        return self.posttrans(agg, residual=res, post_act='relu')

    def __repr__(self):
        return '{}(in_channels={}, out_channels={})'.format(self.__class__.__name__, self.in_dim, self.out_dim)
