"""Net3D (the 3D distance-graph network) on the MI355X kernels - drop-in for reference models/net3d.py.

Same constructor kwargs (unknown ones swallowed, reference models/net3d.py:18), sub-module / parameter names
(`node_embedding`, `edge_input`, `mp_layers.{l}.message_network|update_network|soft_edge_network`, `output`) and
side effects (`ndata['feat']`, `edata['d']` overwritten).  Edge tensors are kept destination-sorted, so the
`fn.mean` reduce is a contiguous segmented mean; the [h_src | h_dst | d] concat + first Linear of the message
network is the node-level P trick of layers.EdgeFCFn.
"""
from typing import List

import torch
import torch.nn as nn

from . import net3d_native, ops, streams, tape
from .graph import as_batched_graph
from .layers import MLP, ReadoutFn, act_name, bn_counter_scope
from .mol_encoder import AtomEncoder


class _BroadcastRowFn(torch.autograd.Function):
    """h = node_embedding[None, :].expand(N, -1)  (reference models/net3d.py:61); backward = column sum."""

    @staticmethod
    def forward(ctx, emb, n):
        return ops.broadcast_row(emb.contiguous(), n)

    @staticmethod
    def backward(ctx, g):
        return ops.colsum(g.contiguous()), None


class _AddFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        return ops.add(a.contiguous(), b.contiguous())

    @staticmethod
    def backward(ctx, g):
        return g, g


class SoftEdgeFn(torch.autograd.Function):
    """w = sigmoid(Linear_{H->1}(m)); msg = m * w   (reference models/net3d.py:106, 117-118)."""

    @staticmethod
    def forward(ctx, m, ws, bs):
        m = m.contiguous()
        msg, w = ops.soft_edge_fwd(m, ws.contiguous(), bs.contiguous())
        ctx.save_for_backward(m, w, ws)
        return msg

    @staticmethod
    def backward(ctx, gmsg):
        m, w, ws = ctx.saved_tensors
        gm, gg = ops.soft_edge_bwd(gmsg.contiguous(), m, w, ws.contiguous())
        gws = ops.colsum(m, w=gg).view_as(ws)
        gbs = ops.colsum(gg.view(-1, 1))
        return gm, gws, gbs


class SegmentReduceFn(torch.autograd.Function):
    """DGL builtin fn.sum / fn.mean over in-edges (reference models/net3d.py:93-96, 109); edges destination-sorted."""

    @staticmethod
    def forward(ctx, msg, index, mean):
        ctx.cfg = (index, mean)
        return ops.segment_sum(msg.contiguous(), index.in_ptr, None, index.num_nodes, mean=mean)

    @staticmethod
    def backward(ctx, g):
        index, mean = ctx.cfg
        return ops.segment_bcast(g.contiguous(), index.in_ptr, index.dst_s, index.num_edges, mean=mean), None, None


class Net3DFn(torch.autograd.Function):
    """the whole network as one node (net3d_native.py)"""

    @staticmethod
    def forward(ctx, model, graph, *params):
        return net3d_native.forward(ctx, model, graph, params)

    @staticmethod
    def backward(ctx, grad):
        return (None, None) + net3d_native.backward(ctx, grad)


class Net3D(nn.Module):
    """reference models/net3d.py:14-81."""

    def __init__(self, node_dim, edge_dim, hidden_dim, target_dim, readout_aggregators: List[str], batch_norm=False,
                 node_wise_output_layers=2, readout_batchnorm=True, batch_norm_momentum=0.1, reduce_func='sum',
                 dropout=0.0, propagation_depth: int = 4, readout_layers: int = 2, readout_hidden_dim=None,
                 fourier_encodings=0, activation: str = 'SiLU', update_net_layers=2, message_net_layers=2,
                 use_node_features=False, **kwargs):
        super().__init__()
        self.fourier_encodings = fourier_encodings
        edge_in_dim = 1 if fourier_encodings == 0 else 2 * fourier_encodings + 1
        self.edge_input = MLP(in_dim=edge_in_dim, hidden_size=hidden_dim, out_dim=hidden_dim, mid_batch_norm=batch_norm,
                              last_batch_norm=batch_norm, batch_norm_momentum=batch_norm_momentum, layers=1,
                              mid_activation=activation, dropout=dropout, last_activation=activation)
        self.use_node_features = use_node_features
        if self.use_node_features:
            self.atom_encoder = AtomEncoder(hidden_dim)
        else:
            self.node_embedding = nn.Parameter(torch.empty((hidden_dim,)))
            nn.init.normal_(self.node_embedding)
        self.mp_layers = nn.ModuleList()
        for _ in range(propagation_depth):
            self.mp_layers.append(
                Net3DLayer(edge_dim=hidden_dim, hidden_dim=hidden_dim, batch_norm=batch_norm,
                           batch_norm_momentum=batch_norm_momentum, dropout=dropout, mid_activation=activation,
                           reduce_func=reduce_func, message_net_layers=message_net_layers,
                           update_net_layers=update_net_layers))
        self.node_wise_output_layers = node_wise_output_layers
        if self.node_wise_output_layers > 0:
            self.node_wise_output_network = MLP(in_dim=hidden_dim, hidden_size=hidden_dim, out_dim=hidden_dim,
                                                mid_batch_norm=batch_norm, last_batch_norm=batch_norm,
                                                batch_norm_momentum=batch_norm_momentum,
                                                layers=node_wise_output_layers, mid_activation=activation,
                                                dropout=dropout, last_activation='None')
        if readout_hidden_dim is None:
            readout_hidden_dim = hidden_dim
        self.readout_aggregators = readout_aggregators
        self._readout_codes = [ops.AGG[a] for a in readout_aggregators]
        self.output = MLP(in_dim=hidden_dim * len(self.readout_aggregators), hidden_size=readout_hidden_dim,
                          mid_batch_norm=readout_batchnorm, batch_norm_momentum=batch_norm_momentum, out_dim=target_dim,
                          layers=readout_layers)

    def forward(self, graph, *unused):
        g = as_batched_graph(graph)
        side = None
        if self.training and torch.is_grad_enabled() and g.device.type == 'cuda':
            side = streams.side_stream_for(getattr(g, 'ready_event', None), g.device)
        if side is None:
            with bn_counter_scope():
                return tape.run_model(self, lambda: self._forward(g))
        # next to the 2D network on a side stream (streams.py); the caller's stream waits before anything is handed back
        main = torch.cuda.current_stream(g.device)
        with torch.cuda.stream(side):
            with bn_counter_scope():
                z = tape.run_model(self, lambda: self._forward(g))
        for t in (z, g.ndata.get('feat'), g.edata.get('d')):
            if torch.is_tensor(t) and t.is_cuda:
                t.record_stream(main)
        main.wait_stream(side)
        return z

    def _forward(self, graph):
        g = as_batched_graph(graph)
        if tape.active() is not None and net3d_native.eligible(self, g):
            params = tape._param_list(self)
            return tape.apply(Net3DFn, self, g, *params)
        idx = g.index()
        if self.use_node_features:
            h = self.atom_encoder(g.ndata['feat'])
        else:
            h = tape.apply(_BroadcastRowFn, self.node_embedding, g.number_of_nodes())
        # distances: edge-id order -> destination-sorted order, then Fourier features (inputs: no gradient)
        d = g.edata['d']
        with torch.no_grad():
            d = ops.gather_rows(d.reshape(-1, 1).contiguous().float(), idx.perm)
            if self.fourier_encodings > 0:
                d = ops.fourier_encode(d.view(-1), self.fourier_encodings)
        # reference models/net3d.py:80-81: d = silu(edge_input(d))
        d = self.edge_input(d, post_act='silu')
        for i, mp_layer in enumerate(self.mp_layers):
            h, d = mp_layer.step(h, d, idx, need_edge_update=i + 1 < len(self.mp_layers))
        if self.node_wise_output_layers > 0:
            h = self.node_wise_output_network(h)
        g.ndata['feat'] = h
        g.edata['d'] = ops.gather_rows(d.detach().contiguous(), idx.inv_perm)   # side effect, edge-id order
        readout = tape.apply(ReadoutFn, h, idx, self._readout_codes)
        return self.output(readout)


class Net3DLayer(nn.Module):
    """reference models/net3d.py:84-125."""

    def __init__(self, edge_dim, reduce_func, hidden_dim, batch_norm, batch_norm_momentum, dropout, mid_activation,
                 message_net_layers, update_net_layers):
        super().__init__()
        self.message_network = MLP(in_dim=hidden_dim * 2 + edge_dim, hidden_size=hidden_dim, out_dim=hidden_dim,
                                   mid_batch_norm=batch_norm, last_batch_norm=batch_norm,
                                   batch_norm_momentum=batch_norm_momentum, layers=message_net_layers,
                                   mid_activation=mid_activation, dropout=dropout, last_activation=mid_activation)
        if reduce_func not in ('sum', 'mean'):
            raise ValueError('reduce function not supported: ', reduce_func)
        self.reduce_mean = reduce_func == 'mean'
        self.update_network = MLP(in_dim=hidden_dim, hidden_size=hidden_dim, out_dim=hidden_dim,
                                  mid_batch_norm=batch_norm, last_batch_norm=batch_norm,
                                  batch_norm_momentum=batch_norm_momentum, layers=update_net_layers,
                                  mid_activation=mid_activation, dropout=dropout, last_activation='None')
        self.soft_edge_network = nn.Linear(hidden_dim, 1)
        act_name(mid_activation)

    def step(self, h, d, idx, need_edge_update=True):
        m = self.message_network.forward_edge(h, d, idx)                        # :113-115
        d_new = tape.apply(_AddFn, d, m) if need_edge_update else d                   # :116 (dead for the last layer)
        msg = tape.apply(SoftEdgeFn, m, self.soft_edge_network.weight, self.soft_edge_network.bias)   # :117-118
        m_sum = tape.apply(SegmentReduceFn, msg, idx, self.reduce_mean)               # :109 fn.mean / fn.sum
        h_new = self.update_network(tape.apply(_AddFn, m_sum, h), residual=h)         # :120-125
        return h_new, d_new

    def forward(self, graph):
        g = as_batched_graph(graph)
        idx = g.index()
        d = ops.gather_rows(g.edata['d'].contiguous(), idx.perm)
        h, d = self.step(g.ndata['feat'], d, idx)
        g.ndata['feat'] = h
        g.edata['d'] = ops.gather_rows(d.detach().contiguous(), idx.inv_perm)
