"""Minimal stand-in for the subset of DGL the 3DInfomax hot path touches.

CONTAINER-ONLY TOOLING.  `dgl` is a third-party dependency of the reference
(environment.yml:12, unpinned) that is not installed here.  This file restates the
*documented* DGL semantics that models/pna.py, models/net3d.py
and datasets/custom_collate.py rely on, so that the unmodified reference modules can be
imported by tests/golden/gen_golden.py to produce golden vectors.  It is never imported
by the product path, by bench.py, or on the GPU box.

Semantics restated:
  graph((src,dst), num_nodes)   edge ids = positions in the given lists
  batch(graphs)                 block-diagonal concat, node ids offset, batch_num_nodes kept
                                (flattened for already-batched elements)
  apply_edges(udf)              udf sees all E edges in edge-id order
  update_all(msg, reduce[,apply]) UDF reduce = degree bucketing: one call per distinct
                                in-degree D>0 with mailbox [n_D, D, F], messages ordered by
                                edge id, zero rows for isolated nodes; builtin fn.mean/fn.sum
  readout_nodes(g, key, op)     per-graph segment reduce in batch order
"""
import sys
import types

import torch

from . import function  # noqa: F401


class _Frame(dict):
    pass


class _EdgeBatch:
    def __init__(self, g):
        self.src = {k: v[g._src] for k, v in g.ndata.items()}
        self.dst = {k: v[g._dst] for k, v in g.ndata.items()}
        self.data = g.edata


class _NodeBatch:
    def __init__(self, data, mailbox=None):
        self.data = data
        self.mailbox = mailbox


class DGLGraph:
    def __init__(self, src, dst, num_nodes, batch_num_nodes=None):
        self._src = torch.as_tensor(src, dtype=torch.long)
        self._dst = torch.as_tensor(dst, dtype=torch.long)
        self._n = int(num_nodes)
        self.ndata = _Frame()
        self.edata = _Frame()
        self._bnn = (torch.tensor([self._n], dtype=torch.long)
                     if batch_num_nodes is None else batch_num_nodes)

    # --- structure queries -------------------------------------------------
    def number_of_nodes(self):
        return self._n

    num_nodes = number_of_nodes

    def number_of_edges(self):
        return int(self._src.numel())

    num_edges = number_of_edges

    def edges(self):
        return self._src, self._dst

    def batch_num_nodes(self):
        return self._bnn

    def in_degrees(self):
        return torch.bincount(self._dst, minlength=self._n)

    def to(self, device):
        g = DGLGraph(self._src.to(device), self._dst.to(device), self._n, self._bnn.to(device))
        g.ndata.update({k: v.to(device) for k, v in self.ndata.items()})
        g.edata.update({k: v.to(device) for k, v in self.edata.items()})
        return g

    @property
    def device(self):
        return self._src.device

    # --- message passing ----------------------------------------------------
    def apply_edges(self, udf):
        self.edata.update(udf(_EdgeBatch(self)))

    def apply_nodes(self, udf):
        self.ndata.update(udf(_NodeBatch(self.ndata)))

    def update_all(self, message_func, reduce_func, apply_node_func=None):
        msgs = message_func(_EdgeBatch(self))
        if isinstance(reduce_func, function._Builtin):
            m = msgs[reduce_func.msg]
            out = torch.zeros((self._n,) + tuple(m.shape[1:]), dtype=m.dtype, device=m.device)
            out = out.index_add(0, self._dst, m)
            if reduce_func.op == 'mean':
                deg = self.in_degrees().clamp(min=1).to(m.dtype)
                out = out / deg.view(-1, *([1] * (m.dim() - 1)))
            self.ndata[reduce_func.out] = out
        else:
            deg = self.in_degrees()
            # edges grouped by destination, ordered by edge id inside each group
            order = torch.sort(self._dst, stable=True)[1]
            rowptr = torch.zeros(self._n + 1, dtype=torch.long)
            rowptr[1:] = torch.cumsum(deg, 0)
            results = {}
            for D in sorted(set(deg.tolist())):
                if D == 0:
                    continue
                nodes = torch.nonzero(deg == D).flatten()
                eids = order[(rowptr[nodes][:, None] + torch.arange(D)[None, :])]  # [n_D, D]
                mailbox = {k: v[eids] for k, v in msgs.items()}
                data = {k: v[nodes] for k, v in self.ndata.items()}
                out = reduce_func(_NodeBatch(data, mailbox))
                for k, v in out.items():
                    if k not in results:
                        results[k] = torch.zeros((self._n,) + tuple(v.shape[1:]), dtype=v.dtype,
                                                 device=v.device)
                    results[k] = results[k].index_copy(0, nodes, v)
            self.ndata.update(results)
        if apply_node_func is not None:
            self.ndata.update(apply_node_func(_NodeBatch(self.ndata)))


def graph(data, num_nodes=None, device=None):
    src, dst = data
    src = torch.as_tensor(src, dtype=torch.long)
    dst = torch.as_tensor(dst, dtype=torch.long)
    if num_nodes is None:
        num_nodes = int(max(src.max().item(), dst.max().item())) + 1 if src.numel() else 0
    return DGLGraph(src, dst, num_nodes)


def batch(graphs):
    srcs, dsts, bnn = [], [], []
    off = 0
    for g in graphs:
        srcs.append(g._src + off)
        dsts.append(g._dst + off)
        bnn.append(g._bnn)
        off += g._n
    out = DGLGraph(torch.cat(srcs), torch.cat(dsts), off, torch.cat(bnn))
    for k in graphs[0].ndata:
        out.ndata[k] = torch.cat([g.ndata[k] for g in graphs], 0)
    for k in graphs[0].edata:
        out.edata[k] = torch.cat([g.edata[k] for g in graphs], 0)
    return out


def readout_nodes(g, feat, weight=None, op='sum', ntype=None):
    x = g.ndata[feat]
    outs = []
    start = 0
    for n in g.batch_num_nodes().tolist():
        seg = x[start:start + n]
        start += n
        if op == 'sum':
            outs.append(seg.sum(0))
        elif op == 'mean':
            outs.append(seg.mean(0))
        elif op == 'max':
            outs.append(seg.max(0)[0])
        elif op == 'min':
            outs.append(seg.min(0)[0])
        else:
            raise ValueError(op)
    return torch.stack(outs, 0)


random = types.ModuleType('dgl.random')
random.seed = lambda s: None
sys.modules['dgl.random'] = random
