"""Batched molecular graph container.

Replaces the slice of DGL the hot path uses (`dgl.graph`, `dgl.batch`, `ndata`/`edata`,
`batch_num_nodes`, `number_of_nodes`, `edges`, `.to`) - see reference
datasets/custom_collate.py:105-114 (contrastive_collate -> dgl.batch) and
trainer/self_supervised_trainer.py:24-29 (what the trainer touches).

On top of the DGL-like surface it carries the index the HIP kernels are driven by
(`GraphIndex`): a CSR **by destination**.  All edge-sized tensors inside the models live
in *destination-sorted* order ("epos" order, stable w.r.t. edge id, so the per-node
message order equals DGL's mailbox order), which makes every neighbourhood a contiguous
row range `[in_ptr[v], in_ptr[v+1])` - the layout the one-pass segmented aggregation
kernel wants - and needs no atomics anywhere (out-edge sums go through `out_ptr/out_epos`).
"""
from dataclasses import dataclass
from typing import Dict, List, Optional

import numpy as np
import torch


@dataclass
class GraphIndex:
    """int32 index arrays, all on one device.  E edges, N nodes, B graphs."""
    num_nodes: int
    num_edges: int
    num_graphs: int
    in_ptr: torch.Tensor      # [N+1] CSR row pointer by destination
    perm: torch.Tensor        # [E]   epos -> edge id (stable sort of dst)
    src_s: torch.Tensor       # [E]   source node of edge at epos
    dst_s: torch.Tensor       # [E]   destination node of edge at epos (non-decreasing)
    out_ptr: torch.Tensor     # [N+1] CSR row pointer by source
    out_epos: torch.Tensor    # [E]   epos of the out-edges of each node, grouped by source
    graph_ptr: torch.Tensor   # [B+1] node offsets of the graphs in the batch
    inv_perm: torch.Tensor    # [E]   edge id -> epos
    max_in_degree: int
    # nodes grouped by in-degree (degree-combined posttrans weights, csrc/grouped.hip); built lazily by degree_groups()
    deg_rows: Optional[torch.Tensor] = None        # [M_pad] node ids grouped by in-degree, each group padded to 64 with -1
    deg_tile_group: Optional[torch.Tensor] = None  # [M_pad/64] group of every 64-row tile
    deg_groups: Optional[tuple] = None             # host: ((D, start, count), ...) for the in-degrees present (0 included)

    def to(self, device):
        out = GraphIndex(self.num_nodes, self.num_edges, self.num_graphs,
                         *[t.to(device, non_blocking=True) for t in
                           (self.in_ptr, self.perm, self.src_s, self.dst_s, self.out_ptr,
                            self.out_epos, self.graph_ptr, self.inv_perm)], self.max_in_degree)
        if self.deg_rows is not None:
            out.deg_rows = self.deg_rows.to(device, non_blocking=True)
            out.deg_tile_group = self.deg_tile_group.to(device, non_blocking=True)
            out.deg_groups = self.deg_groups
        return out

    def degree_groups(self):
        """(deg_rows, deg_tile_group, deg_groups), computed once per batch from in_ptr (host round trip only if the
        batch was not assembled through build_index / dataset.assemble, which fill them in on the host)."""
        if self.deg_rows is None:
            indeg = np.diff(self.in_ptr.cpu().numpy().astype(np.int64))
            rows, tiles, groups = group_nodes_by_degree(indeg, include_zero=True)
            dev = self.in_ptr.device
            self.deg_rows = torch.from_numpy(rows).to(dev)
            self.deg_tile_group = torch.from_numpy(tiles).to(dev)
            self.deg_groups = groups
        return self.deg_rows, self.deg_tile_group, self.deg_groups


def group_nodes_by_degree(indeg, pad=64, include_zero=False):
    """Node ids grouped by in-degree (ascending D, ascending node id), every group padded with -1 to a multiple of
    `pad` rows; tile -> group map; ((D, start, count), ...).  `include_zero`: nodes without in-edges form a group of their
    own (D = 0, scaler coefficients 0: their aggregate is a zero row) so that the groups cover EVERY node - what the batch
    index carries, because the fused posttrans GEMM takes its BatchNorm statistics from the grouped launch's epilogue."""
    indeg = np.asarray(indeg, dtype=np.int64)
    order = np.argsort(indeg, kind='stable')
    if not include_zero:
        order = order[indeg[order] > 0]
    degs, counts = np.unique(indeg[order], return_counts=True)
    padded = (counts + pad - 1) // pad * pad
    starts = np.cumsum(padded) - padded
    rows = np.full(int(padded.sum()), -1, dtype=np.int32)
    src_off = np.cumsum(counts) - counts
    dst_idx = np.repeat(starts - src_off, counts) + np.arange(order.shape[0])
    rows[dst_idx] = order.astype(np.int32)
    tiles = np.repeat(np.arange(degs.shape[0], dtype=np.int32), padded // pad)
    groups = tuple((int(d), int(s), int(c)) for d, s, c in zip(degs, starts, counts))
    return rows, tiles, groups


def build_index(src, dst, num_nodes, batch_num_nodes) -> GraphIndex:
    """Host-side (numpy) construction of the kernel index from an edge list."""
    src = np.asarray(src, dtype=np.int64)
    dst = np.asarray(dst, dtype=np.int64)
    bnn = np.asarray(batch_num_nodes, dtype=np.int64)
    E = src.shape[0]
    perm = np.argsort(dst, kind='stable')
    src_s, dst_s = src[perm], dst[perm]
    indeg = np.bincount(dst, minlength=num_nodes)
    in_ptr = np.zeros(num_nodes + 1, dtype=np.int64)
    np.cumsum(indeg, out=in_ptr[1:])
    out_epos = np.argsort(src_s, kind='stable')
    out_ptr = np.zeros(num_nodes + 1, dtype=np.int64)
    np.cumsum(np.bincount(src, minlength=num_nodes), out=out_ptr[1:])
    graph_ptr = np.zeros(bnn.shape[0] + 1, dtype=np.int64)
    np.cumsum(bnn, out=graph_ptr[1:])
    inv_perm = np.empty(E, dtype=np.int64)
    inv_perm[perm] = np.arange(E)
    i32 = lambda a: torch.from_numpy(np.ascontiguousarray(a.astype(np.int32)))
    rows, tiles, groups = group_nodes_by_degree(indeg, include_zero=True)
    return GraphIndex(int(num_nodes), int(E), int(bnn.shape[0]), i32(in_ptr), i32(perm), i32(src_s),
                      i32(dst_s), i32(out_ptr), i32(out_epos), i32(graph_ptr), i32(inv_perm),
                      int(indeg.max()) if E else 0, torch.from_numpy(rows), torch.from_numpy(tiles), groups)


class BatchedMolGraph:
    """A (batch of) graph(s) with DGL's node/edge-frame surface."""

    def __init__(self, src, dst, num_nodes: int, batch_num_nodes=None,
                 ndata: Optional[Dict[str, torch.Tensor]] = None,
                 edata: Optional[Dict[str, torch.Tensor]] = None,
                 index: Optional[GraphIndex] = None):
        self._src = torch.as_tensor(src, dtype=torch.long)
        self._dst = torch.as_tensor(dst, dtype=torch.long)
        self._n = int(num_nodes)
        if batch_num_nodes is None:
            batch_num_nodes = torch.tensor([self._n], dtype=torch.long)
        self._bnn = torch.as_tensor(batch_num_nodes, dtype=torch.long)
        self.ready_event = None
        self.ndata: Dict[str, torch.Tensor] = dict(ndata or {})
        self.edata: Dict[str, torch.Tensor] = dict(edata or {})
        self._index = index

    # ---- DGL-like surface --------------------------------------------------------------
    def number_of_nodes(self):
        return self._n

    num_nodes = number_of_nodes

    def number_of_edges(self):
        return int(self._src.shape[0])

    num_edges = number_of_edges

    def edges(self):
        return self._src, self._dst

    def batch_num_nodes(self):
        return self._bnn

    @property
    def batch_size(self):
        return int(self._bnn.shape[0])

    @property
    def device(self):
        return self._src.device

    def to(self, device):
        device = torch.device(device)
        g = BatchedMolGraph(self._src.to(device), self._dst.to(device), self._n, self._bnn.to(device),
                            {k: v.to(device) for k, v in self.ndata.items()},
                            {k: v.to(device) for k, v in self.edata.items()},
                            self.index().to(device))
        if device.type == 'cuda':
            g.mark_ready()
        return g

    def mark_ready(self):
        """Record that everything this batch consists of has been enqueued on the current stream (streams.py: lets an
        independent consumer on another stream wait for the batch only)."""
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self._src.device))
        self.ready_event = ev

    def local_copy(self):
        """Same structure and tensors, fresh frames: the forward pass overwrites ndata/edata['feat'] (reference
        models/pna.py:162-163), so a resident batch that is stepped on repeatedly is forwarded through a copy."""
        g = BatchedMolGraph(self._src, self._dst, self._n, self._bnn, dict(self.ndata), dict(self.edata),
                            self._index)
        g.ready_event = self.ready_event
        return g

    # ---- kernel index -----------------------------------------------------------------
    def index(self) -> GraphIndex:
        if self._index is None:
            idx = build_index(self._src.cpu().numpy(), self._dst.cpu().numpy(), self._n,
                              self._bnn.cpu().numpy())
            self._index = idx.to(self._src.device)
        return self._index


def batch(graphs: List[BatchedMolGraph]) -> BatchedMolGraph:
    """Block-diagonal batching (DGL `dgl.batch` semantics): nodes and
    edges concatenated in list order, node ids offset, frames concatenated,
    batch_num_nodes flattened."""
    srcs, dsts, bnn = [], [], []
    off = 0
    # items may be DGL-like graphs (the reference's own datasets yield `dgl.DGLGraph`s and `from infomax3d_amd import *`
    # rebinds the collate functions that land here): duck-typed through as_batched_graph, frames shared
    graphs = [as_batched_graph(g) for g in graphs]
    for g in graphs:
        srcs.append(g._src + off)
        dsts.append(g._dst + off)
        bnn.append(g._bnn)
        off += g._n
    out = BatchedMolGraph(torch.cat(srcs), torch.cat(dsts), off, torch.cat(bnn))
    for k in graphs[0].ndata:
        out.ndata[k] = torch.cat([g.ndata[k] for g in graphs], 0)
    for k in graphs[0].edata:
        out.edata[k] = torch.cat([g.edata[k] for g in graphs], 0)
    return out


def bond_graph(mol) -> BatchedMolGraph:
    """2D bond graph of one `synth.Molecule` (reference datasets/qm9_dataset.py:221-231)."""
    return BatchedMolGraph(torch.from_numpy(mol.src), torch.from_numpy(mol.dst), mol.n_atoms,
                           ndata={'feat': torch.from_numpy(mol.atom_feat)},
                           edata={'feat': torch.from_numpy(mol.bond_feat)})


def complete_graph(mol, coords=None) -> BatchedMolGraph:
    """Complete 3D distance graph of one molecule (reference datasets/qm9_dataset.py:233-244)."""
    from .synth import complete_graph_edges, pairwise_distances
    coords = mol.coords if coords is None else coords
    src, dst = complete_graph_edges(mol.n_atoms)
    d = pairwise_distances(coords, src, dst)
    return BatchedMolGraph(torch.from_numpy(src), torch.from_numpy(dst), mol.n_atoms,
                           ndata={'feat': torch.from_numpy(mol.atom_feat)},
                           edata={'d': torch.from_numpy(d)})


def contrastive_collate(batch_items):
    """Mirror of reference datasets/custom_collate.py:105-114 (items may be BatchedMolGraphs or DGL-like graphs)."""
    graphs, graphs3d, *targets = map(list, zip(*batch_items))
    if targets:
        return [batch(graphs)], [batch(graphs3d)], torch.stack(*targets).float()
    return [batch(graphs)], [batch(graphs3d)]


def conformer_collate(batch_items):
    """Mirror of reference datasets/custom_collate.py:155-157."""
    graphs, confs = map(list, zip(*batch_items))
    return [batch(graphs)], [batch(confs)]


def graph_collate(batch_items):
    """Mirror of reference datasets/custom_collate.py:12-18 (the collate of the fine-tuning configs, e.g.
    configs_clean/tune_QM9_homo.yml): ([batched graph], targets [B, T]); 1-d targets get a trailing axis."""
    graphs, targets = map(list, zip(*batch_items))
    targets = torch.stack(targets).float()
    if len(targets.shape) == 1:
        targets = targets.unsqueeze(-1)
    return [batch(graphs)], targets


def _snorm_n(graphs):
    """sqrt(1 / n_atoms) per node, [N, 1] (reference datasets/custom_collate.py:45-47, 96-98: the graph-size
    normalisation factor of the original PNA, models/pna_original.py:258-259)."""
    sizes = [as_batched_graph(g).number_of_nodes() for g in graphs]
    return torch.cat([torch.full((n, 1), 1.0 / float(n), dtype=torch.float32) for n in sizes]).sqrt()


def s_norm_graph_collate(batch_items):
    """Mirror of reference datasets/custom_collate.py:43-49: ([batched graph, snorm_n], targets)."""
    graphs, targets = map(list, zip(*batch_items))
    return [batch(graphs), _snorm_n(graphs)], torch.stack(targets).float()


def s_norm_contrastive_collate(batch_items):
    """Mirror of reference datasets/custom_collate.py:93-102: ([batched graph, snorm_n], [batched 3D graph])."""
    graphs, graphs3d = map(list, zip(*batch_items))
    return [batch(graphs), _snorm_n(graphs)], [batch(graphs3d)]


def as_batched_graph(g) -> BatchedMolGraph:
    """Accept a BatchedMolGraph, or any DGL-like object exposing edges(), number_of_nodes(),
    batch_num_nodes(), ndata, edata (real `dgl.DGLGraph` included).  The frames are shared,
    so the forward pass's side effects (reference models/pna.py:162-163,213) land on the
    caller's object."""
    if isinstance(g, BatchedMolGraph):
        return g
    cached = getattr(g, '_amd_batched', None)
    if cached is not None:
        return cached
    src, dst = g.edges()
    bg = BatchedMolGraph(src, dst, g.number_of_nodes(), g.batch_num_nodes())
    bg.ndata = g.ndata
    bg.edata = g.edata
    try:
        g._amd_batched = bg
    except Exception:
        pass
    return bg
