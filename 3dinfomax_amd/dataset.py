"""Flat-tensor molecule store + vectorised batch assembly.

The reference keeps a dataset as flat tensors with per-molecule slices (`features_tensor`, `e_features_tensor`,
`edge_indices`, `coordinates`, `atom_slices`, `edge_slices`: reference datasets/qm9_dataset.py:153, 189-208) but then
builds 2 x B Python graph objects per step and `dgl.batch`es them in the training thread
(datasets/qm9_dataset.py:221-244, datasets/custom_collate.py:105-114) - on a GPU that is what bounds the step.

Here a batch is assembled WITHOUT a per-molecule Python loop:
  * 2D bond graph: numpy gathers over the flat arrays; because batching is block-diagonal, the destination-sorted
    kernel index of the batch is the concatenation of per-molecule indices that are precomputed ONCE
    (`perm`, in/out degrees, `out_epos`), shifted by the node/edge offsets;
  * 3D complete graph: only coordinates and node offsets go to the device, the edges, distances and the index are
    built there by one kernel (csrc/batch.hip).
The result is identical to `graph.batch([bond_graph(m) ...])` / `graph.batch([complete_graph(m) ...])`
(tests/test_dataset.py).
"""
from typing import List, Sequence

import numpy as np
import torch

from .graph import BatchedMolGraph, GraphIndex, group_nodes_by_degree


def _ranges(starts, lengths):
    """Concatenation of arange(s, s+l) for every (s, l) - vectorised."""
    total = int(lengths.sum())
    if total == 0:
        return np.zeros(0, dtype=np.int64)
    offs = np.cumsum(lengths) - lengths
    return np.repeat(starts - offs, lengths) + np.arange(total, dtype=np.int64)


class FlatMolDataset:
    def __init__(self, mols: Sequence):
        n = np.array([m.n_atoms for m in mols], dtype=np.int64)
        e = np.array([m.src.shape[0] for m in mols], dtype=np.int64)
        self.n_atoms, self.n_edges = n, e
        self.atom_start = np.cumsum(n) - n
        self.edge_start = np.cumsum(e) - e
        self.atom_feat = np.concatenate([m.atom_feat for m in mols]).astype(np.int64)
        self.bond_feat = np.concatenate([m.bond_feat for m in mols]).astype(np.int64)
        self.coords = np.concatenate([m.coords for m in mols]).astype(np.float32)
        self.src = np.concatenate([m.src for m in mols]).astype(np.int64)          # local node ids, edge-id order
        self.dst = np.concatenate([m.dst for m in mols]).astype(np.int64)
        # per-molecule kernel index, computed once (block-diagonal batching concatenates them)
        perm, inv, out_epos, indeg, outdeg = [], [], [], [], []
        for m in mols:
            p = np.argsort(m.dst, kind='stable')
            ip = np.empty_like(p)
            ip[p] = np.arange(p.shape[0])
            perm.append(p)
            inv.append(ip)
            out_epos.append(np.argsort(m.src[p], kind='stable'))
            indeg.append(np.bincount(m.dst, minlength=m.n_atoms))
            outdeg.append(np.bincount(m.src, minlength=m.n_atoms))
        self.perm = np.concatenate(perm).astype(np.int64)            # local epos -> local edge id
        self.inv_perm = np.concatenate(inv).astype(np.int64)
        self.out_epos = np.concatenate(out_epos).astype(np.int64)
        self.indeg = np.concatenate(indeg).astype(np.int64)
        self.outdeg = np.concatenate(outdeg).astype(np.int64)

    def __len__(self):
        return self.n_atoms.shape[0]

    # ------------------------------------------------------------------------------------------------------
    def assemble(self, ids, device, pin=False):
        """-> ([g2d], [g3d]) on `device`, the layout `contrastive_collate` returns (reference
        datasets/custom_collate.py:105-114)."""
        g2, xyz, graph_ptr_dev, n, bnn = self.assemble_2d(ids, device, pin)
        return [g2], [complete_graphs_on_device(xyz, graph_ptr_dev, n, bnn, g2._edge_ptr3)]

    def assemble_2d(self, ids, device, pin=False):
        """The bond-graph half (pure numpy + three H2D copies; also usable on the CPU for tests)."""
        return host_batch_to_device(self.assemble_host(ids), device, pin)

    def assemble_host(self, ids):
        """The host half of a batch: a dict of CPU tensors and plain ints, picklable - what a DataLoader worker process
        hands to the training process (BatchStream)."""
        ids = np.asarray(ids, dtype=np.int64)
        n, e = self.n_atoms[ids], self.n_edges[ids]
        B, N, E = ids.shape[0], int(n.sum()), int(e.sum())
        node_off = np.cumsum(n) - n
        edge_off = np.cumsum(e) - e
        ngi = _ranges(self.atom_start[ids], n)            # rows of the flat node arrays
        egi = _ranges(self.edge_start[ids], e)            # rows of the flat edge arrays
        e_node_off = np.repeat(node_off, e)
        e_edge_off = np.repeat(edge_off, e)
        src = self.src[egi] + e_node_off                  # edge-id order
        dst = self.dst[egi] + e_node_off
        perm = self.perm[egi] + e_edge_off
        i32 = np.empty(2 * (N + 1) + 5 * E + (B + 1), dtype=np.int32)
        o = 0
        cuts = []

        def put(a):
            nonlocal o
            i32[o:o + a.shape[0]] = a
            cuts.append((o, o + a.shape[0]))
            o += a.shape[0]
        in_ptr = np.zeros(N + 1, dtype=np.int64)
        np.cumsum(self.indeg[ngi], out=in_ptr[1:])
        out_ptr = np.zeros(N + 1, dtype=np.int64)
        np.cumsum(self.outdeg[ngi], out=out_ptr[1:])
        graph_ptr = np.zeros(B + 1, dtype=np.int64)
        np.cumsum(n, out=graph_ptr[1:])
        for a in (in_ptr, perm, src[perm], dst[perm], out_ptr, self.out_epos[egi] + e_edge_off, graph_ptr,
                  self.inv_perm[egi] + e_edge_off):
            put(a)                                         # s_in, s_perm, s_src, s_dst, s_out, s_oe, s_gp, s_inv
        i64 = np.concatenate([src, dst, self.atom_feat[ngi].ravel(), self.bond_feat[egi].ravel()])
        indeg = self.indeg[ngi]
        rows, tiles, groups = group_nodes_by_degree(indeg, include_zero=True)
        edge_ptr = np.zeros(B + 1, dtype=np.int32)          # complete-graph edge offsets, n (n - 1) per molecule
        np.cumsum(n * (n - 1), out=edge_ptr[1:])
        # ONE packed byte buffer (16-byte aligned segments): the whole batch goes to the device with a single copy
        segs = (('i64', i64), ('i32', i32), ('f32', np.ascontiguousarray(self.coords[ngi])), ('rows', rows), ('tiles', tiles),
                ('edge_ptr', edge_ptr))
        layout, o = [], 0
        for name, a in segs:
            layout.append((name, o, int(a.size)))
            o += (a.nbytes + 15) & ~15
        buf = np.zeros(max(o, 16), dtype=np.uint8)           # (zeros: the alignment padding is part of the picklable batch)
        for (name, off, cnt), (_, a) in zip(layout, segs):
            buf[off:off + a.nbytes] = a.reshape(-1).view(np.uint8)
        return {'buf': torch.from_numpy(buf), 'layout': tuple(layout), 'n': torch.from_numpy(n), 'groups': groups, 'cuts': cuts,
                'dims': (B, N, E), 'max_indeg': int(indeg.max()) if E else 0}


_SEG_DTYPE = {'i64': torch.int64, 'i32': torch.int32, 'f32': torch.float32, 'rows': torch.int32, 'tiles': torch.int32,
              'edge_ptr': torch.int32}


class _Stager:
    """Host-to-device copies of whole batches off the compute stream: a few reusable pinned staging buffers and a copy
    stream per device.  A pageable `tensor.to(device, non_blocking=True)` on the compute stream is synchronous AND
    stream-ordered - the host then waits until the GPU has finished everything enqueued before it, i.e. it loses its whole
    run-ahead once per batch (measured: 178 k molecules/s with six such copies per batch against 206 k resident)."""
    SLOTS = 3

    def __init__(self, device):
        self.stream = torch.cuda.Stream(device=device)
        self.slots = [[None, None] for _ in range(self.SLOTS)]       # [pinned uint8 tensor, event of its last copy]
        self.k = 0

    def to_device(self, buf, device):
        n = buf.numel()
        if buf.is_pinned():
            src = buf
            slot = None
        else:
            slot = self.slots[self.k]
            self.k = (self.k + 1) % self.SLOTS
            if slot[1] is not None:
                slot[1].synchronize()                                # the copy that last read this slot has finished
            if slot[0] is None or slot[0].numel() < n:
                slot[0] = torch.empty(max(n, 1 << 20) * 5 // 4, dtype=torch.uint8).pin_memory()
            # plain memcpy: torch's CPU copy_ goes through the intra-op thread pool, which stalls for ~90 ms every few
            # calls next to autograd's worker thread on a 128-core host (tools/dbg_asm.py)
            np.copyto(slot[0].numpy()[:n], buf.numpy())
            src = slot[0][:n]
        main = torch.cuda.current_stream(device)
        with torch.cuda.stream(self.stream):
            d = torch.empty(n, dtype=torch.uint8, device=device)
            d.copy_(src, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(self.stream)
        if slot is not None:
            slot[1] = ev
        main.wait_event(ev)
        d.record_stream(main)
        return d


_stagers = {}


def _stager(device):
    device = torch.device(device)
    key = device.index if device.index is not None else torch.cuda.current_device()
    st = _stagers.get(key)
    if st is None:
        st = _stagers[key] = _Stager(device)
    return st


def host_batch_to_device(hb, device, pin=False):
    """ONE H2D copy of a host batch (FlatMolDataset.assemble_host), on a copy stream through pinned staging, and the bond
    graph on `device` -> (g2d, xyz, graph_ptr on device, atoms per molecule (numpy), the same as a tensor)."""
    B, N, E = (int(v) for v in hb['dims'])      # (a DataLoader's default conversion turns tuples into lists)
    device = torch.device(device)
    buf = hb['buf']
    d = _stager(device).to_device(buf, device) if device.type == 'cuda' else buf
    seg = {}
    for name, off, cnt in hb['layout']:
        dt = _SEG_DTYPE[name]
        seg[name] = d[int(off):int(off) + int(cnt) * dt.itemsize].view(dt)
    d32, d64, xyz = seg['i32'], seg['i64'], seg['f32'].view(-1, 3)
    bnn = hb['n']
    s_in, s_perm, s_src, s_dst, s_out, s_oe, s_gp, s_inv = (slice(int(a), int(b)) for a, b in hb['cuts'])
    idx2 = GraphIndex(N, E, B, d32[s_in], d32[s_perm], d32[s_src], d32[s_dst], d32[s_out], d32[s_oe], d32[s_gp],
                      d32[s_inv], int(hb['max_indeg']), seg['rows'], seg['tiles'],
                      tuple(tuple(int(v) for v in gr) for gr in hb['groups']))
    g2 = BatchedMolGraph(d64[:E], d64[E:2 * E], N, bnn,
                         ndata={'feat': d64[2 * E:2 * E + 9 * N].view(N, 9)},
                         edata={'feat': d64[2 * E + 9 * N:].view(E, 3)}, index=idx2)
    if device.type == 'cuda':
        g2.mark_ready()
    g2._edge_ptr3 = seg['edge_ptr']              # complete-graph edge offsets, already on the device
    return g2, xyz, d32[s_gp], bnn.numpy(), bnn


class BatchStream(torch.utils.data.Dataset):
    """Shuffled batches of a FlatMolDataset as host batches, for `torch.utils.data.DataLoader(stream, batch_size=None,
    num_workers=k, pin_memory=True)`: the numpy assembly runs in the worker processes - where the reference runs its
    per-molecule graph construction (DataLoader workers, train.py:589-600) - and the training process only issues the
    H2D copies and the device-side complete-graph build (`to_device`).  Item i is batch i of a fixed, seeded sequence."""

    def __init__(self, flat: 'FlatMolDataset', batch_size: int, steps: int, seed: int = 0, drop_last: bool = True):
        self.flat, self.batch_size, self.steps, self.seed = flat, batch_size, steps, seed
        self.per_epoch = max(len(flat) // batch_size, 1) if drop_last else -(-len(flat) // batch_size)

    def __len__(self):
        return self.steps

    def __getitem__(self, i):
        epoch, k = divmod(i, self.per_epoch)
        order = np.random.default_rng(self.seed + epoch).permutation(len(self.flat))
        return self.flat.assemble_host(order[k * self.batch_size:(k + 1) * self.batch_size])

    @staticmethod
    def to_device(hb, device):
        """-> ([g2d], [g3d]) on `device` (the tensors of `hb` are pinned when the loader was built with pin_memory=True)"""
        g2, xyz, graph_ptr_dev, n, bnn = host_batch_to_device(hb, device)
        return [g2], [complete_graphs_on_device(xyz, graph_ptr_dev, n, bnn, g2._edge_ptr3)]


class DevicePrefetcher:
    """Iterates `loader` (host batches, e.g. a DataLoader over a BatchStream) with the device half of the batch assembly - the H2D copy,
    the index views, the complete-graph build: `to_device`, ~0.5 ms of Python per batch - done by a helper thread `depth` batches ahead
    of the training loop instead of inside it.  The reference does this part in its training loop too (`move_to_device` +
    per-batch graph handling, trainer/trainer.py:111-115); here the loop is ~1.4 ms of host time per step against 2.0 ms on the
    device, so on a slower host another 0.5 ms per batch makes the HOST the bound (bench.py: 179-200 k against 252 k molecules/s; with
    a 200 us switch interval the helper brings that to 215-219 k).  On a box whose host keeps up anyway the two Python threads only
    contend for the interpreter lock (246 -> 199 k measured): use it when the loop is host-bound.  The helper runs its
    Python while the training thread is inside the library's C calls (they release the interpreter lock: ~0.9 ms per step).
    Everything is enqueued on the stream that was current where the prefetcher was built, in the order the batches are handed
    out, so a batch's build kernels are always in front of its first use; `switch_interval` (seconds, optional) bounds how long
    the training thread can wait for the interpreter lock when it returns from a C call (sys.setswitchinterval: process-wide,
    restored by close())."""

    _END = object()

    def __init__(self, loader, device, depth=2, to_device=None, switch_interval=None):
        import queue
        import sys
        import threading
        self.device = torch.device(device)
        self.to_device = to_device or BatchStream.to_device
        self.q = queue.Queue(maxsize=max(int(depth), 1))
        self.stream = torch.cuda.current_stream(self.device) if self.device.type == 'cuda' else None
        self._old_interval = None
        if switch_interval is not None:
            self._old_interval = sys.getswitchinterval()
            sys.setswitchinterval(float(switch_interval))
        self._stop = False
        self._done = None
        self.thread = threading.Thread(target=self._run, args=(iter(loader),), name='i3d-device-prefetch', daemon=True)
        self.thread.start()

    def _run(self, it):
        try:
            if self.stream is not None:
                torch.cuda.set_device(self.device)
            while not self._stop:
                try:
                    hb = next(it)
                except StopIteration:
                    break
                if self.stream is not None:
                    with torch.cuda.stream(self.stream):
                        item = self.to_device(hb, self.device)
                else:
                    item = self.to_device(hb, self.device)
                self.q.put(item)
            self.q.put(self._END)
        except BaseException as e:          # handed to the consumer
            self.q.put(e)

    def __iter__(self):
        return self

    def __next__(self):
        if self._done is not None:       # the end (or the helper's exception) is sticky: a second next() must not wait for ever
            raise self._done
        item = self.q.get()
        if item is self._END:
            self._done = StopIteration()
            raise StopIteration
        if isinstance(item, BaseException):
            self._done = item
            raise item
        return item

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    def __del__(self):          # a consumer that stopped early without close(): release the helper, restore the switch interval
        try:
            if self.thread.is_alive() or self._old_interval is not None:
                self.close()
        except Exception:       # noqa: BLE001 - interpreter shutdown
            pass

    def close(self):
        """stops the helper (batches it still holds are dropped) and restores the switch interval"""
        import queue
        import sys
        self._stop = True
        while self.thread.is_alive():
            try:
                self.q.get(timeout=0.05)
            except queue.Empty:
                pass
        self.thread.join()
        if self._old_interval is not None:
            sys.setswitchinterval(self._old_interval)
            self._old_interval = None


def complete_graphs_on_device(xyz, graph_ptr_dev, n_atoms_host, bnn, edge_ptr_dev=None) -> BatchedMolGraph:
    """Complete distance graphs of a batch, built by csrc/batch.hip from coordinates [N,3] on the device."""
    from . import _lib
    dev = xyz.device
    B, N = int(n_atoms_host.shape[0]), int(xyz.shape[0])
    E3 = int((n_atoms_host * (n_atoms_host - 1)).sum())
    if edge_ptr_dev is None:
        edge_ptr = np.zeros(B + 1, dtype=np.int32)
        np.cumsum(n_atoms_host * (n_atoms_host - 1), out=edge_ptr[1:])
        edge_ptr_d = torch.from_numpy(edge_ptr).to(dev, non_blocking=True)
    else:
        edge_ptr_d = edge_ptr_dev
    ints = torch.empty((N + 1) + 5 * E3, dtype=torch.int32, device=dev)
    in_ptr = ints[:N + 1]
    src_s, dst_s, perm, inv_perm, out_epos = [ints[N + 1 + k * E3:N + 1 + (k + 1) * E3] for k in range(5)]
    ids = torch.empty(2 * E3, dtype=torch.int64, device=dev)
    d = torch.empty(E3, 1, dtype=torch.float32, device=dev)
    L = _lib.load()
    stream = torch._C._cuda_getCurrentRawStream(dev.index if dev.index is not None else torch.cuda.current_device())
    _lib.check(L.i3d_complete_graph_build(xyz.data_ptr(), graph_ptr_dev.data_ptr(), edge_ptr_d.data_ptr(), B, N, E3,
                                          in_ptr.data_ptr(), src_s.data_ptr(), dst_s.data_ptr(), perm.data_ptr(),
                                          inv_perm.data_ptr(), out_epos.data_ptr(), ids[:E3].data_ptr(),
                                          ids[E3:].data_ptr(), d.data_ptr(), stream), 'i3d_complete_graph_build')
    idx3 = GraphIndex(N, E3, B, in_ptr, perm, src_s, dst_s, in_ptr, out_epos, graph_ptr_dev, inv_perm,
                      int(n_atoms_host.max()) - 1)
    g3 = BatchedMolGraph(ids[:E3], ids[E3:], N, bnn, ndata={}, edata={'d': d}, index=idx3)
    g3.mark_ready()          # everything the 3D batch consists of is enqueued: an independent stream may wait for just this
    return g3
