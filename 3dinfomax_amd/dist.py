"""Data-parallel wiring over RCCL (torch.distributed backend "nccl" on ROCm; "gloo" in the tests).

The reference has no distributed code at all; BASELINE.json:north_star asks for molecule
sharding across the 8 GPUs of a node.  The path shards by molecule with three exchange points:
  C1  all-gather of the 3D-view embeddings for NT-Xent's negatives (losses._AllGatherRowsFn, backward =
      reduce-scatter),
  C2  gradient all-reduce (SUM: each rank's loss share already carries 1/B_global) through one persistent flat buffer,
      so a ring over xGMI moves one large message,
  C3  (optional, `setup(sync_bn=True)`) synchronised BatchNorm statistics (fp64 [sum, sumsq, count] all-reduce per BN,
      forward and backward - layers._Tail): with it the N-rank step equals the single-process step on the global
      batch bit for bit in structure (the reference normalises over the whole batch it is given); without it every rank
      normalises over its own 512 molecules, which is what torch's DistributedDataParallel does by default.

The three collective helpers below call RCCL directly for the "nccl" backend.  For "gloo" (CPU tests, and the
2-process-on-one-GPU parity test) device tensors are staged through the host and reduce-scatter is emulated with
all-reduce + slice, because gloo implements neither on HIP tensors.
"""
import os

import torch
import torch.distributed as dist

from . import _lib


def _is_gloo(group):
    return dist.get_backend(group) == 'gloo'


def all_reduce_sum(t, group=None):
    if _is_gloo(group) and t.is_cuda:
        h = t.cpu()
        dist.all_reduce(h, op=dist.ReduceOp.SUM, group=group)
        t.copy_(h)
    else:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t


def all_gather_rows(x, group=None):
    world = dist.get_world_size(group)
    x = x.contiguous()
    if _is_gloo(group):
        parts = [torch.empty_like(x, device='cpu') for _ in range(world)]
        dist.all_gather(parts, x.cpu(), group=group)
        return torch.cat(parts, 0).to(x.device)
    out = torch.empty((world * x.shape[0],) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
    dist.all_gather_into_tensor(out, x, group=group)
    return out


def reduce_scatter_rows(g, group=None):
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    g = g.contiguous()
    per = g.shape[0] // world
    if _is_gloo(group):
        h = g.cpu()
        dist.all_reduce(h, op=dist.ReduceOp.SUM, group=group)
        return h[rank * per:(rank + 1) * per].to(g.device)
    out = torch.empty((per,) + tuple(g.shape[1:]), dtype=g.dtype, device=g.device)
    dist.reduce_scatter_tensor(out, g, op=dist.ReduceOp.SUM, group=group)
    return out


def _row_index(counts, pad, device):
    """rows of the padded gather layout [world * pad, ...] that are real, in rank order"""
    key = (tuple(counts), pad, str(device))
    idx = _row_index_cache.get(key)
    if idx is None:
        idx = torch.cat([torch.arange(c, dtype=torch.int64) + r * pad for r, c in enumerate(counts)]).to(device)
        _row_index_cache.clear()
        _row_index_cache[key] = idx
    return idx


_row_index_cache = {}


def all_gather_rows_var(x, counts, group=None):
    """all-gather along dim 0 with a different number of rows per rank (`counts`, known on every rank: the shard plan is a
    function of the global batch): every rank contributes its rows padded to max(counts), the real rows are picked out."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    assert len(counts) == world and x.shape[0] == counts[rank], (counts, rank, x.shape)
    pad = max(counts)
    if pad == min(counts):
        return all_gather_rows(x, group)
    xp = torch.zeros((pad,) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
    xp[:x.shape[0]] = x
    return all_gather_rows(xp, group).index_select(0, _row_index(counts, pad, x.device))


def reduce_scatter_rows_var(g, counts, group=None):
    """backward of all_gather_rows_var: g [sum(counts), ...] -> this rank's rows of the sum over ranks"""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    pad = max(counts)
    if pad == min(counts):
        return reduce_scatter_rows(g, group)
    gp = torch.zeros((world * pad,) + tuple(g.shape[1:]), dtype=g.dtype, device=g.device)
    gp.index_copy_(0, _row_index(counts, pad, g.device), g.contiguous())
    return reduce_scatter_rows(gp, group)[:counts[rank]]


def warm_up(device, steps=2):
    """Run a tiny pre-training step on `device` BEFORE `init_process_group`.

    Measured on MI355X / ROCm 7 (tools/step_segments.py --pg-first [--prepare-step]): a process that creates its RCCL
    communicator before it has launched its own kernels and started autograd's worker thread runs EVERY later step ~7 %
    slower (3.39-3.45 ms against 3.13-3.21 ms; the slowdown is per launch, on all segments of the step, and is there
    even when no collective is ever called).  Two throw-away steps of a small model first - HIP library loaded, compute
    and side streams created, autograd thread running, allocator pools populated - and the effect is gone."""
    from . import synth
    from .graph import batch, bond_graph, complete_graph
    from .losses import NTXent
    from .net3d import Net3D
    from .pna import PNA
    device = torch.device(device)
    mols = synth.make_dataset(16, seed=0)
    g2 = batch([bond_graph(m) for m in mols]).to(device)
    g3 = batch([complete_graph(m) for m in mols]).to(device)
    pna = PNA(hidden_dim=32, target_dim=16, propagation_depth=2, aggregators=['mean', 'max', 'min', 'std'],
              scalers=['identity', 'amplification', 'attenuation'], readout_aggregators=['min', 'max', 'mean'],
              avg_d=1.0, device=device).to(device).train()
    net = Net3D(node_dim=0, edge_dim=1, hidden_dim=16, target_dim=16, propagation_depth=1, avg_d=1.0,
                readout_aggregators=['min', 'max', 'mean']).to(device).train()
    loss_fn = NTXent(tau=0.1)
    for _ in range(steps):
        a, b = g2.local_copy(), g3.local_copy()
        loss_fn(pna(a), net(b)).backward()
    torch.cuda.synchronize(device)


# I3D_NATIVE_SYNC_BN=0: synchronised BatchNorm only on the per-block Python path (round-2 behaviour)
NATIVE_SYNC_BN = os.environ.get('I3D_NATIVE_SYNC_BN', '1') != '0'
# provider of the library's process-wide collectives (csrc/comm.hip): "peer" (one-shot peer-write exchange over IPC-mapped
# mailboxes, csrc/peer.hip: the default), "rccl" (a communicator of the library's own), "callbacks" (host-staged through the
# torch group: functional check)
SYNC_PROVIDER = os.environ.get('I3D_SYNC_PROVIDER', 'peer')
_native_sync = None      # dict(group, provider, keep-alive objects, handles to release) while the collectives are set


def native_sync_active():
    return _native_sync is not None


def native_sync_provider():
    return _native_sync['provider'] if _native_sync is not None else None


_fallbacks = []      # every provider fallback taken in this process, in order (bench.py reports it next to the provider in use)


def native_sync_fallbacks():
    return list(_fallbacks)


def _all_ok(ok, group):
    """every rank's `ok` ANDed: a set-up step either holds on every rank or on none (the ranks then take the same fallback)"""
    flags = [None] * dist.get_world_size(group)
    dist.all_gather_object(flags, bool(ok), group=group)
    return all(flags)


# I3D_PEER_SELFTEST=0: skip the pattern exchange at set-up (world > 1 only; ~100 small launches)
PEER_SELFTEST = True


def _peer_selftest(L, state, device, with_side):
    """Pattern exchange through the installed peer provider (default context on the current stream, the 3D network's context on
    its side stream): all-gathers and fp64 rank-order sums of values every rank can predict, payload sizes from one word to the
    BatchNorm vectors' [3 x 200] and beyond, enough rounds to reuse every slot many times.  -> bool (this rank saw only correct
    values and no time-out); the caller combines the ranks' verdicts."""
    from . import streams
    group = state['group']
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    todo = [torch.cuda.current_stream(device)] + ([streams.side_stream(device)] if with_side else [])
    wrong = 0
    try:
        for stream in todo:
            stream.synchronize()
            with torch.cuda.stream(stream):
                ptr = stream.cuda_stream
                bad = torch.zeros((), dtype=torch.int64, device=device)      # (a counter per stream: nothing crosses streams unordered)
                for rnd in range(48):
                    count = (1, 7, 64, 600, 601, 1201, 4097)[rnd % 7]
                    base = torch.arange(count, dtype=torch.float32, device=device)
                    send = base * (rank + 1) + (rnd + 0.25)
                    recv = torch.empty(world, count, dtype=torch.float32, device=device)
                    if L.i3d_collectives_all_gather_f32(send.data_ptr(), recv.data_ptr(), count, ptr) != 0:
                        return False
                    want = torch.stack([base * (r + 1) + (rnd + 0.25) for r in range(world)])
                    bad += (recv != want).sum()
                    b64 = torch.arange(count, dtype=torch.float64, device=device)
                    buf = b64 / (rank + 3) + rnd
                    if L.i3d_collectives_all_reduce_f64(buf.data_ptr(), count, ptr) != 0:
                        return False
                    tot = torch.zeros(count, dtype=torch.float64, device=device)
                    for r in range(world):          # rank order: the order the exchange adds in
                        tot = tot + (b64 / (r + 3) + rnd)
                    bad += (buf != tot).sum()
                wrong += int(bad.item())
        torch.cuda.synchronize(device)
        ok = wrong == 0 and all(L.i3d_peer_status(c) == 0 for c in state['peers'])
        if not ok and _lib.TEST_HOOKS['verbose_selftest']:
            print(f'[rank {rank}] peer self-test: {wrong} wrong values, status {[L.i3d_peer_status(c) for c in state["peers"]]}', flush=True)
    except Exception:      # noqa: BLE001 - a failing exchange must end in the fallback, not in a crash of one rank
        if _lib.TEST_HOOKS['verbose_selftest']:
            import traceback
            traceback.print_exc()
        ok = False
    if _lib.TEST_HOOKS['peer_selftest_fail'] == str(rank):      # test hook: this rank "saw a wrong value"
        ok = False
    return ok


def _peer_context(L, group, device, timeout_s):
    """One mailbox of this rank, exported, every rank's handle gathered through `group`, the peers' mailboxes mapped.  Returns the
    context or None - None on EVERY rank when any rank failed at any step (nothing stays allocated or mapped then)."""
    import ctypes
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    hb = L.i3d_peer_handle_bytes()
    box, handle = ctypes.c_void_p(), ctypes.create_string_buffer(hb)
    with torch.cuda.device(device):
        ok = L.i3d_peer_alloc(ctypes.byref(box), handle) == 0
        if _lib.TEST_HOOKS['peer_fail'] == str(rank):      # test hook: this rank "cannot" set the exchange up
            ok = False
        msgs = [None] * world
        dist.all_gather_object(msgs, (bool(ok), bytes(handle.raw)), group=group)       # (every rank takes part, failed or not)
        ctx = ctypes.c_void_p()
        if all(m[0] for m in msgs):
            ok = L.i3d_peer_open(box, b''.join(m[1] for m in msgs), rank, world, float(timeout_s), ctypes.byref(ctx)) == 0
            if not ok:
                ctx = ctypes.c_void_p()
        else:
            ok = False
        if not _all_ok(ok, group):
            if ctx:
                L.i3d_peer_close(ctx)              # (frees the mailbox too)
            elif box:
                L.i3d_peer_free(box)
            return None
    return ctx


def enable_native_sync(group, device, provider=None, timeout_s=0.0):
    """Synchronised BatchNorm from INSIDE the C sequencers (csrc/comm.hip): the library's BatchNorm entry points run their
    collectives themselves, on the stream they are called on, while a process-wide collective table is set.

    provider "peer" (default): csrc/peer.hip - every rank maps every other rank's mailbox (hipIpc handles gathered through
    `group`, any backend), the BatchNorm kernels write their small vectors straight into the peers' mailboxes and wait for
    the peers' flags; the statistics finalisation does it inside its one launch.  The 3D network's side stream gets a context
    of its own and keeps running beside the 2D network.  "rccl": a communicator of the library's own, its id distributed
    through `group` ("nccl" backend only); "callbacks": host-staged through `group` (functional check; the default for a gloo
    group is still "peer": IPC works between processes that share a GPU).  With "rccl" / "callbacks" ONE stream issues the
    collectives (the 3D network joins the 2D network's stream) and the early gradient all-reduce is off (two communicators
    must not be in flight at once, GradReducer.overlap).  Idempotent for the same group and provider; returns False when it
    cannot be set up (the per-block path then synchronises).

    Threading constraint: the 3D network's context is bound to the side stream of THE THREAD THAT CALLS THIS FUNCTION
    (streams.side_stream is per thread).  The training loop - forward passes included - must run on that thread: a forward
    pass on another thread issues the 3D network's collectives on a stream without a context of its own, the default context
    then serves two streams and its slot reuse (which relies on ONE stream's order) is no longer protected; the symptom is an
    exchange time-out, not a silent wrong sum (the tags still have to match).  `multithreaded_seeds`-style runs (one training per
    thread) are single-process, without data parallelism, and never come here."""
    global _native_sync
    import ctypes
    from . import _lib, streams
    provider = provider or SYNC_PROVIDER
    if provider not in ('peer', 'rccl', 'callbacks'):
        raise ValueError(f'I3D_SYNC_PROVIDER={provider!r}: peer, rccl or callbacks')
    if provider == 'rccl' and _is_gloo(group):
        provider = 'callbacks'
    if _native_sync is not None:
        if _native_sync['group'] is group and _native_sync['provider'] == provider:
            return True
        disable_native_sync()
    L = _lib.load()
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    device = torch.device(device)
    _one_launch_needs_a_gpu_per_rank(device, group)
    scratch = torch.zeros(1 << 20, dtype=torch.uint8, device=device)
    import threading
    state = dict(group=group, provider=provider, device=device, net3d_stream=streams.NET3D_STREAM, keep=[scratch], comm=None, peers=[],
                 thread=threading.get_ident())
    if provider == 'peer':
        torch.cuda.synchronize(device)
        main = _peer_context(L, group, device, timeout_s)
        ok = main is not None
        if ok:
            state['peers'].append(main)
            ok = L.i3d_set_collectives_peer(main, scratch.data_ptr(), scratch.numel()) == 0
        side = None
        if _all_ok(ok, group) and streams.NET3D_STREAM:
            # the 3D network's stream: own mailbox, own sequence, own scratch (every rank creates both, in this order)
            side_scratch = torch.zeros(1 << 20, dtype=torch.uint8, device=device)
            side = _peer_context(L, group, device, timeout_s)
            ok = side is not None
            if ok:
                state['peers'].append(side)
                state['keep'].append(side_scratch)
                ok = L.i3d_peer_bind_stream(side, ctypes.c_void_p(streams.side_stream(device).cuda_stream), side_scratch.data_ptr(),
                                            side_scratch.numel()) == 0
        if _all_ok(ok, group) and world > 1 and PEER_SELFTEST:
            # Before any BatchNorm depends on it: exchange known patterns through BOTH contexts and compare bit for bit.  The
            # protocol rests on properties of the fabric nobody could test without a multi-GPU node (an 8-byte system-scope
            # store arrives whole at a peer's uncached memory; mapped mailboxes of another device are pollable): if they do
            # not hold here, some rank sees a wrong value or a time-out NOW, and every rank takes the next provider.
            ok = _peer_selftest(L, state, device, side is not None)
        if _all_ok(ok, group):
            _native_sync = state
            streams.BOUND_THREAD = state['thread'] if side is not None else None
            return True
        # some rank could not set the exchange up (no IPC between these devices, no uncached memory...): every rank tears it
        # down and takes the next provider - the decision is collective, the ranks never run different providers
        L.i3d_set_collectives(None)
        for ctx in state['peers']:
            L.i3d_peer_close(ctx)
        state['peers'], state['keep'] = [], [scratch]
        import warnings
        warnings.warn('3dinfomax_amd.dist: the peer-write exchange could not be set up (or failed its pattern exchange) on every rank '
                      f'({L.i3d_last_error().decode() if L.i3d_last_error() else "no message"}); falling back to '
                      + ('host-staged callbacks' if _is_gloo(group) else 'the RCCL provider'))
        provider = state['provider'] = 'callbacks' if _is_gloo(group) else 'rccl'
        _fallbacks.append(f'peer -> {provider}')
    if provider == 'rccl':
        if not _all_ok(L.i3d_rccl_available(), group):
            _fallbacks.append('rccl -> per-block (librccl not loadable on some rank)')
            return False
        buf = ctypes.create_string_buffer(128)
        if rank == 0:
            _lib.check(L.i3d_rccl_unique_id(buf), 'i3d_rccl_unique_id')
        box = [bytes(buf.raw)]
        dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not dist.group.WORLD else 0, group=group)
        comm = ctypes.c_void_p()
        torch.cuda.synchronize(device)
        _lib.check(L.i3d_rccl_init(box[0], rank, world, ctypes.byref(comm)), 'i3d_rccl_init')
        _lib.check(L.i3d_set_collectives_rccl(comm, world, scratch.data_ptr(), scratch.numel()), 'i3d_set_collectives_rccl')
        state['comm'] = comm
    else:
        base = scratch.data_ptr()

        def view(ptr, count, dtype):
            n = count * (8 if dtype == torch.float64 else 4)
            return scratch[ptr - base:ptr - base + n].view(dtype)

        def all_gather(_user, send, recv, count, stream):
            try:
                torch.cuda.synchronize(device)          # (functional path: the stream's producers are done)
                mine = view(send, count, torch.float32).cpu()
                parts = [torch.empty_like(mine) for _ in range(world)]
                dist.all_gather(parts, mine, group=group)
                view(recv, count * world, torch.float32).copy_(torch.cat(parts))
                torch.cuda.synchronize(device)
                return 0
            except Exception:      # noqa: BLE001 - reported through the C return code
                import traceback
                traceback.print_exc()
                return -2

        def all_reduce(_user, buf, count, stream):
            try:
                torch.cuda.synchronize(device)
                t = view(buf, count, torch.float64)
                h = t.cpu()
                dist.all_reduce(h, op=dist.ReduceOp.SUM, group=group)
                t.copy_(h)
                torch.cuda.synchronize(device)
                return 0
            except Exception:      # noqa: BLE001
                import traceback
                traceback.print_exc()
                return -2
        cb_g, cb_r = _lib.ALL_GATHER_F32(all_gather), _lib.ALL_REDUCE_F64(all_reduce)
        c = _lib.Collectives(world, cb_g, cb_r, None, scratch.data_ptr(), scratch.numel())
        _lib.check(L.i3d_set_collectives(ctypes.byref(c)), 'i3d_set_collectives')
        state['keep'] += [cb_g, cb_r, c]
    # ONE stream issues the collectives of a communicator, in the same order on every rank: the 3D network joins the
    # 2D network's stream.  Its fused edge stage stays: the forward statistics go through i3d_bn_finalize_partials like
    # everyone's, the backward sums of its two BatchNorms are exchanged between its own kernels (csrc/net3d_edge.hip:
    # sync_backward_sums)
    streams.NET3D_STREAM = False
    _native_sync = state
    return True


def disable_native_sync():
    """Switch the library's collectives off and release what enable_native_sync created (communicator / mailboxes); restores
    the 3D network's side stream.  Collective when a peer or RCCL provider was set: every rank calls it."""
    global _native_sync
    if _native_sync is None:
        return
    from . import _lib, streams
    st, _native_sync = _native_sync, None
    L = _lib.load()
    if st['device'].type == 'cuda':
        torch.cuda.synchronize(st['device'])
    L.i3d_set_collectives(None)
    if st['peers'] or st['comm'] is not None:
        try:        # nobody unmaps a mailbox a peer's kernel may still write to
            dist.barrier(group=st['group'])
        except Exception:      # noqa: BLE001 - the group may be gone already (interpreter shutdown)
            pass
    for ctx in st['peers']:
        L.i3d_peer_close(ctx)
    if st['comm'] is not None:
        L.i3d_rccl_destroy(st['comm'])
    streams.NET3D_STREAM = st['net3d_stream']
    streams.BOUND_THREAD = None


def _one_launch_needs_a_gpu_per_rank(dev, group):
    """The one-launch BatchNorm backward (csrc/bn.hip: bn_bwd_fused_kernel) keeps every workgroup of its grid resident across an
    in-launch wait; that is guaranteed for the launches of ONE process on a GPU.  Ranks that share a physical GPU (the one-GPU
    data-parallel tests; never a production layout) compete for its CUs, so every rank switches to the two-pass kernels -
    collectively: the ranks must not run different summation orders."""
    if dev is None or torch.device(dev).type != 'cuda' or dist.get_world_size(group) == 1:
        return
    pr = torch.cuda.get_device_properties(dev)
    mine = (os.uname().nodename, getattr(pr, 'uuid', None) and str(pr.uuid),
            tuple(getattr(pr, k, 0) for k in ('pci_domain_id', 'pci_bus_id', 'pci_device_id')))
    seen = [None] * dist.get_world_size(group)
    dist.all_gather_object(seen, mine, group=group)
    if len(set(seen)) < len(seen):
        from . import ops
        ops.set_bn_bwd_one_launch(False)


def setup(modules, loss=None, group=None, sync_bn=False, broadcast=True):
    """Attach `group` to every FCLayer (sync-BN) of `modules` and to the loss; broadcast rank-0 weights.

    sync_bn: BatchNorm statistics over all ranks (the reference's statistics are over the whole batch it is given).  With
    the library's own collectives (enable_native_sync) every code path - whole-model sequencer included - synchronises
    inside the C calls and the modules need no group; otherwise the per-block Python path does it (FCLayer.sync_group)."""
    from .layers import FCLayer
    group = group if group is not None else dist.group.WORLD
    _one_launch_needs_a_gpu_per_rank(next((p.device for m in modules for p in m.parameters() if p.is_cuda), None), group)
    native = False
    if sync_bn and NATIVE_SYNC_BN:
        dev = next((p.device for m in modules for p in m.parameters() if p.is_cuda), None)
        native = dev is not None and enable_native_sync(group, dev)
    if not native:
        disable_native_sync()
    for m in modules:
        for sub in m.modules():
            if isinstance(sub, FCLayer):
                sub.sync_group = group if (sync_bn and not native) else None
        if broadcast:
            for t in list(m.parameters()) + list(m.buffers()):
                if _is_gloo(group) and t.is_cuda:
                    h = t.data.cpu()
                    dist.broadcast(h, src=0, group=group)
                    t.data.copy_(h)
                else:
                    dist.broadcast(t.data, src=0, group=group)
    if loss is not None and hasattr(loss, 'attach_group'):
        loss.attach_group(group)
    return group


# I3D_OVERLAP_ALLREDUCE=0: one all-reduce of the whole gradient buffer after the backward pass (round-1 behaviour)
OVERLAP_ALLREDUCE = True


class GradReducer:
    """C2: SUM all-reduce of all parameter gradients through ONE persistent flat buffer.

    The models' backward passes (tape.ModelFn) deliver their parameter gradients straight into slices of the buffer
    (one multi-tensor copy per model) and autograd stores those slices as `p.grad`, so a step costs one all-reduce
    (20 MB for PNA + Net3D: ring all-reduce over xGMI is per-link bound, one large message beats many small ones)
    and no copy back.  Gradients that arrive another way (per-block autograd nodes, I3D_FUSED_MODEL=0) are copied in
    here and `p.grad` is re-pointed.  (The per-tensor version - 110 x copy_ - cost ~4 ms of host time per step.)"""

    def __init__(self, params, group=None):
        from . import tape
        self.params = [p for p in params if p.requires_grad]
        self.group = group if group is not None else dist.group.WORLD
        p0 = self.params[0]
        sizes = [p.numel() for p in self.params]
        self.flat = torch.zeros(sum(sizes), dtype=p0.dtype, device=p0.device)
        self.views = [v.view_as(p) for v, p in zip(self.flat.split(sizes), self.params)]
        self.view_of = {id(p): v for p, v in zip(self.params, self.views)}
        self.span_of, o = {}, 0
        for p, n in zip(self.params, sizes):
            self.span_of[id(p)] = (o, o + n)
            o += n
        self._pending, self._launched = [], []        # async all-reduces of this step, element ranges they cover
        self.early_spans = []                         # [start, end) element ranges reduced early, fixed by the model structure
        self.early_module = None                      # id() of the ONE module whose backward pass may start them
        self._late, self._late_used = None, False     # gradients that arrive for a slice AFTER its early all-reduce started
        self.overlap = OVERLAP_ALLREDUCE
        self._agreed = None                           # None: not checked yet; True / False: every rank has the same early plan
        self._tape = tape

    def attach(self, modules):
        """route the gradients of these modules (whole-model tape nodes) into the buffer"""
        for m in modules:
            if any(p.requires_grad for p in m.parameters()):
                self._tape.register_grad_sink(m, self._sink, self.view_of)
                self._tape.model_state(m).reducer = self
                self._plan_early(m)

    def _plan_early(self, module):
        """The slices of the flat buffer that are reduced EARLY (while the backward pass is still running) are a function of
        the model structure alone - head + upper half of the message-passing layers of a PNA - so that every rank issues the
        same collectives in the same order whichever code path its backward pass takes (a rank that fell back to the
        Python-sequenced path reduces the same slices, later)."""
        gnn = getattr(module, 'node_gnn', None)
        layers = list(getattr(gnn, 'mp_layers', [])) if gnn is not None else []
        if len(layers) < 2 or not hasattr(module, 'output'):
            return
        if self.early_spans:
            # ONE module per reducer starts collectives from inside its backward pass: with two, a rank whose second module
            # fell back to the Python-sequenced path would issue that module's early slices in reduce(), i.e. in another
            # order than a rank that started them in the backward pass - RCCL needs the same order everywhere.  The other
            # modules' gradients travel with the complement.
            return
        self.early_module = id(module)
        early = [p for layer in layers[len(layers) // 2:] for p in layer.parameters()] + list(module.output.parameters())
        spans = sorted(self.span_of[id(p)] for p in early if id(p) in self.span_of)
        for a, b in spans:
            if self.early_spans and self.early_spans[-1][1] == a:
                self.early_spans[-1][1] = b
            else:
                self.early_spans.append([a, b])
        self.early_spans.sort()

    def launch_async(self, module=None):
        """Start the all-reduce of the EARLY slices now (their gradients are final in stream order and live in the flat
        buffer): a model backward calls this after the part of the pass that produces them (pna_native.PNAModelFn: head +
        upper half of the layers) while the rest is still to be enqueued, so the collective runs next to it.  `module`: the
        caller - only the module the plan was made for may start it (another module's backward pass says nothing about
        whether THOSE gradients are final); None: reduce() itself."""
        if self._launched or not self._agreed:       # (no early collective before the ranks have agreed on the plan)
            return
        if module is not None and native_sync_provider() == 'rccl':
            # The library's own RCCL communicator issues the BatchNorm collectives of the rest of the backward pass on the
            # compute stream; an asynchronous all-reduce of torch's communicator next to them would be two communicators in
            # flight without a common order across ranks (documented as unsafe unless both kernels are always co-resident).
            # The peer provider has no communicator: the early all-reduce stays on with it.  Only the start from INSIDE a
            # backward pass is switched off: reduce() (module None; the backward pass and its BatchNorm collectives are enqueued)
            # still reduces these slices - it skips them in its complement (round 5: it used to return here too, which left
            # the head's and the upper layers' gradients un-reduced whenever the RCCL provider was in use at world > 1).
            return
        if module is not None and id(module) != self.early_module:
            return
        for a, b in self.early_spans:
            t = self.flat[a:b]
            if _is_gloo(self.group):
                all_reduce_sum(t, self.group)                       # host-staged, synchronous: same arithmetic
            else:
                self._pending.append(dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
            self._launched.append((a, b))

    def _sink(self, params, grads):
        """Gradients of one model backward -> what autograd should store / accumulate.

        `p.grad` empty: the new gradient is copied into the parameter's slice of the flat buffer and the slice is handed
        out.  `p.grad` already set (gradient accumulation over several backward passes, `zero_grad(set_to_none=False)`,
        two forwards before one backward): the slice IS (or will be copied from) the accumulated value - it must not be
        overwritten; the fresh gradient goes back to autograd, whose AccumulateGrad adds it onto `p.grad` in place."""
        out, dst, src = [], [], []
        for p, g in zip(params, grads):
            d = self.view_of.get(id(p))
            if d is not None and g is not None and g is not d and p.grad is not None and self._in_flight(id(p)):
                # a second backward pass of this step (gradient accumulation) delivers a gradient for a slice whose all-reduce
                # has already been started by the first: adding it onto `p.grad` - the slice - would race with the collective
                # and leave "reduced first pass + local second pass".  It is summed in a side buffer, reduced on its own in
                # reduce() and added to the slice afterwards.
                if self._late is None:
                    self._late = torch.zeros_like(self.flat)
                a, b = self.span_of[id(p)]
                self._late[a:b].view_as(p).add_(g)
                self._late_used = True
                out.append(None)
                continue
            if d is None or g is None or g is d or p.grad is not None:
                out.append(g)
                continue
            dst.append(d)
            src.append(g)
            out.append(d)
        if dst:
            torch._foreach_copy_(dst, src)
        return out

    def _in_flight(self, pid):
        if not self._launched:
            return False
        a, b = self.span_of[pid]
        return any(a < hi and lo < b for lo, hi in self._launched)

    def reduce(self):
        grads = [p.grad for p in self.params]
        # already in the buffer: the very view, or (when autograd's AccumulateGrad stored it) an alias of that view
        todo = [(v, g) for v, g in zip(self.views, grads)
                if g is not v and not (g is not None and g.data_ptr() == v.data_ptr() and g.shape == v.shape
                                       and g.is_contiguous())]
        if todo:
            missing = [v for v, g in todo if g is None]
            if missing:                                 # a parameter without gradient this step counts as zero
                torch._foreach_zero_(missing)
            have = [(v, g) for v, g in todo if g is not None]
            if have:
                torch._foreach_copy_([v for v, _ in have], [g for _, g in have])
        if self._agreed is None and dist.get_world_size(self.group) > 1:
            # once, in the first reduce() (a point every rank passes with nothing else in flight): do all ranks hold the same
            # early plan?  A rank whose reducer was created without its modules has none - then nobody splits the reduction.
            mine = torch.tensor([len(self.early_spans), sum(b - a for a, b in self.early_spans),
                                 self.early_spans[0][0] if self.early_spans else -1], dtype=torch.float64)
            both = torch.cat([mine, -mine]).to(self.flat.device if not _is_gloo(self.group) else 'cpu')
            dist.all_reduce(both, op=dist.ReduceOp.MAX, group=self.group)
            both = both.cpu()
            self._agreed = bool(self.overlap and self.early_spans and torch.equal(both[:3], -both[3:]))
        if self._agreed is None and _lib.TEST_HOOKS['force_early_allreduce']:
            self._agreed = bool(self.overlap and self.early_spans)      # test hook: the split reduction at world 1 (one-GPU box)
        if self._agreed:
            # the same collectives in the same order on every rank: the early slices (now, if the backward pass did not
            # start them), then the complement; then wait for the early ones
            self.launch_async()
            o = 0
            for a, b in self.early_spans + [[self.flat.numel(), self.flat.numel()]]:
                if a > o:
                    all_reduce_sum(self.flat[o:a], self.group)
                o = max(o, b)
            if self._late_used:                       # (every rank ran the same number of backward passes: same decision)
                for a, b in self._launched:
                    all_reduce_sum(self._late[a:b], self.group)
            for w in self._pending:
                w.wait()
            if self._late_used:
                for a, b in self._launched:
                    self.flat[a:b].add_(self._late[a:b])
                self._late.zero_()
                self._late_used = False
        else:
            all_reduce_sum(self.flat, self.group)
        self._pending, self._launched = [], []
        if todo:
            for p, v in zip(self.params, self.views):
                p.grad = v


_reducers = {}


def grad_reducer(params, group=None, modules=None):
    """The GradReducer of this parameter list (created on first use); `modules`: attach their backward passes to it."""
    group = group if group is not None else dist.group.WORLD
    key = (id(group), tuple(id(p) for p in params))
    red = _reducers.get(key)
    if red is None:
        red = _reducers[key] = GradReducer(params, group)
        if modules:
            red.attach(modules)
    return red


def allreduce_grads(params, group=None):
    """Sum the gradients over ranks (C2); see GradReducer."""
    grad_reducer(params, group).reduce()


def global_loss(loss_share, group=None):
    """Sum of the per-rank loss shares = the reference's full-batch loss (for logging)."""
    return all_reduce_sum(loss_share.detach().clone(), group if group is not None else dist.group.WORLD)


def shard_plan(sizes, world):
    """Partition of a global batch over `world` ranks, deterministic (every rank computes the same plan from the same
    batch): molecule counts differ by at most one (nothing is dropped), and the ATOM counts are balanced - molecules are
    dealt largest first in snake order (0..w-1, w-1..0, ...), because the work of a rank follows its atoms (edges ~ atoms,
    complete-graph edges ~ atoms^2).  -> list of index lists (ascending inside a rank)."""
    n = len(sizes)
    cap = [n // world + (1 if r < n % world else 0) for r in range(world)]
    order = sorted(range(n), key=lambda i: (-int(sizes[i]), i))
    plan = [[] for _ in range(world)]
    r, step = 0, 1
    for i in order:
        while len(plan[r]) >= cap[r]:                 # a full rank is skipped (only the +1 remainder ranks differ)
            r, step = _snake_next(r, step, world)
        plan[r].append(i)
        r, step = _snake_next(r, step, world)
    return [sorted(p) for p in plan]


def _snake_next(r, step, world):
    if world == 1:
        return 0, 1
    nr = r + step
    if nr < 0 or nr >= world:
        return r, -step                               # turn around: the end rank takes two in a row
    return nr, step


def shard_molecules(mols, rank, world, balance=None):
    """This rank's molecules of the (already shuffled) global batch.  balance=None: contiguous ranges, the first
    len(mols) % world ranks hold one molecule more (nothing is dropped); balance='atoms': shard_plan on `n_atoms`."""
    if balance == 'atoms':
        return [mols[i] for i in shard_plan([m.n_atoms for m in mols], world)[rank]]
    n = len(mols)
    start = rank * (n // world) + min(rank, n % world)
    return mols[start:start + n // world + (1 if rank < n % world else 0)]


def shard_counts(n, world):
    """molecules per rank of both shard_molecules modes"""
    return [n // world + (1 if r < n % world else 0) for r in range(world)]
