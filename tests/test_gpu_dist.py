"""-m gpu: the data-parallel path THROUGH THE HIP KERNELS.  Two processes share cuda:0 (gloo stages the tiny
collectives through the host - RCCL needs one device per rank, the driver's 8-GPU node exercises that path), each
owns half of the molecules; with synchronised BatchNorm, all-gathered negatives and summed gradients the sharded
step must reproduce the single-process full-batch step: loss, parameter gradients and BN running statistics."""
import importlib
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
WORLD = 2
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _models(amd, seed=0):
    from helpers import NET3D_YML, PNA_YML
    torch.manual_seed(seed)
    pna = amd.PNA(avg_d=1.0, device='cuda:0', **dict(PNA_YML, propagation_depth=2, hidden_dim=40, readout_hidden_dim=40,
                                                      target_dim=32))
    net = amd.Net3D(node_dim=0, edge_dim=1, avg_d=1.0, **dict(NET3D_YML, target_dim=32))
    with torch.no_grad():       # O(1) pre-BN scale so that the statistics matter
        for m in (pna, net):
            for n, p in m.named_parameters():
                if n.endswith('linear.weight'):
                    p.mul_(p.shape[1] * 0.7)
    return pna.cuda().train(), net.cuda().train()


def _batch(amd, mols):
    return (amd.batch([amd.bond_graph(m) for m in mols]).to('cuda:0'),
            amd.batch([amd.complete_graph(m) for m in mols]).to('cuda:0'))


def _worker(rank, port, path, native_sync, WORLD=WORLD):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), I3D_NATIVE_SYNC_BN='1' if native_sync else '0',
                      I3D_SYNC_PROVIDER=native_sync or 'peer', I3D_PEER_TIMEOUT_S='20')
    if native_sync == 'peer_selftest_fails_on_rank_0':      # the exchange is set up but ONE rank reads a wrong value in the pattern exchange
        os.environ.update(I3D_SYNC_PROVIDER='peer', I3D_TESTING='1', I3D_TEST_PEER_SELFTEST_FAIL='0')
        import warnings
        warnings.simplefilter('ignore')
        native_sync = 'callbacks'
    if native_sync == 'peer_fails_on_rank_1':       # the exchange cannot be set up on ONE rank: every rank must take the fallback
        os.environ.update(I3D_SYNC_PROVIDER='peer', I3D_TESTING='1', I3D_TEST_PEER_FAIL='1')
        import warnings
        warnings.simplefilter('ignore')
        native_sync = 'callbacks'
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    dist.init_process_group('gloo', rank=rank, world_size=WORLD)
    amd = importlib.import_module('3dinfomax_amd')
    adist = importlib.import_module('3dinfomax_amd.dist')
    native = importlib.import_module('3dinfomax_amd.pna_native')
    calls = []
    real_run = native.run
    native.run = lambda *a, **k: (calls.append(1), real_run(*a, **k))[1]
    mols = amd.synth.make_dataset(16, seed=21)
    pna, net = _models(amd)
    loss_fn = amd.NTXent(tau=0.1)
    adist.setup([pna, net], loss_fn, sync_bn=True)      # global-batch BN statistics: equals the single-process step
    assert adist.native_sync_active() == bool(native_sync) and adist.native_sync_provider() == (native_sync or None)
    g2, g3 = _batch(amd, adist.shard_molecules(mols, rank, WORLD))
    params = list(pna.parameters()) + list(net.parameters())
    if rank == 0:    # one rank delivers its gradients straight into the all-reduce buffer, the other through the copy path
        adist.grad_reducer(params, modules=[pna, net])
    share = loss_fn(pna(g2), net(g3))
    share.backward()
    adist.allreduce_grads(params)
    total = adist.global_loss(share)
    # the library's own collectives: the whole-model C sequencer takes the synchronised step (round 2: per-block path only)
    assert len(calls) == (1 if native_sync else 0), calls
    if rank == 0:
        out = {'loss': total.item()}
        for tag, m in (('pna', pna), ('net', net)):
            for k, p in m.named_parameters():
                out[f'g/{tag}/{k}'] = p.grad.cpu().numpy()
            for k, b in m.named_buffers():
                out[f'b/{tag}/{k}'] = b.cpu().numpy()
        np.savez(path, **out)
    if native_sync:      # the provider's own two collectives (what the BatchNorm entry points call when no fused exchange applies)
        L = importlib.import_module('3dinfomax_amd._lib').load()
        ops = importlib.import_module('3dinfomax_amd.ops')
        stream = torch.cuda.current_stream().cuda_stream
        for count in (1, 37, 1201):
            send = torch.arange(count, device='cuda', dtype=torch.float32) * (rank + 1) + 0.25
            recv = torch.empty(WORLD, count, device='cuda')
            ops.check(L.i3d_collectives_all_gather_f32(send.data_ptr(), recv.data_ptr(), count, stream), 'all_gather')
            want = torch.stack([torch.arange(count, dtype=torch.float32) * (r + 1) + 0.25 for r in range(WORLD)])
            assert torch.equal(recv.cpu(), want), (count, recv, want)
            buf = (torch.arange(count, dtype=torch.float64) / (rank + 3)).cuda()       # (inputs from the CPU: same bits as `tot` below)
            ops.check(L.i3d_collectives_all_reduce_f64(buf.data_ptr(), count, stream), 'all_reduce')
            tot = torch.zeros(count, dtype=torch.float64)
            for r in range(WORLD):          # rank order: the same bits on every rank
                tot = tot + torch.arange(count, dtype=torch.float64) / (r + 3)
            if native_sync == 'peer':       # sums formed in rank order by every rank
                assert torch.equal(buf.cpu(), tot), (count, buf.cpu() - tot)
            else:                           # (the host-staged provider adds in gloo's order)
                assert torch.allclose(buf.cpu(), tot, rtol=1e-14, atol=0), (count, buf.cpu() - tot)
    if native_sync == 'peer':
        L = importlib.import_module('3dinfomax_amd._lib').load()
        st = adist._native_sync
        # both contexts (the 2D network's stream and the 3D network's side stream) have exchanged, nothing timed out
        assert len(st['peers']) == 2 and all(L.i3d_peer_sequence(c) > 0 and L.i3d_peer_status(c) == 0 for c in st['peers'])
    adist.disable_native_sync()
    assert not adist.native_sync_active() and importlib.import_module('3dinfomax_amd.streams').NET3D_STREAM
    dist.destroy_process_group()


@pytest.mark.parametrize('native_sync,world', [('peer', 2), ('peer', 4), ('callbacks', 2), ('peer_fails_on_rank_1', 2),
                                               ('peer_selftest_fails_on_rank_0', 2), (False, 2)])
def test_two_rank_sharded_step_equals_full_batch(tmp_path, native_sync, world):
    """BatchNorm synchronised inside the C sequencers (csrc/comm.hip) - the whole-model sequencer runs - through 'peer': the
    one-shot peer-write exchange (csrc/peer.hip: each process maps the other's mailbox through hipIpc - which works between
    processes that share a GPU - and the BatchNorm kernels exchange their vectors themselves; the 3D network keeps its side
    stream with a context of its own) or 'callbacks': host-staged through gloo; False: the per-block Python path of round 2.
    (RCCL refuses two ranks on one device: its provider has a world-1 test below.)  'peer_fails_on_rank_1': one rank cannot set
    the exchange up - the decision to fall back is collective, both ranks end on the same provider; 'peer_selftest_fails_on_rank_0':
    the exchange is up but one rank's pattern exchange at set-up (dist._peer_selftest: what would catch a fabric that does not
    deliver the 8-byte words whole) reports a wrong value - same collective fallback.  ('peer', 4): four ranks
    on the one GPU - rank indexing of the exchange (a lane per rank on the reading side) beyond a pair."""
    assert torch.cuda.is_available()
    amd = importlib.import_module('3dinfomax_amd')
    from helpers import close, grads_close
    path = str(tmp_path / 'dp.npz')
    mp.spawn(_worker, args=(_free_port(), path, native_sync, world), nprocs=world, join=True)
    z = np.load(path)
    mols = amd.synth.make_dataset(16, seed=21)
    pna, net = _models(amd)
    g2, g3 = _batch(amd, mols)
    loss = amd.NTXent(tau=0.1)(pna(g2), net(g3))
    loss.backward()
    assert abs(float(z['loss']) - loss.item()) < 1e-5 * abs(loss.item())
    for tag, m in (('pna', pna), ('net', net)):
        ref = {k: p.grad.cpu().numpy() for k, p in m.named_parameters()}
        grads_close({k: z[f'g/{tag}/{k}'] for k in ref}, ref, 2e-4, tag + ' ')
        for k, b in m.named_buffers():
            assert close(z[f'b/{tag}/{k}'], b.cpu(), 1e-5, 1e-6), k


def _accum_worker(rank, port, path):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    dist.init_process_group('gloo', rank=0, world_size=1)
    amd = importlib.import_module('3dinfomax_amd')
    adist = importlib.import_module('3dinfomax_amd.dist')
    mols = amd.synth.make_dataset(12, seed=5)
    pna, net = _models(amd)
    loss_fn = amd.NTXent(tau=0.1)
    adist.setup([pna, net], loss_fn, sync_bn=False)
    params = list(pna.parameters()) + list(net.parameters())
    red = adist.grad_reducer(params, modules=[pna, net])
    g2, g3 = _batch(amd, mols)

    def backward_once():
        loss_fn(pna(g2.local_copy()), net(g3.local_copy())).backward()

    out = {}
    backward_once()                                   # .grad empty: straight into the all-reduce buffer
    adist.allreduce_grads(params)
    out.update({f'one/{i}': p.grad.cpu().numpy().copy() for i, p in enumerate(params)})
    for p in params:
        p.grad = None
    backward_once()
    backward_once()                                   # gradient accumulation: .grad is the buffer's view already
    adist.allreduce_grads(params)
    out.update({f'two/{i}': p.grad.cpu().numpy().copy() for i, p in enumerate(params)})
    for p in params:                                  # zero_grad(set_to_none=False): zeros stay where they are
        p.grad.zero_()
    backward_once()
    adist.allreduce_grads(params)
    out.update({f'zeroed/{i}': p.grad.cpu().numpy().copy() for i, p in enumerate(params)})
    out['in_buffer'] = np.array([p.grad.data_ptr() == v.data_ptr() for p, v in zip(red.params, red.views)])
    np.savez(path, **out)
    dist.destroy_process_group()


def test_grad_reducer_accumulates_instead_of_clobbering(tmp_path):
    """with `.grad` already set the sink used to copy the new gradient over the accumulated one and autograd
    then added the buffer onto itself (2 g_new instead of g_old + g_new)."""
    path = str(tmp_path / 'acc.npz')
    mp.spawn(_accum_worker, args=(_free_port(), path), nprocs=1, join=True)
    z = np.load(path)
    n = sum(1 for k in z.files if k.startswith('one/'))
    assert n > 50 and bool(z['in_buffer'].all())
    for i in range(n):
        g = z[f'one/{i}']
        scale = max(float(np.abs(g).max()), 1e-30)
        # the BatchNorm running statistics differ between the passes, the batch statistics (train mode) do not: same bits
        assert np.abs(z[f'two/{i}'] - 2 * g).max() <= 1e-6 * scale, i
        assert np.abs(z[f'zeroed/{i}'] - g).max() <= 1e-6 * scale, i


def _uneven_worker(rank, port, path):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    dist.init_process_group('gloo', rank=rank, world_size=WORLD)
    amd = importlib.import_module('3dinfomax_amd')
    adist = importlib.import_module('3dinfomax_amd.dist')
    native = importlib.import_module('3dinfomax_amd.pna_native')
    mols = amd.synth.make_dataset(13, seed=33)                      # 13 molecules on 2 ranks: 7 + 6, nothing dropped
    plan = adist.shard_plan([m.n_atoms for m in mols], WORLD)
    pna, net = _models(amd)
    loss_fn = amd.NTXent(tau=0.1)
    adist.setup([pna, net], loss_fn, sync_bn=False)                 # per-rank BatchNorm statistics (DistributedDataParallel semantics)
    loss_fn.set_shard_counts([len(p) for p in plan])
    params = list(pna.parameters()) + list(net.parameters())
    red = adist.grad_reducer(params, modules=[pna, net])
    calls = []
    real = red.launch_async
    red.launch_async = lambda *a: (calls.append(sum(b - a_ for a_, b in red.early_spans) if not red._launched else 0), real(*a))[1]
    g2, g3 = _batch(amd, [mols[i] for i in plan[rank]])
    assert native.eligible(pna, g2)                                 # the whole-model C sequencer is the path that runs
    # step 1: the ranks agree on the early plan inside the first reduce() (no early collective before that); step 2 (the one
    # that is checked; same weights: no optimizer step in between) runs the split backward pass with the early all-reduce
    for _ in range(2):
        for p in params:
            p.grad = None
        share = loss_fn(pna(g2.local_copy()), net(g3.local_copy()))
        share.backward()
        adist.allreduce_grads(params)
    assert red._agreed is True
    total = adist.global_loss(share)
    if rank == 0:
        out = {'loss': total.item(), 'early_params': np.array(calls)}
        for tag, m in (('pna', pna), ('net', net)):
            for k, p in m.named_parameters():
                out[f'g/{tag}/{k}'] = p.grad.cpu().numpy()
        np.savez(path, **out)
    dist.destroy_process_group()


def test_two_rank_uneven_shards_local_bn_and_early_allreduce(tmp_path):
    """Atom-balanced shards of different sizes (dist.shard_plan: 7 + 6 molecules), per-rank BatchNorm statistics, the
    gradient all-reduce of the head and the upper layers started in the middle of the PNA backward pass
    (GradReducer.launch_async): loss and summed gradients equal the single-process emulation - both shards through the
    same weights, every shard's 2D rows scored against ALL 3D rows."""
    amd = importlib.import_module('3dinfomax_amd')
    adist = importlib.import_module('3dinfomax_amd.dist')
    losses = importlib.import_module('3dinfomax_amd.losses')
    from helpers import grads_close
    path = str(tmp_path / 'uneven.npz')
    mp.spawn(_uneven_worker, args=(_free_port(), path), nprocs=WORLD, join=True)
    z = np.load(path)
    # the early slices (head + upper half of the layers: > 10 k gradient elements) were started from inside the backward pass
    assert list(z['early_params']) and int(z['early_params'][0]) > 10000
    mols = amd.synth.make_dataset(13, seed=33)
    sizes = [m.n_atoms for m in mols]
    plan = adist.shard_plan(sizes, WORLD)
    assert sorted(i for p in plan for i in p) == list(range(13)) and [len(p) for p in plan] == [7, 6]
    atoms = [sum(sizes[i] for i in p) for p in plan]
    assert abs(atoms[0] - atoms[1]) <= max(sizes)                              # balanced by atoms, not only by count
    pna, net = _models(amd)
    z2, z3 = [], []
    for p in plan:
        g2, g3 = _batch(amd, [mols[i] for i in p])
        z2.append(pna(g2))
        z3.append(net(g3))
    z3_full = torch.cat(z3)
    total, off = 0, 0
    for r, p in enumerate(plan):
        total = total + losses.NTXentFn.apply(z2[r], z3_full, 0.1, 1e-8, 1, off, 13)
        off += len(p)
    total.backward()
    assert abs(float(z['loss']) - total.item()) < 1e-5 * abs(total.item())
    for tag, m in (('pna', pna), ('net', net)):
        ref = {k: p.grad.cpu().numpy() for k, p in m.named_parameters()}
        grads_close({k: z[f'g/{tag}/{k}'] for k in ref}, ref, 2e-4, tag + ' ')


def _rccl_worker(rank, port, path, provider='rccl'):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), I3D_SYNC_PROVIDER=provider)
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    torch.cuda.set_device(0)
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda:0'))
    amd = importlib.import_module('3dinfomax_amd')
    adist = importlib.import_module('3dinfomax_amd.dist')
    native = importlib.import_module('3dinfomax_amd.pna_native')
    calls = []
    real_run = native.run
    native.run = lambda *a, **k: (calls.append(1), real_run(*a, **k))[1]
    mols = amd.synth.make_dataset(16, seed=21)
    pna, net = _models(amd)
    loss_fn = amd.NTXent(tau=0.1)
    adist.setup([pna, net], loss_fn, sync_bn=True)
    ok = (adist.native_sync_active() and adist.native_sync_provider() == provider
          and importlib.import_module('3dinfomax_amd._lib').load().i3d_collectives_world() == 1)
    g2, g3 = _batch(amd, mols)
    params = list(pna.parameters()) + list(net.parameters())
    adist.grad_reducer(params, modules=[pna, net])
    share = loss_fn(pna(g2), net(g3))
    share.backward()
    adist.allreduce_grads(params)
    torch.cuda.synchronize()
    out = {'loss': share.item(), 'ok': ok and len(calls) == 1}
    for tag, m in (('pna', pna), ('net', net)):
        for k, p in m.named_parameters():
            out[f'g/{tag}/{k}'] = p.grad.cpu().numpy()
        for k, b in m.named_buffers():
            out[f'b/{tag}/{k}'] = b.cpu().numpy()
    np.savez(path, **out)
    adist.disable_native_sync()
    dist.destroy_process_group()


@pytest.mark.parametrize('provider', ['rccl', 'peer'])
def test_rccl_and_peer_providers_of_the_native_collectives_on_one_rank(tmp_path, provider):
    """The RCCL side of csrc/comm.hip (run-time binding, communicator of the library's own from a broadcast id,
    ncclAllGather / ncclAllReduce enqueued on the sequencer's stream) and the peer-write exchange (csrc/peer.hip) under an
    "nccl" process group, on the one GPU a test box has: a world of one rank runs every collective of the synchronised
    step and must reproduce the plain step."""
    amd = importlib.import_module('3dinfomax_amd')
    from helpers import close, grads_close
    path = str(tmp_path / 'rccl1.npz')
    mp.spawn(_rccl_worker, args=(_free_port(), path, provider), nprocs=1, join=True)
    z = np.load(path)
    assert bool(z['ok'])
    mols = amd.synth.make_dataset(16, seed=21)
    pna, net = _models(amd)
    g2, g3 = _batch(amd, mols)
    loss = amd.NTXent(tau=0.1)(pna(g2), net(g3))
    loss.backward()
    assert abs(float(z['loss']) - loss.item()) < 1e-5 * abs(loss.item())
    for tag, m in (('pna', pna), ('net', net)):
        ref = {k: p.grad.cpu().numpy() for k, p in m.named_parameters()}
        grads_close({k: z[f'g/{tag}/{k}'] for k in ref}, ref, 2e-4, tag + ' ')
        for k, b in m.named_buffers():
            assert close(z[f'b/{tag}/{k}'], b.cpu(), 1e-5, 1e-6), k


def _lost_rank_worker(rank, port, out):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), I3D_SYNC_PROVIDER='peer', I3D_PEER_TIMEOUT_S='1.5')
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    dist.init_process_group('gloo', rank=rank, world_size=WORLD)
    amd = importlib.import_module('3dinfomax_amd')
    adist = importlib.import_module('3dinfomax_amd.dist')
    L = importlib.import_module('3dinfomax_amd._lib').load()
    ops = importlib.import_module('3dinfomax_amd.ops')
    assert adist.enable_native_sync(dist.group.WORLD, torch.device('cuda:0'))
    x = torch.randn(64, 24, device='cuda:0') + rank
    dist.barrier()
    res = None
    if rank == 0:       # rank 1 never issues its BatchNorm: rank 0's kernel gives up after the timeout, the NEXT call reports it
        import time
        t0 = time.perf_counter()
        ops.act_stats_fwd(x, 'relu', 1e-5, 0.1)
        torch.cuda.synchronize()
        waited = time.perf_counter() - t0
        status = L.i3d_peer_status(adist._native_sync['peers'][0])
        try:
            ops.act_stats_fwd(x, 'relu', 1e-5, 0.1)
            raised = False
        except Exception as exc:      # noqa: BLE001
            raised = 'timed out' in str(exc)
        res = (1.0 < waited < 10.0) and status != 0 and raised
    dist.barrier()
    if rank == 0:
        out[0] = bool(res)
    else:
        out[1] = True
    # (no disable_native_sync: a rank is marked dead; the mailboxes go with the process)
    dist.destroy_process_group()


def test_peer_exchange_with_a_lost_rank_times_out_and_reports():
    """A rank that never issues its collective must cost the others a bounded wait and a clean error from the next call - not
    a hung GPU (csrc/peer.h: bounded spin, host-visible status word)."""
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_lost_rank_worker, args=(_free_port(), out), nprocs=WORLD, join=True)
    assert dict(out) == {0: True, 1: True}


BOTH_LIVE_TOL = 5e-2      # absolute, on losses between 3.0 and 0.08 (measured 1.4e-2)


def _both_live_worker(rank, port, path, provider):
    """world 1 under an "nccl" group, the split gradient reduction forced on (I3D_TEST_FORCE_EARLY_ALLREDUCE): 50 optimisation
    steps with the provider's BatchNorm collectives and torch's own communicator both in use"""
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), I3D_SYNC_PROVIDER=provider, I3D_TESTING='1', I3D_TEST_FORCE_EARLY_ALLREDUCE='1')
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    torch.cuda.set_device(0)
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda:0'))
    amd = importlib.import_module('3dinfomax_amd')
    adist = importlib.import_module('3dinfomax_amd.dist')
    mols = amd.synth.make_dataset(16, seed=21)
    pna, net = _models(amd)
    loss_fn = amd.NTXent(tau=0.1)
    adist.setup([pna, net], loss_fn, sync_bn=True)
    assert adist.native_sync_provider() == provider
    params = list(pna.parameters()) + list(net.parameters())
    red = adist.grad_reducer(params, modules=[pna, net])
    optim = amd.Adam(params, lr=1e-4)
    started_in_backward, started, real = [], [], red.launch_async

    def spy(module=None):
        before = len(red._launched)
        real(module)
        if len(red._launched) > before:
            (started_in_backward if module is not None else started).append(len(red._pending))
    red.launch_async = spy
    g2, g3 = _batch(amd, mols)
    losses = []
    for _ in range(50):
        loss = loss_fn(pna(g2.local_copy()), net(g3.local_copy()))
        loss.backward()
        adist.allreduce_grads(params)
        optim.step()
        optim.zero_grad()
        losses.append(loss.item())
    torch.cuda.synchronize()
    out = {'losses': np.array(losses), 'in_backward': len(started_in_backward), 'in_reduce': len(started),
           'async_works': max(started_in_backward + started + [0])}
    for tag, m in (('pna', pna), ('net', net)):
        for k, p in m.named_parameters():
            out[f'p/{tag}/{k}'] = p.detach().cpu().numpy()
    np.savez(path, **out)
    adist.disable_native_sync()
    dist.destroy_process_group()


@pytest.mark.parametrize('provider', ['rccl', 'peer'])
def test_provider_collectives_and_the_split_gradient_reduction_both_live(tmp_path, provider):
    """VERDICT round 4, next #8a.  50 optimisation steps at world 1 under an "nccl" process group with the split gradient
    all-reduce forced on (it needs an agreement between > 1 ranks otherwise).  peer: the early slices are started from inside
    the backward pass on torch's communicator (async work objects live) while the BatchNorm vectors travel through the
    mailboxes; rccl: the library's own communicator carries the BatchNorm collectives, the early START is switched off (two
    communicators must not be in flight at once) and reduce() issues those slices itself.  Both must reproduce the 50 steps of
    the plain single-process training (a world of one: every collective is the identity)."""
    amd = importlib.import_module('3dinfomax_amd')
    path = str(tmp_path / 'live.npz')
    mp.spawn(_both_live_worker, args=(_free_port(), path, provider), nprocs=1, join=True)
    z = np.load(path)
    # (step 1 runs the unsplit backward pass: the plan is agreed on in the first reduce(), which then starts the slices itself)
    if provider == 'rccl':
        assert int(z['in_backward']) == 0 and int(z['in_reduce']) == 50
    else:
        assert int(z['in_backward']) == 49 and int(z['in_reduce']) == 1
    assert int(z['async_works']) >= 1          # torch's communicator was used asynchronously (work objects pending)
    mols = amd.synth.make_dataset(16, seed=21)
    pna, net = _models(amd)
    params = list(pna.parameters()) + list(net.parameters())
    optim = amd.Adam(params, lr=1e-4)
    g2, g3 = _batch(amd, mols)
    loss_fn = amd.NTXent(tau=0.1)
    losses = []
    for _ in range(50):
        loss = loss_fn(pna(g2.local_copy()), net(g3.local_copy()))
        loss.backward()
        optim.step()
        optim.zero_grad()
        losses.append(loss.item())
    # (the synchronised path merges the statistics of `world` ranks in fp64 where the plain path finalises its own: same values up
    # to the last bits, 50 Adam steps apart)
    rel = np.abs(z['losses'] - np.array(losses)) / np.abs(np.array(losses))
    print(f'{provider}: 50 steps, loss {losses[0]:.5f} -> {losses[-1]:.5f}; relative difference to the plain training: first 5 steps '
          f'{rel[:5].max():.2e}, all {rel.max():.2e}')
    assert np.isfinite(z['losses']).all()
    # (the 16 molecules are overfitted within the 50 steps - the loss falls from 3.0 to below 0.1 - and rounding-level differences of the
    # first steps grow with every Adam step: relative on the first steps, absolute on the whole trajectory)
    assert rel[:5].max() < 1e-3 and float(np.abs(z['losses'] - np.array(losses)).max()) < BOTH_LIVE_TOL


def _soak_worker(rank, port, path, world, steps, seq0):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), I3D_SYNC_PROVIDER='peer', I3D_PEER_TIMEOUT_S='30',
                      I3D_TESTING='1', I3D_PEER_SEQ0=str(seq0))
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    amd = importlib.import_module('3dinfomax_amd')
    adist = importlib.import_module('3dinfomax_amd.dist')
    L = importlib.import_module('3dinfomax_amd._lib').load()
    pool = [amd.synth.make_dataset(4 * world, seed=300 + i) for i in range(3)]      # three resident global batches of different shapes
    pna, net = _models(amd)
    loss_fn = amd.NTXent(tau=0.1)
    adist.setup([pna, net], loss_fn, sync_bn=True)
    assert adist.native_sync_provider() == 'peer'
    ctxs = adist._native_sync['peers']
    first_seq = [int(L.i3d_peer_sequence(c)) for c in ctxs]
    params = list(pna.parameters()) + list(net.parameters())
    adist.grad_reducer(params, modules=[pna, net])
    optim = amd.Adam(params, lr=2e-4)
    shards = [_batch(amd, adist.shard_molecules(m, rank, world)) for m in pool]
    probe_step, probe = steps - 7, None
    for it in range(steps):
        g2, g3 = shards[it % 3]
        if it == probe_step and rank == 0:          # the weights this step starts from (every rank holds the same)
            probe = {f'w/{tag}/{k}': v.detach().cpu().numpy().copy() for tag, m in (('pna', pna), ('net', net))
                     for k, v in m.state_dict().items()}
        share = loss_fn(pna(g2.local_copy()), net(g3.local_copy()))
        share.backward()
        adist.allreduce_grads(params)
        if it == probe_step:
            total = adist.global_loss(share)
            if rank == 0:
                probe['loss'] = total.item()
                for tag, m in (('pna', pna), ('net', net)):
                    for k, p in m.named_parameters():
                        probe[f'g/{tag}/{k}'] = p.grad.cpu().numpy().copy()
        optim.step()
        optim.zero_grad()
    torch.cuda.synchronize()
    last_seq = [int(L.i3d_peer_sequence(c)) for c in ctxs]
    status = [int(L.i3d_peer_status(c)) for c in ctxs]
    # every rank ends with the same weights, bit for bit (rank-order sums in the exchange, the same all-reduced gradients)
    flat = torch.cat([p.detach().flatten() for p in params]).cpu()
    gathered = [torch.empty_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat)
    same = all(torch.equal(gathered[0], t) for t in gathered)
    if rank == 0:
        np.savez(path, first_seq=np.array(first_seq, dtype=np.float64), last_seq=np.array(last_seq, dtype=np.float64),
                 status=np.array(status), same=np.array(same), probe_batch=np.array(probe_step % 3), **probe)
    adist.disable_native_sync()
    dist.destroy_process_group()


def test_peer_exchange_soak_across_the_wrap_of_the_32_bit_tag(tmp_path):
    """VERDICT round 4, next #8b / ADVICE (peer.h tag).  Four ranks on the one GPU, 2000 optimisation steps (~28 k exchanges
    on the 2D network's context, ~12 k on the 3D network's: 14 + 6 per step of this small model), the sequence started 5 000 below 2^32 so that BOTH contexts cross
    the wrap of the tag early (tag 0 is skipped: a never-written mailbox word must not validate): no wait times out, every
    rank ends with bit-identical weights, and a step near the END of the run - slots reused tens of thousands of times, tags on
    the far side of the wrap - still equals the single-process full-batch step from the same weights."""
    amd = importlib.import_module('3dinfomax_amd')
    from helpers import grads_close
    world, steps, seq0 = 4, 2000, (1 << 32) - 5000
    path = str(tmp_path / 'soak.npz')
    mp.spawn(_soak_worker, args=(_free_port(), path, world, steps, seq0), nprocs=world, join=True)
    z = np.load(path)
    assert list(z['status']) == [0, 0] and bool(z['same'])
    # (the pattern exchange at set-up has already used the first ~100 sequence numbers of both contexts)
    assert (z['first_seq'] >= seq0).all() and (z['first_seq'] < seq0 + 1000).all() and (z['last_seq'] > (1 << 32)).all(), (z['first_seq'], z['last_seq'])
    pna, net = _models(amd)
    for tag, m in (('pna', pna), ('net', net)):
        m.load_state_dict({k: torch.from_numpy(z[f'w/{tag}/{k}']) for k in m.state_dict()})
    pna.cuda().train(), net.cuda().train()
    mols = amd.synth.make_dataset(4 * world, seed=300 + int(z['probe_batch']))
    g2, g3 = _batch(amd, mols)
    loss = amd.NTXent(tau=0.1)(pna(g2), net(g3))
    loss.backward()
    assert abs(float(z['loss']) - loss.item()) < 1e-5 * abs(loss.item())
    for tag, m in (('pna', pna), ('net', net)):
        ref = {k: p.grad.cpu().numpy() for k, p in m.named_parameters()}
        # (weights after 2000 steps; four shards sum in another order than the full batch: measured 3e-4 of a tensor's maximum)
        grads_close({k: z[f'g/{tag}/{k}'] for k in ref}, ref, 1e-3, tag + ' ')
