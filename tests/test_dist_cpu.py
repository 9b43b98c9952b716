"""world_size-2 (and, for the sharded loss, world_size-4: chunk / offset arithmetic that a pair of ranks cannot get wrong)
`gloo` tests (CPU, no GPU needed) of the data-parallel plumbing in 3dinfomax_amd/dist.py and
losses._AllGatherRowsFn: the molecule-sharded formulation - every rank scores its local 2D rows against the
ALL-GATHERED 3D rows, backward reduce-scatters dz2, gradients are summed - must reproduce the single-process loss
and gradients of the reference formula (oracle/pna3d_oracle.ntxent).  The compute inside each rank is done with
the oracle's arithmetic here; the same wiring around the HIP kernels is checked on the GPU in test_gpu_dist.py."""
import importlib
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

WORLD = 2


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _share(z1_local, z2_full, pos_offset, global_batch, tau=0.1):
    """One rank's loss share with the reference arithmetic (commons/losses.py:143-155) on its local rows."""
    sim = z1_local @ z2_full.T
    sim = sim / (z1_local.norm(dim=1)[:, None] * z2_full.norm(dim=1)[None, :] + 1e-8)
    sim = torch.exp(sim / tau)
    idx = torch.arange(z1_local.shape[0]) + pos_offset
    pos = sim[torch.arange(z1_local.shape[0]), idx]
    return (-torch.log(pos / (sim.sum(1) - pos))).sum() / global_batch


def _worker(rank, port, out, WORLD=WORLD):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=WORLD)
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    adist = importlib.import_module('3dinfomax_amd.dist')
    losses = importlib.import_module('3dinfomax_amd.losses')
    from oracle import pna3d_oracle as O
    B, dim = 12, 16
    g = torch.Generator().manual_seed(0)
    z1 = torch.randn(B, dim, generator=g)
    z2 = torch.randn(B, dim, generator=g)
    W = torch.randn(dim, dim, generator=g)          # a shared "model" so parameter gradients can be checked
    per = B // WORLD
    # ---- single-process reference
    Wr = W.clone().requires_grad_(True)
    full = O.ntxent(z1 @ Wr, z2 @ Wr, tau=0.1)
    full.backward()
    # ---- sharded
    Ws = W.clone().requires_grad_(True)
    a = z1[rank * per:(rank + 1) * per] @ Ws
    b_local = z2[rank * per:(rank + 1) * per] @ Ws
    b_full = losses._AllGatherRowsFn.apply(b_local, dist.group.WORLD)         # C1
    share = _share(a, b_full, rank * per, B)
    share.backward()
    adist.allreduce_grads([Ws])                                               # C2 (persistent flat buffer)
    total = adist.global_loss(share)
    ok = (abs(total.item() - full.item()) < 1e-5 * abs(full.item())
          and torch.allclose(Ws.grad, Wr.grad, rtol=1e-4, atol=1e-6))
    # ---- helpers
    x = torch.full((3, 2), float(rank + 1))
    gathered = adist.all_gather_rows(x)
    ok = ok and gathered.shape == (3 * WORLD, 2) and all(gathered[3 * r:3 * r + 3].eq(r + 1).all() for r in range(WORLD))
    rs = adist.reduce_scatter_rows(torch.arange(8.).view(4, 2) * (rank + 1))
    per_rs = 4 // WORLD
    ok = ok and torch.equal(rs, (torch.arange(8.).view(4, 2) * (WORLD * (WORLD + 1) // 2))[rank * per_rs:(rank + 1) * per_rs])
    # sync-BN statistic algebra: fp64 [sum, sumsq, count] all-reduce == full-batch mean / biased var
    rows = torch.randn(10 + 7 * rank, 5, generator=torch.Generator().manual_seed(10 + rank)).double() + 3
    sums = torch.cat([rows.sum(0), (rows * rows).sum(0), torch.tensor([float(rows.shape[0])], dtype=torch.float64)])
    adist.all_reduce_sum(sums)
    allrows = torch.cat([torch.randn(10 + 7 * r, 5, generator=torch.Generator().manual_seed(10 + r)).double() + 3
                         for r in range(WORLD)])
    n = sums[-1]
    mean, var = sums[:5] / n, sums[5:10] / n - (sums[:5] / n) ** 2
    ok = ok and torch.allclose(mean, allrows.mean(0)) and torch.allclose(var, allrows.var(0, unbiased=False))
    mols = list(range(10))
    cnt = adist.shard_counts(10, WORLD)
    ok = ok and adist.shard_molecules(mols, rank, WORLD) == mols[sum(cnt[:rank]):sum(cnt[:rank + 1])]
    out[rank] = bool(ok)
    dist.destroy_process_group()


@pytest.mark.parametrize('world', [2, 4])
def test_sharded_ntxent_and_collectives_match_single_process(world):
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(_free_port(), out, world), nprocs=world, join=True)
    assert dict(out) == {r: True for r in range(world)}


def test_shard_plan_balances_atoms_and_drops_nothing():
    adist = importlib.import_module('3dinfomax_amd.dist')
    rng = __import__('numpy').random.default_rng(0)
    for n, world in ((512, 8), (13, 2), (500, 8), (7, 8), (64, 3)):
        sizes = rng.integers(3, 60, size=n).tolist()
        plan = adist.shard_plan(sizes, world)
        assert sorted(i for p in plan for i in p) == list(range(n))
        assert [len(p) for p in plan] == adist.shard_counts(n, world)
        atoms = [sum(sizes[i] for i in p) for p in plan]
        if n >= 4 * world:
            assert max(atoms) - min(atoms) <= max(sizes), (n, world, atoms)
        assert plan == adist.shard_plan(sizes, world)                      # deterministic: every rank computes the same plan
    mols = list(range(11))
    assert [adist.shard_molecules(mols, r, 3) for r in range(3)] == [[0, 1, 2, 3], [4, 5, 6, 7], [8, 9, 10]]


def _var_worker(rank, port, out, counts=(7, 4)):
    WORLD = len(counts)
    counts = list(counts)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=WORLD)
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    adist = importlib.import_module('3dinfomax_amd.dist')
    losses = importlib.import_module('3dinfomax_amd.losses')
    from oracle import pna3d_oracle as O
    dim = 8
    B = sum(counts)
    g = torch.Generator().manual_seed(1)
    z1, z2, W = torch.randn(B, dim, generator=g), torch.randn(B, dim, generator=g), torch.randn(dim, dim, generator=g)
    Wr = W.clone().requires_grad_(True)
    full = O.ntxent(z1 @ Wr, z2 @ Wr, tau=0.1)
    full.backward()
    lo = sum(counts[:rank])
    Ws = W.clone().requires_grad_(True)
    a, b_local = z1[lo:lo + counts[rank]] @ Ws, z2[lo:lo + counts[rank]] @ Ws
    b_full = losses._AllGatherRowsFn.apply(b_local, dist.group.WORLD, counts)      # shards of different sizes
    share = _share(a, b_full, lo, B)
    share.backward()
    adist.allreduce_grads([Ws])
    total = adist.global_loss(share)
    ok = (b_full.shape[0] == B and abs(total.item() - full.item()) < 1e-5 * abs(full.item())
          and torch.allclose(Ws.grad, Wr.grad, rtol=1e-4, atol=1e-6))
    out[rank] = bool(ok)
    dist.destroy_process_group()


@pytest.mark.parametrize('counts', [(7, 4), (5, 3, 6, 2)])
def test_sharded_ntxent_with_uneven_shards(counts):
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_var_worker, args=(_free_port(), out, counts), nprocs=len(counts), join=True)
    assert dict(out) == {r: True for r in range(len(counts))}


def _accum_worker(rank, port, out):
    """GradReducer with an early (overlapped) all-reduce and a SECOND backward pass before reduce() (gradient
    accumulation): the early slice must end as sum over ranks of (first + second), not reduced(first) + local(second)."""
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=WORLD)
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    adist = importlib.import_module('3dinfomax_amd.dist')
    g = torch.Generator().manual_seed(5)
    params = [torch.nn.Parameter(torch.randn(*s, generator=g)) for s in ((4, 3), (5,), (2, 6), (7,))]
    grads = [[torch.randn(WORLD, 2, *p.shape, generator=g)] for p in params]          # [rank][pass]
    red = adist.GradReducer(params)
    owner = torch.nn.Linear(1, 1)                       # stands for the module whose backward starts the early slices
    other = torch.nn.Linear(1, 1)
    red.early_spans = [list(red.span_of[id(params[2])]), list(red.span_of[id(params[3])])]
    red.early_spans = [[red.early_spans[0][0], red.early_spans[1][1]]]
    red.early_module = id(owner)
    red._agreed = True
    # first backward pass: the gradients land in the flat buffer, the early slice is started - but not by another module
    first = red._sink(params, [gr[0][rank, 0].clone() for gr in grads])
    for p, v in zip(params, first):
        p.grad = v
    red.launch_async(other)
    ok = red._launched == []
    red.launch_async(owner)
    ok = ok and red._launched == [tuple(red.early_spans[0])]
    # second backward pass before reduce(): what the sink hands back is what autograd's AccumulateGrad would add to p.grad
    second = red._sink(params, [gr[0][rank, 1].clone() for gr in grads])
    for p, v in zip(params, second):
        if v is not None:
            p.grad.add_(v)
    ok = ok and second[2] is None and second[3] is None and second[0] is not None
    red.reduce()
    for p, gr in zip(params, grads):
        want = gr[0].sum(dim=(0, 1))
        ok = ok and torch.allclose(p.grad, want, rtol=1e-5, atol=1e-6)
    # the next step starts clean
    ok = ok and red._launched == [] and not red._late_used and float(red._late.abs().sum()) == 0.0
    out[rank] = bool(ok)
    dist.destroy_process_group()


def test_early_all_reduce_with_gradient_accumulation():
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_accum_worker, args=(_free_port(), out), nprocs=WORLD, join=True)
    assert dict(out) == {0: True, 1: True}


def _rccl_switch_worker(rank, port, out):
    """The RCCL provider of the synchronised BatchNorm switches the EARLY start of the gradient all-reduce off (two communicators
    must not be in flight at once) - the slices it would have started early must still be reduced by reduce()."""
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=WORLD)
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    adist = importlib.import_module('3dinfomax_amd.dist')
    adist.native_sync_provider = lambda: 'rccl'          # (the provider itself needs one GPU per rank: its decision is what is tested)
    g = torch.Generator().manual_seed(7)
    params = [torch.nn.Parameter(torch.randn(*s, generator=g)) for s in ((4, 3), (5,), (2, 6), (7,))]
    grads = [torch.randn(WORLD, *p.shape, generator=g) for p in params]
    red = adist.GradReducer(params)
    owner = torch.nn.Linear(1, 1)
    red.early_spans = [[red.span_of[id(params[2])][0], red.span_of[id(params[3])][1]]]
    red.early_module = id(owner)
    red._agreed = True
    for p, v in zip(params, red._sink(params, [gr[rank].clone() for gr in grads])):
        p.grad = v
    red.launch_async(owner)                                 # from inside the backward pass: switched off under RCCL
    ok = red._launched == []
    red.reduce()
    for p, gr in zip(params, grads):                        # EVERY slice reduced, the early ones included
        ok = ok and torch.allclose(p.grad, gr.sum(0), rtol=1e-5, atol=1e-6)
    out[rank] = bool(ok)
    dist.destroy_process_group()


def test_rccl_provider_switches_the_early_start_off_but_every_gradient_is_reduced():
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_rccl_switch_worker, args=(_free_port(), out), nprocs=WORLD, join=True)
    assert dict(out) == {0: True, 1: True}


def _equal_check_worker(rank, port, out):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=WORLD)
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    losses = importlib.import_module('3dinfomax_amd.losses')
    loss = losses.NTXent(tau=0.1).attach_group(dist.group.WORLD)
    issued = []
    real = dist.all_reduce

    def counting(t, *a, **k):
        issued.append(tuple(t.shape))
        return real(t, *a, **k)
    dist.all_reduce = counting
    ok = True
    for _ in range(3):                      # the steady state: every call issues the collective on every rank
        loss._check_equal_shards(torch.zeros(6, 4), dist)
    ok = ok and len(issued) == 3 and 6 in loss._equal_checked
    # the last partial batch: rank 0 still holds a row count it has verified, rank 1 a new one - BOTH must issue the same
    # collective (no hang, no mismatched collectives) and both must get the ValueError
    try:
        loss._check_equal_shards(torch.zeros(6 if rank == 0 else 5, 4), dist)
        ok = False
    except ValueError as exc:
        ok = ok and 'different numbers of molecules' in str(exc)
    ok = ok and len(issued) == 4
    dist.all_reduce = real
    out[rank] = bool(ok)
    dist.destroy_process_group()


def test_equal_shard_check_issues_the_same_collective_on_every_rank():
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_equal_check_worker, args=(_free_port(), out), nprocs=WORLD, join=True)
    assert dict(out) == {0: True, 1: True}
