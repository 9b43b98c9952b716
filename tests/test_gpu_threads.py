"""-m gpu: several trainings in threads of ONE process - the reference's `multithreaded_seeds` (train.py:647-655: one
`train(args)` per seed, each in a threading.Thread of the launching process; SURVEY.md 5.2 / 8(b) "Ownership / threading").

Each thread owns its models, its optimizer, its batches and its HIP stream; what the threads share is the library (the
weight-gradient stream table behind a mutex, per-(device, stream) scratch, thread-local error strings and per-call flags),
the caching allocator and autograd's ONE worker thread per device, which enqueues every thread's backward pass.  The
trainings must not see each other: the threaded run ends in the bits of the same trainings run one after the other.
What IS process-wide (documented in INTEGRATION.md): the matmul precision (`ops.set_matmul_precision`) - autograd's worker
thread runs the backward kernels of every training, a per-thread switch could not reach them - and the collective table
of synchronised BatchNorm (`i3d_set_collectives*`: one data-parallel training per process)."""
import importlib
import threading

import pytest
import torch

from helpers import NET3D_YML, PNA_YML, synth

pytestmark = pytest.mark.gpu
STEPS = 5


def _build(amd, seed):
    torch.manual_seed(seed)          # (global RNG: the models are built before the threads start)
    pna = amd.PNA(avg_d=1.0, device='cuda:0', **dict(PNA_YML, propagation_depth=3, hidden_dim=64, readout_hidden_dim=64,
                                                      target_dim=32)).cuda().train()
    net = amd.Net3D(node_dim=0, edge_dim=1, avg_d=1.0, **dict(NET3D_YML, target_dim=32)).cuda().train()
    with torch.no_grad():            # O(1) pre-BatchNorm scale
        for m in (pna, net):
            for n, p in m.named_parameters():
                if n.endswith('linear.weight'):
                    p.mul_(p.shape[1] * 0.7)
    return pna, net


def _train(amd, models, mols, out, key, stream=None, barrier=None):
    pna, net = models
    try:
        ctx = torch.cuda.stream(stream) if stream is not None else torch.cuda.stream(torch.cuda.current_stream())
        with ctx:
            g2 = amd.batch([amd.bond_graph(m) for m in mols]).to('cuda:0')
            g3 = amd.batch([amd.complete_graph(m) for m in mols]).to('cuda:0')
            loss_fn = amd.NTXent(tau=0.1)
            params = list(pna.parameters()) + list(net.parameters())
            optim = amd.Adam(params, lr=1e-3)
            losses = []
            if barrier is not None:
                barrier.wait()       # both trainings really run at the same time
            for _ in range(STEPS):
                a, b = g2.local_copy(), g3.local_copy()
                loss = loss_fn(pna(a), net(b))
                loss.backward()
                optim.step()
                optim.zero_grad()
                losses.append(loss)
            torch.cuda.current_stream().synchronize()
            out[key] = ([x.item() for x in losses], [p.detach().clone() for p in params],
                        [b.detach().clone().float() for m in (pna, net) for b in m.buffers()])
    except BaseException as exc:      # noqa: BLE001 - reported by the test body
        out[key] = exc
        if barrier is not None:
            barrier.abort()


def test_two_trainings_in_threads_equal_the_sequential_trainings():
    assert torch.cuda.is_available()
    amd = importlib.import_module('3dinfomax_amd')
    data = {0: synth.make_dataset(96, seed=41), 1: synth.make_dataset(80, seed=42)}
    states = {}
    for k in (0, 1):
        pna, net = _build(amd, 100 + k)
        states[k] = ({n: v.clone() for n, v in pna.state_dict().items()}, {n: v.clone() for n, v in net.state_dict().items()})

    def fresh(k):
        pna, net = _build(amd, 100 + k)
        pna.load_state_dict(states[k][0])
        net.load_state_dict(states[k][1])
        return pna, net

    # one after the other, on the default stream
    seq = {}
    for k in (0, 1):
        _train(amd, fresh(k), data[k], seq, k)
        assert not isinstance(seq[k], BaseException), seq[k]
    # together: a thread and a stream each
    thr, models = {}, {k: fresh(k) for k in (0, 1)}
    torch.cuda.synchronize()
    bar = threading.Barrier(2)
    threads = [threading.Thread(target=_train, args=(amd, models[k], data[k], thr, k, torch.cuda.Stream(), bar)) for k in (0, 1)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(300)
        assert not t.is_alive(), 'a training thread hangs'
    torch.cuda.synchronize()
    for k in (0, 1):
        assert not isinstance(thr[k], BaseException), thr[k]
        assert thr[k][0] == seq[k][0], (k, thr[k][0], seq[k][0])                   # losses, bit for bit
        for x, y in zip(thr[k][1] + thr[k][2], seq[k][1] + seq[k][2]):
            assert torch.equal(x, y)


def test_matmul_precision_is_process_wide():
    """`ops.set_matmul_precision` set in one thread is what every thread (autograd's worker included) computes with."""
    ops = importlib.import_module('3dinfomax_amd.ops')
    seen = {}
    prev = ops.set_matmul_precision('bf16')
    try:
        t = threading.Thread(target=lambda: seen.setdefault('other', ops.get_matmul_precision()))
        t.start()
        t.join()
        assert seen['other'] == 'bf16' and ops.get_matmul_precision() == 'bf16'
    finally:
        ops.set_matmul_precision(prev)
    assert ops.get_matmul_precision() == prev
