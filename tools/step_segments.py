"""Host time of the segments of the pre-training step (forward+loss, backward, gradient all-reduce, Adam, zero_grad),
single-process form vs data-parallel form (world size 1 over RCCL), both in ONE process so that box-to-box noise
cancels.   python tools/step_segments.py [--steps 300]"""
import argparse
import importlib
import os
import sys
import time

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def build(amd, dev):
    torch.manual_seed(123)
    pna = amd.PNA(avg_d=1.0, device=dev, **bench.PNA_KW).to(dev).train()
    net = amd.Net3D(node_dim=0, edge_dim=1, avg_d=1.0, **bench.NET3D_KW).to(dev).train()
    named = list(pna.named_parameters()) + list(net.named_parameters())
    optim = amd.Adam([{'params': [p for k, p in named if 'batch_norm' in k], 'weight_decay': 0},
                      {'params': [p for k, p in named if 'batch_norm' not in k]}], lr=8e-5, fused=True)
    return pna, net, amd.NTXent(tau=0.1), [p for _, p in named], optim


def run(label, g2, g3, pna, net, loss_fn, params, optim, reduce, steps):
    seg = [0.0] * 5
    pc = time.perf_counter

    def step(timed):
        t0 = pc()
        a, b = g2.local_copy(), g3.local_copy()
        loss = loss_fn(pna(a), net(b), nodes_per_graph=a.batch_num_nodes())
        t1 = pc()
        loss.backward()
        t2 = pc()
        if reduce is not None:
            reduce()
        t3 = pc()
        optim.step()
        t4 = pc()
        optim.zero_grad()
        t5 = pc()
        if timed:
            for i, d in enumerate((t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4)):
                seg[i] += d

    for _ in range(40):
        step(False)
    torch.cuda.synchronize()
    t0 = pc()
    for _ in range(steps):
        step(True)
    torch.cuda.synchronize()
    total = (pc() - t0) / steps * 1e3
    names = ('fwd+loss', 'backward', 'allreduce', 'adam', 'zero_grad')
    print(f'{label:34s} {total:6.3f} ms/step   ' + '  '.join(f'{n} {s / steps * 1e3:6.3f}' for n, s in zip(names, seg)), flush=True)


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=300)
    ap.add_argument('--prepare', action='store_true', help='with --pg-first: create the compute streams before the process group')
    ap.add_argument('--prepare-lib', action='store_true', help='with --pg-first: load the HIP library and launch one of its kernels first')
    ap.add_argument('--prepare-step', action='store_true', help='with --pg-first: run two whole steps first')
    ap.add_argument('--pg-first', action='store_true', help='initialise the process group before anything touches the GPU')
    a = ap.parse_args()
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29534')
    dev = torch.device('cuda:0')
    torch.cuda.set_device(dev)
    if a.prepare:
        st = importlib.import_module('3dinfomax_amd.streams')
        x = torch.zeros(16, device=dev)
        with torch.cuda.stream(st._side(dev)):
            y = x + 1
        torch.cuda.synchronize()
    if a.prepare_lib:
        ops = importlib.import_module('3dinfomax_amd.ops')
        x = torch.randn(256, 256, device=dev)
        ops.gemm(x, x)
        torch.cuda.synchronize()
    if a.prepare_step:
        amd0 = importlib.import_module('3dinfomax_amd')
        mols0 = amd0.synth.make_dataset(64, seed=1)
        h2, h3 = amd0.batch([amd0.bond_graph(m) for m in mols0]).to(dev), amd0.batch([amd0.complete_graph(m) for m in mols0]).to(dev)
        pna0, net0, loss0, params0, optim0 = build(amd0, dev)
        for _ in range(2):
            aa, bb = h2.local_copy(), h3.local_copy()
            loss0(pna0(aa), net0(bb)).backward()
            optim0.step()
            optim0.zero_grad()
        torch.cuda.synchronize()
        del pna0, net0, loss0, params0, optim0
    if a.pg_first:
        dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
    amd = importlib.import_module('3dinfomax_amd')
    adist = importlib.import_module('3dinfomax_amd.dist')
    mols = amd.synth.make_dataset(512, seed=1000)
    g2 = amd.batch([amd.bond_graph(m) for m in mols]).to(dev)
    g3 = amd.batch([amd.complete_graph(m) for m in mols]).to(dev)
    m = build(amd, dev)
    run('single, process group first' if a.pg_first else 'single, no process group', g2, g3, *m, None, a.steps)
    if not a.pg_first:
        dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
        run('single, process group initialised', g2, g3, *m, None, a.steps)
    m = build(amd, dev)
    pna, net, loss_fn, params, optim = m
    adist.setup([pna, net], loss_fn)
    adist.grad_reducer(params, modules=[pna, net])
    for rep in range(2):
        run('data parallel (world 1)', g2, g3, *m, lambda: adist.allreduce_grads(params), a.steps)
    m = build(amd, dev)
    run('single again', g2, g3, *m, None, a.steps)
    dist.destroy_process_group()
