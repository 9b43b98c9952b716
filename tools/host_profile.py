"""Where does the host time of a step go?  Wraps every ops.* wrapper with a wall-clock accumulator (works across the
autograd engine's thread) and times the forward / backward / optimizer phases of the enqueue (no device sync).

    python tools/host_profile.py [--depth 4]
"""
import argparse
import importlib
import os
import sys
import time
import types

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--depth', type=int, default=4)
    ap.add_argument('--steps', type=int, default=30)
    a = ap.parse_args()
    amd = importlib.import_module('3dinfomax_amd')
    ops = importlib.import_module('3dinfomax_amd.ops')
    acc = {}

    def wrap(name, fn):
        def w(*args, **kw):
            t = time.perf_counter()
            r = fn(*args, **kw)
            d = time.perf_counter() - t
            c = acc.setdefault(name, [0, 0.0])
            c[0] += 1
            c[1] += d
            return r
        return w
    for name, fn in list(vars(ops).items()):
        if isinstance(fn, types.FunctionType) and not name.startswith('_') and name not in ('agg_codes', 'scaler_codes'):
            setattr(ops, name, wrap(name, fn))

    dev = torch.device('cuda:0')
    mols = amd.synth.make_dataset(512, seed=1000)
    g2 = amd.batch([amd.bond_graph(m) for m in mols]).to(dev)
    g3 = amd.batch([amd.complete_graph(m) for m in mols]).to(dev)
    pna = amd.PNA(avg_d=1.0, device=dev, **dict(bench.PNA_KW, propagation_depth=a.depth)).to(dev).train()
    net = amd.Net3D(node_dim=0, edge_dim=1, avg_d=1.0, **bench.NET3D_KW).to(dev).train()
    loss_fn = amd.NTXent(tau=0.1)
    named = list(pna.named_parameters()) + list(net.named_parameters())
    optim = amd.Adam([{'params': [p for k, p in named if 'batch_norm' in k], 'weight_decay': 0},
                              {'params': [p for k, p in named if 'batch_norm' not in k]}], lr=8e-5)
    phases = {'fwd_pna': 0.0, 'fwd_net3d': 0.0, 'loss': 0.0, 'backward': 0.0, 'optim': 0.0, 'zero_grad': 0.0}

    def step(timed):
        t = [time.perf_counter()]
        a2, b3 = g2.local_copy(), g3.local_copy()
        z2 = pna(a2); t.append(time.perf_counter())
        z3 = net(b3); t.append(time.perf_counter())
        loss = loss_fn(z2, z3); t.append(time.perf_counter())
        loss.backward(); t.append(time.perf_counter())
        optim.step(); t.append(time.perf_counter())
        optim.zero_grad(); t.append(time.perf_counter())
        if timed:
            for k, d in zip(phases, [t[i + 1] - t[i] for i in range(6)]):
                phases[k] += d
    for _ in range(5):
        step(False)
    torch.cuda.synchronize()
    acc.clear()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step(True)
        torch.cuda.synchronize()        # keep the launch queue empty: pure host cost, no back-pressure
    total = time.perf_counter() - t0
    n = a.steps
    print(f'host+device wall per step (synchronised each step): {total / n * 1e3:.3f} ms')
    print('host enqueue per phase (ms/step): ' + ', '.join(f'{k} {v / n * 1e3:.3f}' for k, v in phases.items())
          + f'  | sum {sum(phases.values()) / n * 1e3:.3f}')
    tot_ops = sum(v[1] for v in acc.values())
    print(f'inside ops.* wrappers: {tot_ops / n * 1e3:.3f} ms/step over {sum(v[0] for v in acc.values()) / n:.0f} calls/step')
    for k, (c, s) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
        print(f'   {k:24s} {c / n:6.1f} calls/step  {s / c * 1e6:7.2f} us/call  {s / n * 1e3:7.3f} ms/step')


if __name__ == '__main__':
    main()
