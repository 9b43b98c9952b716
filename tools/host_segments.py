"""Where the host time of a step goes (perf_counter around the big pieces; no profiler overhead).
    python tools/host_segments.py [--steps 200]"""
import argparse
import collections
import importlib
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

ACC = collections.defaultdict(float)
CNT = collections.defaultdict(int)
ON = [False]
SERIES = []


def wrap(obj, name, label):
    fn = getattr(obj, name)

    def timed(*a, **k):
        if not ON[0]:
            return fn(*a, **k)
        t0 = time.perf_counter()
        try:
            return fn(*a, **k)
        finally:
            ACC[label] += time.perf_counter() - t0
            CNT[label] += 1
    setattr(obj, name, timed)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--workload', default='qm9', choices=['qm9', 'qmugs'])
    ap.add_argument('--series', action='store_true', help='print the per-step host times of the big pieces')
    a = ap.parse_args()
    amd = importlib.import_module('3dinfomax_amd')
    mods = {n: importlib.import_module('3dinfomax_amd.' + n) for n in
            ('pna', 'net3d', 'losses', 'tape', 'layer_native', 'net3d_native', 'optim', 'layers', 'ops', 'streams')}
    dev = torch.device('cuda:0')
    qmugs = a.workload == 'qmugs'
    mols = amd.synth.make_dataset(500 if qmugs else 512, seed=1000, kind='qmugs' if qmugs else 'qm9')
    g2 = amd.batch([amd.bond_graph(m) for m in mols]).to(dev)
    if qmugs:
        import numpy as np
        rng = np.random.default_rng(2000)
        g3 = amd.batch([amd.complete_graph(m, c) for m in mols for c in amd.synth.conformers(m, rng, 3)]).to(dev)
    else:
        g3 = amd.batch([amd.complete_graph(m) for m in mols]).to(dev)
    torch.manual_seed(123)
    pna = amd.PNA(avg_d=1.0, device=dev, **dict(bench.PNA_KW, propagation_depth=7 if qmugs else 4)).to(dev).train()
    net = amd.Net3D(node_dim=0, edge_dim=1, avg_d=1.0, **bench.NET3D_KW).to(dev).train()
    loss_fn = amd.NTXentMultiplePositives(tau=0.1) if qmugs else amd.NTXent(tau=0.1)
    named = list(pna.named_parameters()) + list(net.named_parameters())
    optim = amd.Adam([{'params': [p for k, p in named if 'batch_norm' in k], 'weight_decay': 0},
                      {'params': [p for k, p in named if 'batch_norm' not in k]}], lr=8e-5, fused=True)
    wrap(mods['pna'].PNA, 'forward', 'fwd: PNA.forward')
    wrap(mods['net3d'].Net3D, 'forward', 'fwd: Net3D.forward')
    wrap(mods['losses'].NTXent, 'forward', 'fwd: NTXent.forward')
    wrap(mods['layer_native'], 'forward', 'fwd:   PNA layers (layer_native.forward, 4x)')
    wrap(mods['layer_native'], 'backward', 'bwd:   PNA layers (layer_native.backward, 4x)')
    wrap(mods['net3d_native'], 'forward', 'fwd:   Net3D native forward')
    wrap(mods['net3d_native'], 'backward', 'bwd:   Net3D native backward')
    wrap(mods['tape'], '_model_backward', 'bwd: model backward (tape, both models)')
    wrap(mods['losses'].NTXentFn, 'backward', 'bwd: NTXentFn.backward')
    pn = importlib.import_module('3dinfomax_amd.pna_native')
    wrap(pn.PNAModelFn, 'backward', 'bwd: PNAModelFn.backward (whole-model C sequencer + Python around it)')
    wrap(pn.PNAModelFn, 'forward', 'fwd:  PNAModelFn.forward')
    L = importlib.import_module('3dinfomax_amd._lib').load()
    for nm in ('i3d_pna_model_fwd', 'i3d_pna_model_bwd_part', 'i3d_net3d_edge_fwd', 'i3d_net3d_edge_bwd', 'i3d_ntxent_loss_fwd', 'i3d_ntxent_loss_bwd', 'i3d_adam_step'):
        if hasattr(L, nm):
            wrap(L, nm, 'C call: ' + nm)
    wrap(mods['pna'].PNAGNN, 'forward', 'fwd:  PNAGNN.forward (embeddings + layers)')
    wrap(mods['layers'].MLP, 'forward', 'fwd:  MLP.forward (heads)')

    def step():
        t0 = time.perf_counter()
        x, y = g2.local_copy(), g3.local_copy()
        loss = loss_fn(pna(x), net(y), nodes_per_graph=x.batch_num_nodes())
        t1 = time.perf_counter()
        loss.backward()
        t2 = time.perf_counter()
        optim.step()
        t3 = time.perf_counter()
        optim.zero_grad()
        t4 = time.perf_counter()
        if ON[0]:
            for k, v in (('STEP fwd+loss', t1 - t0), ('STEP backward', t2 - t1), ('STEP adam', t3 - t2), ('STEP zero_grad', t4 - t3)):
                ACC[k] += v
                CNT[k] += 1
            SERIES.append((t1 - t0, t2 - t1, t3 - t2))

    for _ in range(30):
        step()
    torch.cuda.synchronize()
    ON[0] = True
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    host = time.perf_counter() - t0
    torch.cuda.synchronize()
    total = time.perf_counter() - t0
    print(f'{a.steps} steps: host enqueue {1e3 * host / a.steps:.3f} ms/step, with GPU drain {1e3 * total / a.steps:.3f} ms/step')
    for k in sorted(ACC):
        print(f'  {k:58s} {1e3 * ACC[k] / a.steps:7.3f} ms/step  ({CNT[k] / a.steps:.1f} calls)')
    if a.series:
        print('per step (ms): fwd+loss | backward | adam')
        for f, b, o in SERIES[:60]:
            print(f'  {1e3 * f:7.3f} {1e3 * b:7.3f} {1e3 * o:7.3f}')


if __name__ == '__main__':
    main()
