"""The 3D network alone at the configs[3] shape (QMugs-shaped molecules, 3 conformers: ~3.9 M complete-graph edges): forward +
backward back to back with nothing else on the device - run under `rocprofv3 --kernel-trace` for the stand-alone time of every
edge-stage kernel (csrc/net3d_edge.hip), or by itself for the per-direction wall time.

    python tools/n3_bench.py [--mols 500] [--conf 3] [--steps 20] [--dtype fp32|bf16] [--kind qmugs|qm9]
"""
import argparse
import importlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--mols', type=int, default=500)
    ap.add_argument('--conf', type=int, default=3)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--dtype', default='fp32')
    ap.add_argument('--kind', default='qmugs')
    args = ap.parse_args()
    amd = importlib.import_module('3dinfomax_amd')
    ops = importlib.import_module('3dinfomax_amd.ops')
    bench = importlib.import_module('bench')
    ops.set_matmul_precision(args.dtype)
    dev = torch.device('cuda:0')
    mols = amd.synth.make_dataset(args.mols, seed=1000, kind=args.kind)
    rng = np.random.default_rng(2000)
    if args.conf > 1:
        g3 = amd.batch([amd.complete_graph(m, c) for m in mols for c in amd.synth.conformers(m, rng, args.conf)]).to(dev)
    else:
        g3 = amd.batch([amd.complete_graph(m) for m in mols]).to(dev)
    torch.manual_seed(123)
    net = amd.Net3D(node_dim=0, edge_dim=1, avg_d=1.0, **bench.NET3D_KW).to(dev).train()
    E3, N = g3.number_of_edges(), g3.number_of_nodes()
    cot = torch.randn(args.mols * args.conf, 256, device=dev) * 0.01

    def step():
        z = net(g3.local_copy())
        z.backward(cot)
        for p in net.parameters():
            p.grad = None

    for _ in range(5):
        step()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(args.steps):
        step()
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / args.steps
    print(f'n3_bench: {args.kind} {args.mols} x {args.conf}, {N} atoms, {E3} complete-graph edges, {args.dtype}: '
          f'{ms:.3f} ms per forward + backward; one [E3,20] fp32 tensor = {E3 * 80 / 1e6:.1f} MB')


if __name__ == '__main__':
    main()
