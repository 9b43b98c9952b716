"""Per-step losses of the pre-training step on one resident batch: the HIP path (whatever I3D_* switches are set) and,
with --oracle N, the CPU oracle for the first N steps from the same initial weights.
    python tools/loss_trajectory.py [--batch 128] [--steps 40] [--oracle 6] [--depth 4]"""
import argparse
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=128)
    ap.add_argument('--steps', type=int, default=40)
    ap.add_argument('--oracle', type=int, default=0)
    ap.add_argument('--depth', type=int, default=4)
    a = ap.parse_args()
    amd = importlib.import_module('3dinfomax_amd')
    dev = torch.device('cuda:0')
    mols = amd.synth.make_dataset(a.batch, seed=1000)
    kw2 = dict(bench.PNA_KW, propagation_depth=a.depth)
    torch.manual_seed(123)
    pna = amd.PNA(avg_d=1.0, device='cpu', **kw2)
    net = amd.Net3D(node_dim=0, edge_dim=1, avg_d=1.0, **bench.NET3D_KW)
    sd2 = {k: v.clone() for k, v in pna.state_dict().items()}
    sd3 = {k: v.clone() for k, v in net.state_dict().items()}
    ref = []
    if a.oracle:
        from oracle import pna3d_oracle as O
        cfg2, cfg3 = O.pna_config(**kw2), O.net3d_config(**bench.NET3D_KW)
        P2, P3 = O.require_grad({k: v.clone() for k, v in sd2.items()}), O.require_grad({k: v.clone() for k, v in sd3.items()})
        named = [(k, P2[k]) for k in O.trainable(P2)] + [(k, P3[k]) for k in O.trainable(P3)]
        opt = torch.optim.Adam([{'params': [p for k, p in named if 'batch_norm' in k], 'weight_decay': 0},
                                {'params': [p for k, p in named if 'batch_norm' not in k]}], lr=8e-5)
        g2, g3 = O.graphs_from_molecules(mols)
        ref = [O.train_step(g2, g3, P2, cfg2, P3, cfg3, opt, 0.1).item() for _ in range(a.oracle)]
    pna.to(dev).train(), net.to(dev).train()
    g2 = amd.batch([amd.bond_graph(m) for m in mols]).to(dev)
    g3 = amd.batch([amd.complete_graph(m) for m in mols]).to(dev)
    loss_fn = amd.NTXent(tau=0.1)
    named = list(pna.named_parameters()) + list(net.named_parameters())
    optim = amd.Adam([{'params': [p for k, p in named if 'batch_norm' in k], 'weight_decay': 0},
                      {'params': [p for k, p in named if 'batch_norm' not in k]}], lr=8e-5)
    losses = []
    for _ in range(a.steps):
        x, y = g2.local_copy(), g3.local_copy()
        loss = loss_fn(pna(x), net(y), nodes_per_graph=x.batch_num_nodes())
        loss.backward()
        optim.step()
        optim.zero_grad()
        losses.append(loss.item())
    print('switches:', {k: v for k, v in os.environ.items() if k.startswith('I3D_')})
    for i, v in enumerate(losses):
        r = f'  oracle {ref[i]:.6f}  rel {abs(v - ref[i]) / abs(ref[i]):.2e}' if i < len(ref) else ''
        print(f'step {i:3d}  loss {v:.6f}{r}')


if __name__ == '__main__':
    main()
