"""cProfile of the host side of the pre-training step (which Python / C entry points the enqueue time goes to).
    python tools/host_cprofile.py [--steps 30] [--top 45]"""
import argparse
import cProfile
import importlib
import os
import pstats
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=30)
    ap.add_argument('--top', type=int, default=45)
    a = ap.parse_args()
    amd = importlib.import_module('3dinfomax_amd')
    dev = torch.device('cuda:0')
    mols = amd.synth.make_dataset(512, seed=1000)
    g2 = amd.batch([amd.bond_graph(m) for m in mols]).to(dev)
    g3 = amd.batch([amd.complete_graph(m) for m in mols]).to(dev)
    torch.manual_seed(123)
    pna = amd.PNA(avg_d=1.0, device=dev, **bench.PNA_KW).to(dev).train()
    net = amd.Net3D(node_dim=0, edge_dim=1, avg_d=1.0, **bench.NET3D_KW).to(dev).train()
    loss_fn = amd.NTXent(tau=0.1)
    named = list(pna.named_parameters()) + list(net.named_parameters())
    optim = amd.Adam([{'params': [p for k, p in named if 'batch_norm' in k], 'weight_decay': 0},
                              {'params': [p for k, p in named if 'batch_norm' not in k]}], lr=8e-5, fused=True)

    bw_prof = cProfile.Profile()

    class _ProfileBackwardThread(torch.autograd.Function):
        """first node of the backward pass: switches cProfile on in autograd's worker thread (cProfile is per thread)"""
        @staticmethod
        def forward(ctx, x):
            return x.view_as(x)

        @staticmethod
        def backward(ctx, g):
            if state['profile_backward']:
                bw_prof.enable()
            return g

    state = {'profile_backward': False}

    def step():
        a_, b_ = g2.local_copy(), g3.local_copy()
        loss = loss_fn(pna(a_), net(b_), nodes_per_graph=a_.batch_num_nodes())
        loss = _ProfileBackwardThread.apply(loss)
        loss.backward()
        optim.step()
        optim.zero_grad()

    for _ in range(5):
        step()
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    state['profile_backward'] = True
    pr.enable()
    for _ in range(a.steps):
        step()
    pr.disable()
    torch.cuda.synchronize()
    print(f'per-step figures: divide by {a.steps}')
    print('==== main thread (forward, loss, optimizer) ====')
    pstats.Stats(pr).sort_stats('tottime').print_stats(a.top)
    print('==== autograd worker thread (backward) ====')
    pstats.Stats(bw_prof).sort_stats('tottime').print_stats(a.top)


if __name__ == '__main__':
    main()
