"""Probe: the WHOLE training step (forward, loss, backward, Adam) of the headline workload captured as one HIP graph with
torch.cuda.graph on a resident batch, replayed - does everything the step launches capture (C sequencers with their forked
streams and events, the 3D network's side stream, autograd's worker thread), what does a replay cost the host, and what is the
DEVICE time of a step without any host in the loop?  (Adam's bias corrections are launch arguments: a replay repeats the captured
step's - this probe measures, it does not train.)
    python tools/graph_step_probe.py [--dtype bf16] [--steps 200]"""
import argparse
import importlib
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--dtype', default='fp32')
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--depth', type=int, default=4)
    a = ap.parse_args()
    amd = importlib.import_module('3dinfomax_amd')
    ops = importlib.import_module('3dinfomax_amd.ops')
    ops.set_matmul_precision(a.dtype)
    dev = torch.device('cuda:0')
    mols = amd.synth.make_dataset(512, seed=1000)
    g2 = amd.batch([amd.bond_graph(m) for m in mols]).to(dev)
    g3 = amd.batch([amd.complete_graph(m) for m in mols]).to(dev)
    torch.manual_seed(123)
    pna = amd.PNA(avg_d=1.0, device=dev, **dict(bench.PNA_KW, propagation_depth=a.depth)).to(dev).train()
    net = amd.Net3D(node_dim=0, edge_dim=1, avg_d=1.0, **bench.NET3D_KW).to(dev).train()
    loss_fn = amd.NTXent(tau=0.1)
    named = list(pna.named_parameters()) + list(net.named_parameters())
    optim = amd.Adam([{'params': [p for k, p in named if 'batch_norm' in k], 'weight_decay': 0},
                      {'params': [p for k, p in named if 'batch_norm' not in k]}], lr=8e-5, fused=True)
    out = {}

    def step():
        x, y = g2.local_copy(), g3.local_copy()
        loss = loss_fn(pna(x), net(y))
        loss.backward()
        optim.step()
        optim.zero_grad()
        out['loss'] = loss

    def timed(fn, n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        host = time.perf_counter() - t0
        torch.cuda.synchronize()
        return host / n * 1e3, (time.perf_counter() - t0) / n * 1e3
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(20):
            step()
    torch.cuda.current_stream().wait_stream(s)
    print('eager: host %.3f ms, step %.3f ms' % timed(step, a.steps), flush=True)
    graph = torch.cuda.CUDAGraph()
    try:
        with torch.cuda.graph(graph):
            step()
    except Exception as exc:      # noqa: BLE001
        print('capture FAILED:', type(exc).__name__, str(exc)[:1500])
        return
    torch.cuda.synchronize()
    print('captured; loss of the captured step', float(out['loss']))
    graph.replay()
    torch.cuda.synchronize()
    print('replayed once; loss', float(out['loss']))
    print('graph replay: host %.3f ms, step %.3f ms' % timed(graph.replay, a.steps), flush=True)


if __name__ == '__main__':
    main()
