"""How long does the host need, in the FIRST process on a fresh box, before its per-step enqueue time settles?  Runs the headline
step 2000 times and prints the host enqueue time and the step time per block of 50 steps with the seconds since process start.
    python tools/host_settle_probe.py"""
import importlib
import os
import sys
import time

T0 = time.time()
import torch  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

amd = importlib.import_module('3dinfomax_amd')
dev = torch.device('cuda:0')
print(f'import done at {time.time() - T0:.1f} s', flush=True)
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
print(f'set-up done at {time.time() - T0:.1f} s', flush=True)


def step():
    loss = loss_fn(pna(g2.local_copy()), net(g3.local_copy()))
    loss.backward()
    optim.step()
    optim.zero_grad()


for blk in range(40):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        step()
    host = time.perf_counter() - t0
    torch.cuda.synchronize()
    tot = time.perf_counter() - t0
    print(f'block {blk:2d} at {time.time() - T0:6.1f} s: host {host * 20:.3f} ms/step, step {tot * 20:.3f} ms', flush=True)
