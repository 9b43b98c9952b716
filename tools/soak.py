"""Soak run: 2000 optimisation steps over six resident batches of different shapes; prints the loss curve and the
allocator's footprint (no growth = no leak through the tape / arenas / gradient pools).   python tools/soak.py"""
import importlib, sys, time, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
amd = importlib.import_module('3dinfomax_amd')
dev = torch.device('cuda:0')
torch.manual_seed(1)
pna = amd.PNA(avg_d=1.0, device=dev, **bench.PNA_KW).to(dev).train()
net = amd.Net3D(node_dim=0, edge_dim=1, avg_d=1.0, **bench.NET3D_KW).to(dev).train()
loss_fn = amd.NTXent(tau=0.1)
named = list(pna.named_parameters()) + list(net.named_parameters())
opt = amd.Adam([{'params': [p for k, p in named if 'batch_norm' in k], 'weight_decay': 0},
                {'params': [p for k, p in named if 'batch_norm' not in k]}], lr=8e-5, fused=True)
batches = []
for i in range(6):
    mols = amd.synth.make_dataset(512 if i % 2 == 0 else 300 + 37 * i, seed=50 + i)     # varying batch shapes
    batches.append((amd.batch([amd.bond_graph(m) for m in mols]).to(dev), amd.batch([amd.complete_graph(m) for m in mols]).to(dev)))
losses, mem = [], []
t0 = time.time()
for it in range(2000):
    g2, g3 = batches[it % 6]
    a, b = g2.local_copy(), g3.local_copy()
    loss = loss_fn(pna(a), net(b))
    loss.backward(); opt.step(); opt.zero_grad()
    if it % 240 == 0:
        torch.cuda.synchronize()
        losses.append(round(loss.item(), 4)); mem.append(torch.cuda.memory_allocated() >> 20)
print('losses', losses)
print('allocated MiB', mem, 'reserved MiB', torch.cuda.memory_reserved() >> 20, 'seconds', round(time.time() - t0, 1))
assert all(l == l and abs(l) < 1e4 for l in losses)
assert max(mem[2:]) - min(mem[2:]) < 64, mem      # sampled at the same batch of the cycle
