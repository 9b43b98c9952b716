import importlib, os, sys, torch
sys.path.insert(0, '/root/repo')
import bench
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
def step():
    a, b = g2.local_copy(), g3.local_copy()
    loss = loss_fn(pna(a), net(b), nodes_per_graph=a.batch_num_nodes())
    loss.backward(); optim.step(); optim.zero_grad()
for _ in range(5): step()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU], with_stack=True) as prof:
    for _ in range(10): step()
    torch.cuda.synchronize()
ev = prof.key_averages(group_by_stack_n=4)
rows = [(e.key, e.count / 10, e.cpu_time_total / 10, [s for s in e.stack if 'repo' in s or 'bench' in s][:2]) for e in ev if e.key.startswith('aten::')]
rows.sort(key=lambda r: -r[2])
for k, c, t, st in rows[:45]:
    print(f'{k:38s} {c:6.1f}/step {t:8.1f} us/step  {st}')
