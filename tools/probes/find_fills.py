"""which Python lines issue aten::fill_ / zero_ on [N,F]- and [E,F]-sized tensors during one pre-training step"""
import importlib, sys, traceback
import torch
sys.path.insert(0, '/root/repo')
amd = importlib.import_module('3dinfomax_amd')
sys.argv = ['bench.py']
import bench
dev = torch.device('cuda:0')
mols = amd.synth.make_dataset(512, seed=1)
g2 = amd.batch([amd.bond_graph(m) for m in mols]).to(dev)
g3 = amd.batch([amd.complete_graph(m) for m in mols]).to(dev)
pna = amd.PNA(avg_d=1.0, device=dev, **bench.PNA_KW).to(dev).train()
net = amd.Net3D(node_dim=0, edge_dim=1, avg_d=1.0, **bench.NET3D_KW).to(dev).train()
loss_fn = amd.NTXent(tau=0.1)
opt = amd.Adam(list(pna.parameters()) + list(net.parameters()), lr=1e-4, fused=True)
def step():
    a, b = g2.local_copy(), g3.local_copy()
    loss = loss_fn(pna(a), net(b), nodes_per_graph=a.batch_num_nodes())
    loss.backward(); opt.step(); opt.zero_grad()
for _ in range(3): step()
from torch.utils._python_dispatch import TorchDispatchMode
class M(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        n = str(func)
        if 'fill' in n or 'zero' in n or 'ones' in n or 'full' in n:
            shp = [tuple(a.shape) for a in args if isinstance(a, torch.Tensor)]
            print('DISPATCH', n, shp, args[0] if args and not isinstance(args[0], torch.Tensor) else '')
            print(''.join(traceback.format_stack(limit=8)[:-1]))
        return func(*args, **(kwargs or {}))
with M():
    step()
