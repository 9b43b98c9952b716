"""Parameter gradients of the 3D network: fused edge stage (csrc/net3d_edge.hip) and per-block path, each against a float64
torch model of the same network (reference models/net3d.py structure of the pre-training configs).  RED=sum|mean.
    python tools/probes/net3d_edge_vs_fp64.py
"""
import importlib, sys, math, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
amd = importlib.import_module('3dinfomax_amd')
native = importlib.import_module('3dinfomax_amd.net3d_native')
from helpers import NET3D_YML
import os
RED = os.environ.get('RED', 'mean')
NET3D_YML = dict(NET3D_YML, reduce_func=RED)
import torch.nn.functional as F
synth = amd.synth
mols = synth.make_dataset(40, seed=23)
def run(fused):
    native.FUSED_EDGE = fused
    torch.manual_seed(11)
    net = amd.Net3D(node_dim=0, edge_dim=1, avg_d=1.0, **NET3D_YML).cuda().train()
    g3 = amd.batch([amd.complete_graph(m) for m in mols]).to('cuda:0')
    d_raw = g3.edata['d'].clone()
    z = net(g3)
    (z * torch.linspace(-1, 1, z.shape[1], device='cuda:0')).sum().backward()
    return net, g3, d_raw, {n: p.grad.clone() for n, p in net.named_parameters()}
net, g3, d_raw, a = run(True)
_, _, _, b = run(False)
# torch float64 reference of the whole network
idx = g3.index()
P = {n: p.detach().double().requires_grad_(True) for n, p in net.named_parameters()}
d = d_raw.double().view(-1)
src, dst = g3.edges()
src, dst = src.long(), dst.long()
N = g3.number_of_nodes()
def bn(x, g, be):
    mu, var = x.mean(0), x.var(0, unbiased=False)
    return (x - mu) / torch.sqrt(var + 1e-5) * g + be
n_enc = 4
sc = 2.0 ** torch.arange(n_enc, device=d.device).double()
f = torch.cat([torch.sin(d[:, None] / sc), torch.cos(d[:, None] / sc), d[:, None]], 1)
pre = 'edge_input.fully_connected.0.'
e0 = F.silu(bn(F.silu(f @ P[pre + 'linear.weight'].T + P[pre + 'linear.bias']), P[pre + 'batch_norm.weight'], P[pre + 'batch_norm.bias']))
h = P['node_embedding'][None, :].expand(N, -1)
pm = 'mp_layers.0.message_network.fully_connected.0.'
m = bn(F.silu(torch.cat([h[src], h[dst], e0], 1) @ P[pm + 'linear.weight'].T + P[pm + 'linear.bias']), P[pm + 'batch_norm.weight'], P[pm + 'batch_norm.bias'])
w = torch.sigmoid(m @ P['mp_layers.0.soft_edge_network.weight'].T + P['mp_layers.0.soft_edge_network.bias'])
msg = m * w
deg = torch.zeros(N, device=d.device, dtype=torch.float64).index_add_(0, dst, torch.ones_like(d))
m_sum = torch.zeros(N, 20, device=d.device, dtype=torch.float64).index_add_(0, dst, msg) / (deg[:, None] if RED == 'mean' else 1.0)
pu = 'mp_layers.0.update_network.fully_connected.0.'
h1 = bn((m_sum + h) @ P[pu + 'linear.weight'].T + P[pu + 'linear.bias'], P[pu + 'batch_norm.weight'], P[pu + 'batch_norm.bias']) + h
gp = idx.graph_ptr.long().tolist()
ro = []
for op in ['min', 'max', 'mean']:
    rows = []
    for k in range(len(gp) - 1):
        seg = h1[gp[k]:gp[k + 1]]
        rows.append(seg.min(0)[0] if op == 'min' else seg.max(0)[0] if op == 'max' else seg.mean(0))
    ro.append(torch.stack(rows))
ro = torch.cat(ro, 1)
z = ro @ P['output.fully_connected.0.linear.weight'].T + P['output.fully_connected.0.linear.bias']
(z * torch.linspace(-1, 1, z.shape[1], device='cuda:0').double()).sum().backward()
for n in a:
    r = P[n].grad
    print(f'{n:62s} fused-ref {(a[n]-r).abs().max().item():.3e} block-ref {(b[n]-r).abs().max().item():.3e} scale {r.abs().max().item():.3e}')
