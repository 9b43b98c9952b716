"""The tower variant's training step (PNAOriginal, configs/pna_original.yml shape) as a plain loop - the command behind the
rocprofv3 kernel traces of the variant (profiles/r04_tower_*).  python tools/tower_step.py [batch] [steps]"""
import importlib
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
amd = importlib.import_module('3dinfomax_amd')
dev = torch.device('cuda:0')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
STEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 60
mols = amd.synth.make_dataset(B, seed=6000)
g2 = amd.batch([amd.bond_graph(m) for m in mols]).to(dev)
snorm = torch.cat([torch.full((m.n_atoms, 1), float(m.n_atoms) ** -0.5) for m in mols]).to(dev)
targets = torch.randn(B, 1, device=dev)
torch.manual_seed(123)
orig = amd.PNAOriginal(target_dim=1, hidden_dim=90, last_layer_dim=90, mid_batch_norm=True, last_batch_norm=True, graph_norm=True,
                       readout_batchnorm=True, edge_hidden_dim=70, readout_hidden_dim=70, readout_layers=2, dropout=0.0,
                       in_feat_dropout=0.0, propagation_depth=4, towers=5, divide_input_first=False, divide_input_last=True,
                       aggregators=['mean', 'max', 'min', 'std'], scalers=['identity', 'amplification', 'attenuation'],
                       readout_aggregators=['mean', 'max', 'min', 'sum'], pretrans_layers=1, posttrans_layers=1, residual=True,
                       gru=False, avg_d=1.0, device=dev).to(dev).train()
optim = amd.Adam(list(orig.parameters()), lr=1e-4, fused=True)
l1 = torch.nn.L1Loss()


def step():
    l1(orig(g2.local_copy(), snorm), targets).backward()
    optim.step()
    optim.zero_grad()


for _ in range(20):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(STEPS):
    step()
host = time.perf_counter() - t0
torch.cuda.synchronize()
print(f'batch {B}: host {host / STEPS * 1e3:.3f} ms/step, with drain {(time.perf_counter() - t0) / STEPS * 1e3:.3f} ms/step')
