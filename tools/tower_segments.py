"""Where the host time of the tower variant's step goes (perf_counter around the pieces; runs in the threads that execute them)."""
import collections
import importlib
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
amd = importlib.import_module('3dinfomax_amd')
po = importlib.import_module('3dinfomax_amd.pna_original')
layers = importlib.import_module('3dinfomax_amd.layers')
tape = importlib.import_module('3dinfomax_amd.tape')
ACC, CNT, ON = collections.defaultdict(float), collections.defaultdict(int), [False]


def wrap(obj, name, label):
    fn = getattr(obj, name)

    def timed(*a, **k):
        if not ON[0]:
            return fn(*a, **k)
        t0 = time.perf_counter()
        try:
            return fn(*a, **k)
        finally:
            ACC[label] += time.perf_counter() - t0
            CNT[label] += 1
    is_static = isinstance(obj, type) and isinstance(obj.__dict__.get(name), staticmethod)
    setattr(obj, name, staticmethod(timed) if is_static else timed)


dev = torch.device('cuda:0')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
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
for cls, nm in ((po._TowerLayerFn, 'tower layer'), (layers.FCFn, 'FCFn'), (layers.ReadoutFn, 'ReadoutFn'), (layers.EmbeddingSumFn, 'EmbeddingSumFn'),
                (po._StackedModelFn, 'StackedModelFn')):
    wrap(cls, 'forward', f'fwd {nm}')
    wrap(cls, 'backward', f'bwd {nm}')
wrap(po._TowerStacks, 'pack', 'pack')
wrap(po._TowerStacks, 'unpack_grads', 'unpack_grads')
wrap(po._TowerStacks, 'unpack_stats', 'unpack_stats')
wrap(po, '_stacks_for', 'stacks_for (validity check)')
wrap(tape, '_param_list', 'param_list')
seg = collections.defaultdict(float)


def step():
    t0 = time.perf_counter()
    out = orig(g2.local_copy(), snorm)
    t1 = time.perf_counter()
    loss = l1(out, targets)
    t2 = time.perf_counter()
    loss.backward()
    t3 = time.perf_counter()
    optim.step()
    t4 = time.perf_counter()
    optim.zero_grad()
    t5 = time.perf_counter()
    if ON[0]:
        for k, v in (('model forward', t1 - t0), ('loss', t2 - t1), ('backward', t3 - t2), ('optim.step', t4 - t3), ('zero_grad', t5 - t4)):
            seg[k] += v


for _ in range(20):
    step()
torch.cuda.synchronize()
ON[0] = True
n = 200
t0 = time.perf_counter()
for _ in range(n):
    step()
host = time.perf_counter() - t0
torch.cuda.synchronize()
print(f'batch {B}: host {host / n * 1e3:.3f} ms/step')
for k, v in seg.items():
    print(f'  STEP {k:40s} {v / n * 1e3:7.3f} ms')
for k in sorted(ACC):
    print(f'  {k:45s} {ACC[k] / n * 1e3:7.3f} ms/step ({CNT[k] / n:.1f} calls)')
