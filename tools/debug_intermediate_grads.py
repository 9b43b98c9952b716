import importlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests'), os.path.join(ROOT, 'tests', 'golden')]
from helpers import PNA_YML, synth
from fill import det_fill
from oracle import pna3d_oracle as O
from test_gpu_models import _det_load, make_batch
amd = importlib.import_module('3dinfomax_amd')
layers = importlib.import_module('3dinfomax_amd.layers')
pna_mod = importlib.import_module('3dinfomax_amd.pna')

nmol, td, depth, seed = 128, 8, 2, 5
mols = synth.make_dataset(nmol, seed=seed)
kw = dict(PNA_YML, propagation_depth=depth, target_dim=td, batch_norm_momentum=0.1)
pna = amd.PNA(avg_d=1.0, device='cuda:0', **kw)
_det_load(pna, 'pnaF')
P = O.require_grad({k: v.clone() for k, v in pna.state_dict().items()})
cfg = O.pna_config(**kw)
og2, _ = O.graphs_from_molecules(mols)
target = torch.from_numpy(det_fill((nmol, td), 'homo_targets', 2.0))
# ---- oracle with captured intermediates
ref = {}
h = O.embedding_sum(og2['atom_feat'], P, 'node_gnn.atom_encoder.atom_embedding_list', 9)
ef = O.embedding_sum(og2['bond_feat'], P, 'node_gnn.bond_encoder.bond_embedding_list', 3)
for l in range(depth):
    cap = {}
    h = O.pna_layer(h, ef, og2['src'], og2['dst'], P, f'node_gnn.mp_layers.{l}', cfg, True, cap)
    for k in ('e', 'agg'):
        cap[k].retain_grad(); ref[f'{k}{l}'] = cap[k]
    h.retain_grad(); ref[f'h{l}'] = h
r = torch.cat([O.segment_readout(h, og2['batch_num_nodes'], op) for op in cfg['readout_aggregators']], dim=-1)
r.retain_grad(); ref['readout'] = r
rp = O.mlp(r, P, 'output', 2, 'relu', 'none', True, False, cfg['batch_norm_momentum'], True)
(rp * target).sum().backward()
# ---- GPU with captured intermediates
got = {}
orig_agg = layers.AggregateFn.apply
idx_holder = {}
pna.cuda().train()
g2, _ = make_batch(amd, mols)
idx = g2.index()
count = {'l': 0}
def layer_forward(self, g, ef_sorted=None):
    l = count['l']; count['l'] += 1
    hh = g.ndata['feat']
    e = self.pretrans.forward_edge(hh, ef_sorted, idx)
    e.retain_grad(); got[f'e{l}'] = e
    agg = layers.AggregateFn.apply(e, idx, self.aggregators, self.scalers, 1.0)
    agg.retain_grad(); got[f'agg{l}'] = agg
    hn = self.posttrans.forward_concat2(hh, agg, residual=hh if self.residual else None)
    hn.retain_grad(); got[f'h{l}'] = hn
    g.ndata['feat'] = hn
    return hn
pna_mod.PNALayer.forward = layer_forward
pna.node_gnn(g2)
ro = layers.ReadoutFn.apply(g2.ndata['feat'], idx, pna._readout_codes)
ro.retain_grad(); got['readout'] = ro
pred = pna.output(ro)
(pred * target.cuda()).sum().backward()
perm = idx.perm.long().cpu()
print('pred err', ((pred.cpu() - rp).abs().max() / rp.abs().max()).item())
for k in ['readout'] + [f'{n}{l}' for l in reversed(range(depth)) for n in ('h', 'agg', 'e')]:
    a, b = got[k].grad.cpu(), ref[k].grad
    av, bv = got[k].detach().cpu(), ref[k].detach()
    if k.startswith('e'):
        b, bv = b[perm], bv[perm]
    ge = ((a - b).abs().max() / b.abs().max()).item()
    ve = ((av - bv).abs().max() / bv.abs().max()).item()
    worst = (a - b).abs().argmax().item()
    print(f'{k:8s} value err {ve:.2e}   grad err {ge:.2e}   (worst at flat index {worst}, row {worst // a.shape[1]}, col {worst % a.shape[1]}: got {a.flatten()[worst]:.5e} ref {b.flatten()[worst]:.5e})')
gp = idx.graph_ptr.cpu().numpy()
for (row, col) in ((236, 91), (231, 91)):
    gi = int(np.searchsorted(gp, row, side='right') - 1)
    lo, hi = gp[gi], gp[gi + 1]
    a = got['h1'].detach().cpu()[lo:hi, col]; b = ref['h1'].detach()[lo:hi, col]
    print('graph', gi, 'rows', lo, hi, 'col', col)
    print('  gpu vals', [f'{x:.7f}' for x in a.tolist()])
    print('  ref vals', [f'{x:.7f}' for x in b.tolist()])
    print('  gpu grad', [f'{x:.4f}' for x in got['h1'].grad.cpu()[lo:hi, col].tolist()])
    print('  ref grad', [f'{x:.4f}' for x in ref['h1'].grad[lo:hi, col].tolist()])
    m = mols[gi]
    print('  atom feats rows', m.atom_feat[row - lo].tolist(), m.atom_feat[(231 if row == 236 else 236) - lo].tolist() if lo <= (231 if row == 236 else 236) < hi else None)
