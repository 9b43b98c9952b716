"""Fixture for PNA(pairwise_distances=True) (reference models/pna.py:105, 239-249): the unmodified reference PNA on a batch whose
graphs carry atom coordinates in ndata['x'], forward + backward -> tests/golden/pna_pairwise.npz.

    python tests/golden/gen_golden_pairwise.py          (build container only: imports /root/reference)
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as G  # noqa: E402

CFG = dict(target_dim=8, hidden_dim=16, mid_batch_norm=True, last_batch_norm=True, readout_batchnorm=True, batch_norm_momentum=0.9,
           readout_hidden_dim=16, readout_layers=2, dropout=0.0, propagation_depth=2, aggregators=['mean', 'max', 'min', 'std'],
           scalers=['identity', 'amplification', 'attenuation'], readout_aggregators=['min', 'max', 'mean'], pretrans_layers=2,
           posttrans_layers=1, residual=True, pairwise_distances=True)


def main():
    dgl, PNA, PNALayer, Net3D, NTXent, NTXentMP = G.import_reference()
    mols = G.synth.make_dataset(10, seed=21)
    graphs = []
    for m in mols:
        g = dgl.graph((torch.from_numpy(m.src), torch.from_numpy(m.dst)), num_nodes=m.n_atoms)
        g.ndata['feat'] = torch.from_numpy(m.atom_feat)
        g.ndata['x'] = torch.from_numpy(m.coords.astype(np.float32))
        g.edata['feat'] = torch.from_numpy(m.bond_feat)
        graphs.append(g)
    bg = dgl.batch(graphs)
    torch.manual_seed(4)
    model = PNA(avg_d=1.0, device='cpu', **CFG)
    G.make_trained_like(model, 9)
    model.train()
    out = G.mols_to_npz(mols)
    out.update(G.sd_np(model, 'sd'))
    z = model(bg)
    cot = torch.randn(z.shape, generator=torch.Generator().manual_seed(6))
    (z * cot).sum().backward()
    out['out'], out['cot'], out['node_emb'] = z.detach().numpy(), cot.numpy(), bg.ndata['feat'].detach().numpy()
    out.update(G.grads_np(model, 'grad'))
    out.update(G.sd_np(model, 'sd_after'))
    np.savez_compressed(os.path.join(HERE, 'pna_pairwise.npz'), **out)
    print('wrote pna_pairwise.npz', z.shape, float(z.abs().mean()))


if __name__ == '__main__':
    main()
