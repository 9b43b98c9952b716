"""Fixture for PNAOriginal(gru_enable=True) (reference models/pna_original.py:64-84, 179-180, 190-193: a one-step nn.GRU between
the layers, input = the layer's input, hidden state = the layer's output): the unmodified reference model, forward + backward ->
tests/golden/pna_original_gru.npz.

    python tests/golden/gen_golden_originalgru.py          (build container only: imports /root/reference)
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as G  # noqa: E402


def main():
    dgl = G.import_reference()[0]
    from models.pna_original import PNAOriginal
    mols = G.synth.make_dataset(6, seed=44)
    graphs = []
    for m in mols:
        g = dgl.graph((torch.from_numpy(m.src), torch.from_numpy(m.dst)), num_nodes=m.n_atoms)
        g.ndata['feat'] = torch.from_numpy(m.atom_feat)
        g.ndata['x'] = torch.from_numpy(m.coords.astype(np.float32))
        g.edata['feat'] = torch.from_numpy(m.bond_feat)
        graphs.append(g)
    g2 = dgl.batch(graphs)
    torch.manual_seed(323)
    model = PNAOriginal(**dict(G.PNA_ORIG_KW, gru_enable=True))
    G.make_trained_like(model, 24)
    with torch.no_grad():           # GRU weights of O(1) effect (default init: U(-1/sqrt(H), 1/sqrt(H)))
        for n_, p in model.named_parameters():
            if '.gru.' in n_ and 'weight' in n_:
                p.mul_(2.0)
    with torch.no_grad():
        for n_, p in model.named_parameters():
            if n_.endswith('mixing_network.weight') or ('FC_layers' in n_ and n_.endswith('weight')):
                p.mul_(2.0)
    model.train()
    out = G.mols_to_npz(mols)
    out.update(G.sd_np(model, 'sd'))
    snorm = torch.cat([torch.full((m.n_atoms, 1), 1.0 / float(m.n_atoms)) for m in mols]).sqrt()
    z = model(g2, snorm)
    out['out'], out['node_emb'] = z.detach().numpy(), g2.ndata['feat'].detach().numpy()
    c = torch.from_numpy(G.det_fill(tuple(z.shape), 'cot_origgru'))
    out['cot'] = c.numpy()
    (z * c).sum().backward()
    out.update({f'grad/{k}': p.grad.detach().numpy().copy() for k, p in model.named_parameters() if p.grad is not None})
    out.update(G.sd_np(model, 'sd_after'))
    np.savez_compressed(os.path.join(HERE, 'pna_original_gru.npz'), **out)
    print('wrote pna_original_gru.npz', tuple(z.shape), float(z.detach().abs().mean()))


if __name__ == '__main__':
    main()
