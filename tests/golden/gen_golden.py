"""Generate the golden fixtures in tests/golden/*.npz by importing the UNMODIFIED reference
modules from /root/reference (build container only - the reference never travels).

    python tests/golden/gen_golden.py

Third-party packages the reference imports but this image lacks (dgl, ogb, tensorboard) are
replaced by the small stand-ins in tests/golden/_stubs/ (restated semantics, SURVEY.md
Appendix A).  Only DATA is committed: inputs, the reference's own state_dicts and the
reference's outputs/gradients.  Inputs come from the seeded synthetic generator in
3dinfomax_amd/synth.py because the reference's data blobs are absent.
"""
import collections
import collections.abc
import importlib
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = '/root/reference'


def import_reference():
    sys.dont_write_bytecode = True
    sys.path.insert(0, os.path.join(HERE, '_stubs'))
    sys.path.insert(0, REF)
    collections.MutableMapping = collections.abc.MutableMapping       # commons/utils.py:4 on py3.10
    tb = types.ModuleType('torch.utils.tensorboard')                   # commons/utils.py:12
    tb.SummaryWriter = object
    sys.modules['torch.utils.tensorboard'] = tb
    pkg = types.ModuleType('models')                                   # bypass models/__init__.py:8-16
    pkg.__path__ = [os.path.join(REF, 'models')]
    sys.modules['models'] = pkg
    import dgl  # noqa: F401  (the stand-in)
    from models.pna import PNA, PNALayer
    from models.net3d import Net3D
    from commons.losses import NTXent, NTXentMultiplePositives
    return dgl, PNA, PNALayer, Net3D, NTXent, NTXentMultiplePositives


sys.path.insert(0, REPO)
sys.path.insert(0, HERE)
synth = importlib.import_module('3dinfomax_amd.synth')
from fill import det_fill  # noqa: E402

PNA_YML = dict(target_dim=256, hidden_dim=200, mid_batch_norm=True, last_batch_norm=True, readout_batchnorm=True,
               batch_norm_momentum=0.93, readout_hidden_dim=200, readout_layers=2, dropout=0.0, propagation_depth=7,
               aggregators=['mean', 'max', 'min', 'std'], scalers=['identity', 'amplification', 'attenuation'],
               readout_aggregators=['min', 'max', 'mean'], pretrans_layers=2, posttrans_layers=1, residual=True)
NET3D_YML = dict(target_dim=256, hidden_dim=20, hidden_edge_dim=20, node_wise_output_layers=0, message_net_layers=1,
                 update_net_layers=1, reduce_func='mean', fourier_encodings=4, propagation_depth=1, dropout=0.0,
                 batch_norm=True, readout_batchnorm=True, batch_norm_momentum=0.93, readout_hidden_dim=20,
                 readout_layers=1, readout_aggregators=['min', 'max', 'mean'])


def dgl_graphs(dgl, mols):
    g2, g3 = [], []
    for m in mols:
        g = dgl.graph((torch.from_numpy(m.src), torch.from_numpy(m.dst)), num_nodes=m.n_atoms)
        g.ndata['feat'] = torch.from_numpy(m.atom_feat)
        g.edata['feat'] = torch.from_numpy(m.bond_feat)
        g2.append(g)
        s, d = synth.complete_graph_edges(m.n_atoms)
        c = dgl.graph((torch.from_numpy(s), torch.from_numpy(d)), num_nodes=m.n_atoms)
        c.edata['d'] = torch.from_numpy(synth.pairwise_distances(m.coords, s, d))
        g3.append(c)
    return dgl.batch(g2), dgl.batch(g3)


def mols_to_npz(mols, prefix='mol'):
    out = {f'{prefix}_n_atoms': np.array([m.n_atoms for m in mols]),
           f'{prefix}_n_edges': np.array([m.src.shape[0] for m in mols]),
           f'{prefix}_src': np.concatenate([m.src for m in mols]),
           f'{prefix}_dst': np.concatenate([m.dst for m in mols]),
           f'{prefix}_atom_feat': np.concatenate([m.atom_feat for m in mols]),
           f'{prefix}_bond_feat': np.concatenate([m.bond_feat for m in mols]),
           f'{prefix}_coords': np.concatenate([m.coords for m in mols])}
    return out


def sd_np(module, tag):
    return {f'{tag}/{k}': v.detach().cpu().numpy().copy() for k, v in module.state_dict().items()}


def make_trained_like(module, seed):
    """O(1) pre-BN scale: linear weights x in_dim (undoes gain=1/in_dim), random BN affine, non-zero biases."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p in module.named_parameters():
            if name.endswith('linear.weight'):
                p.mul_(p.shape[1] * 0.7)
            elif name.endswith('linear.bias') or name.endswith('batch_norm.bias'):
                p.copy_(torch.randn(p.shape, generator=g) * 0.2)
            elif name.endswith('batch_norm.weight'):
                p.copy_(1 + torch.randn(p.shape, generator=g) * 0.2)


def grads_np(module, tag):
    return {f'{tag}/{k}': p.grad.detach().numpy().copy() for k, p in module.named_parameters()}


# -------------------------------------------------------------------------------------------
def gen_pna_layer(dgl, PNALayer):
    """G1: one PNALayer fwd+bwd, F=8, degrees {0,1,2,3,4,6}, exact ties, D=1 nodes."""
    torch.manual_seed(11)
    F = 8
    # node 0: deg 6, node 1: deg 4, node 2: deg 3, node 3: deg 2, nodes 4,5: deg 1, node 6: deg 0 (isolated), ...
    dst = [0] * 6 + [1] * 4 + [2] * 3 + [3] * 2 + [4] + [5] + [7, 7, 8, 8, 8]
    src = [1, 2, 3, 4, 5, 7, 0, 2, 3, 8, 0, 1, 9, 0, 1, 0, 0, 8, 9, 7, 9, 9]
    perm = torch.randperm(len(src), generator=torch.Generator().manual_seed(3))
    src = torch.tensor(src)[perm]
    dst = torch.tensor(dst)[perm]
    n = 10
    h = torch.randn(n, F)
    h[9] = h[7]            # identical source rows -> bit-identical messages -> ties in max/min
    ef = torch.randn(len(src), F)
    # make the two edges 9->8 and 7->8 carry identical edge features as well (exact tie at node 8)
    e_a = [i for i in range(len(src)) if dst[i] == 8 and src[i] in (7, 9)]
    for i in e_a[1:]:
        ef[i] = ef[e_a[0]]
    out = {}
    for regime in ('init', 'trained'):
        torch.manual_seed(5)
        layer = PNALayer(in_dim=F, out_dim=F, in_dim_edges=F, aggregators=['mean', 'max', 'min', 'std'],
                         scalers=['identity', 'amplification', 'attenuation'], mid_batch_norm=True,
                         last_batch_norm=True, batch_norm_momentum=0.93, posttrans_layers=1, pretrans_layers=2)
        if regime == 'trained':
            make_trained_like(layer, 7)
        layer.train()
        g = dgl.graph((src, dst), num_nodes=n)
        hh = h.clone().requires_grad_(True)
        ee = ef.clone().requires_grad_(True)
        g.ndata['feat'] = hh
        g.edata['feat'] = ee
        cap = {}
        h1 = layer.pretrans.register_forward_hook(lambda m, i, o: cap.__setitem__('e', o))
        h2 = layer.posttrans.register_forward_hook(lambda m, i, o: cap.__setitem__('hcat', i[0]))
        out.update(sd_np(layer, f'{regime}/sd'))
        layer(g)
        h1.remove(), h2.remove()
        hn = g.ndata['feat']
        cot = torch.from_numpy(det_fill(tuple(hn.shape), f'pna_layer_cot_{regime}'))
        (hn * cot).sum().backward()
        out[f'{regime}/e'] = cap['e'].detach().numpy()
        out[f'{regime}/agg'] = cap['hcat'][:, F:].detach().numpy()
        out[f'{regime}/h_out'] = hn.detach().numpy()
        out[f'{regime}/cot'] = cot.numpy()
        out[f'{regime}/grad_h'] = hh.grad.numpy()
        out[f'{regime}/grad_ef'] = ee.grad.numpy()
        out.update(grads_np(layer, f'{regime}/grad'))
        out.update(sd_np(layer, f'{regime}/sd_after'))
    out.update(src=src.numpy(), dst=dst.numpy(), h=h.numpy(), ef=ef.numpy(), n=np.array(n))
    np.savez_compressed(os.path.join(HERE, 'pna_layer.npz'), **out)
    print('pna_layer.npz', len(out), 'arrays')


def gen_models_small(dgl, PNA, Net3D, NTXent):
    """G2 + G3 + G5: PNA(F=16,L=2,target 8) and Net3D(yml, target 8) on 6 molecules; fwd, bwd, BN buffers,
    eval mode; then 3 Adam steps of the joint contrastive objective."""
    mols = synth.make_dataset(6, seed=42)
    out = mols_to_npz(mols)
    pna_kw = dict(PNA_YML, hidden_dim=16, target_dim=8, propagation_depth=2, readout_hidden_dim=16)
    n3_kw = dict(NET3D_YML, target_dim=8)
    for regime in ('init', 'trained'):
        torch.manual_seed(123)
        pna = PNA(avg_d=1.0, device='cpu', **pna_kw)
        net = Net3D(node_dim=0, edge_dim=1, avg_d=1.0, **n3_kw)
        if regime == 'trained':
            make_trained_like(pna, 17)
            make_trained_like(net, 19)
        pna.train(), net.train()
        out.update(sd_np(pna, f'{regime}/pna_sd'))
        out.update(sd_np(net, f'{regime}/net3d_sd'))
        g2, g3 = dgl_graphs(dgl, mols)
        cap = {}
        hk1 = pna.node_gnn.mp_layers[0].pretrans.register_forward_hook(lambda m, i, o: cap.__setitem__('e0', o))
        hk2 = pna.node_gnn.mp_layers[0].posttrans.register_forward_hook(lambda m, i, o: cap.__setitem__('hcat0', i[0]))
        z2 = pna(g2)
        hk1.remove(), hk2.remove()
        z3 = net(g3)
        out[f'{regime}/pna_out'] = z2.detach().numpy()
        out[f'{regime}/pna_node_emb'] = g2.ndata['feat'].detach().numpy()
        out[f'{regime}/pna_e0'] = cap['e0'].detach().numpy()
        out[f'{regime}/pna_agg0'] = cap['hcat0'][:, 16:].detach().numpy()
        out[f'{regime}/net3d_out'] = z3.detach().numpy()
        out[f'{regime}/net3d_node_emb'] = g3.ndata['feat'].detach().numpy()
        c2 = torch.from_numpy(det_fill(tuple(z2.shape), f'cot2_{regime}'))
        c3 = torch.from_numpy(det_fill(tuple(z3.shape), f'cot3_{regime}'))
        out[f'{regime}/cot2'], out[f'{regime}/cot3'] = c2.numpy(), c3.numpy()
        ((z2 * c2).sum() + (z3 * c3).sum()).backward()
        out.update(grads_np(pna, f'{regime}/pna_grad'))
        out.update(grads_np(net, f'{regime}/net3d_grad'))
        out.update(sd_np(pna, f'{regime}/pna_sd_after'))
        out.update(sd_np(net, f'{regime}/net3d_sd_after'))
        # eval mode with the running statistics after that single step
        pna.eval(), net.eval()
        g2, g3 = dgl_graphs(dgl, mols)
        with torch.no_grad():
            out[f'{regime}/pna_out_eval'] = pna(g2).numpy()
            out[f'{regime}/net3d_out_eval'] = net(g3).numpy()
    np.savez_compressed(os.path.join(HERE, 'models_small.npz'), **out)
    print('models_small.npz', len(out), 'arrays')

    # G5: three optimisation steps (reference trainer/trainer.py:116-124 semantics)
    out = mols_to_npz(mols)
    torch.manual_seed(123)
    pna = PNA(avg_d=1.0, device='cpu', **pna_kw)
    net = Net3D(node_dim=0, edge_dim=1, avg_d=1.0, **n3_kw)
    make_trained_like(pna, 17)
    make_trained_like(net, 19)
    out.update(sd_np(pna, 'pna_sd'))
    out.update(sd_np(net, 'net3d_sd'))
    loss_fn = NTXent(tau=0.1)
    # reference trainer/self_supervised_trainer.py:78-86: BN params in their own group (same lr, wd 0)
    params = list(pna.named_parameters()) + list(net.named_parameters())
    bn = [p for n_, p in params if 'batch_norm' in n_]
    rest = [p for n_, p in params if 'batch_norm' not in n_]
    optim = torch.optim.Adam([{'params': bn, 'weight_decay': 0}, {'params': rest}], lr=8e-5)
    losses = []
    pna.train(), net.train()
    for step in range(3):
        g2, g3 = dgl_graphs(dgl, mols)
        loss = loss_fn(pna(g2), net(g3), nodes_per_graph=g2.batch_num_nodes())
        loss.backward()
        optim.step()
        optim.zero_grad()
        losses.append(loss.item())
    out['losses'] = np.array(losses, dtype=np.float64)
    out.update(sd_np(pna, 'pna_sd_final'))
    out.update(sd_np(net, 'net3d_sd_final'))
    np.savez_compressed(os.path.join(HERE, 'train3.npz'), **out)
    print('train3.npz losses', losses)


def gen_ntxent(NTXent, NTXentMultiplePositives):
    """G4: NT-Xent and the multiple-positives variant incl. gradients, B in {2,8,64}, dim 256 (64 for B=64)."""
    out = {}
    for B, dim in ((2, 256), (8, 256), (64, 64)):
        g = torch.Generator().manual_seed(B)
        z1 = torch.randn(B, dim, generator=g).requires_grad_(True)
        z2 = torch.randn(B, dim, generator=g).requires_grad_(True)
        loss = NTXent(tau=0.1)(z1, z2)
        loss.backward()
        out.update({f'nt/{B}/z1': z1.detach().numpy(), f'nt/{B}/z2': z2.detach().numpy(),
                    f'nt/{B}/loss': np.array(loss.item()), f'nt/{B}/g1': z1.grad.numpy(), f'nt/{B}/g2': z2.grad.numpy()})
        C = 3
        z1 = torch.randn(B, dim, generator=g).requires_grad_(True)
        z2 = torch.randn(B * C, dim, generator=g).requires_grad_(True)
        loss = NTXentMultiplePositives(tau=0.1)(z1, z2)
        loss.backward()
        out.update({f'mp/{B}/z1': z1.detach().numpy(), f'mp/{B}/z2': z2.detach().numpy(),
                    f'mp/{B}/loss': np.array(loss.item()), f'mp/{B}/g1': z1.grad.numpy(), f'mp/{B}/g2': z2.grad.numpy()})
    np.savez_compressed(os.path.join(HERE, 'ntxent.npz'), **out)
    print('ntxent.npz', len(out), 'arrays')


def det_state_dict(module, tag):
    """Closed-form weights so the F=200/L=7 fixture needs no 20 MB weight blob."""
    sd = module.state_dict()
    new = {}
    for k, v in sd.items():
        if k.endswith('num_batches_tracked'):
            new[k] = v
        elif k.endswith('running_var'):
            new[k] = torch.from_numpy(det_fill(tuple(v.shape), f'{tag}/{k}', 0.2, 1.0))
        elif k.endswith('batch_norm.weight'):
            new[k] = torch.from_numpy(det_fill(tuple(v.shape), f'{tag}/{k}', 0.2, 1.0))
        elif k.endswith('linear.weight'):
            new[k] = torch.from_numpy(det_fill(tuple(v.shape), f'{tag}/{k}', 1.2 / np.sqrt(v.shape[1])))
        elif 'embedding_list' in k or k == 'node_embedding':
            new[k] = torch.from_numpy(det_fill(tuple(v.shape), f'{tag}/{k}', 1.0))
        else:
            new[k] = torch.from_numpy(det_fill(tuple(v.shape), f'{tag}/{k}', 0.2))
    return new


def gen_full_config(dgl, PNA, Net3D, NTXent):
    """G6: the real pre-train_QM9.yml dimensions (F=200, L=7 and L=4) on 16 molecules, checksums + samples only."""
    mols = synth.make_dataset(16, seed=7)
    out = mols_to_npz(mols)
    for L in (7, 4):
        pna = PNA(avg_d=1.0, device='cpu', **dict(PNA_YML, propagation_depth=L))
        net = Net3D(node_dim=0, edge_dim=1, avg_d=1.0, **NET3D_YML)
        pna.load_state_dict(det_state_dict(pna, f'pna{L}'))
        net.load_state_dict(det_state_dict(net, 'net3d'))
        pna.train(), net.train()
        g2, g3 = dgl_graphs(dgl, mols)
        z2, z3 = pna(g2), net(g3)
        loss = NTXent(tau=0.1)(z2, z3)
        loss.backward()
        emb = g2.ndata['feat'].detach()
        out[f'L{L}/loss'] = np.array(loss.item())
        out[f'L{L}/pna_out'] = z2.detach().numpy()
        out[f'L{L}/net3d_out'] = z3.detach().numpy()
        out[f'L{L}/node_emb_rows'] = emb[::7].numpy()          # every 7th node row
        out[f'L{L}/node_emb_abs_sum'] = np.array(emb.abs().sum().item())
        for k, p in list(pna.named_parameters()):
            g = p.grad
            out[f'L{L}/pna_grad_norm/{k}'] = np.array(g.norm().item())
        w = pna.node_gnn.mp_layers[0].posttrans.fully_connected[0].linear.weight.grad
        out[f'L{L}/pna_grad_sample/post0_w'] = w[::16, ::64].numpy()
        w = pna.node_gnn.mp_layers[L - 1].pretrans.fully_connected[0].linear.weight.grad
        out[f'L{L}/pna_grad_sample/preL_w'] = w[::16, ::32].numpy()
        for k, p in net.named_parameters():
            out[f'L{L}/net3d_grad/{k}'] = p.grad.numpy()
    np.savez_compressed(os.path.join(HERE, 'full_config.npz'), **out)
    print('full_config.npz', len(out), 'arrays')


PNA_ORIG_KW = dict(target_dim=4, hidden_dim=20, last_layer_dim=20, mid_batch_norm=True, last_batch_norm=True,
                   graph_norm=True, readout_batchnorm=True, edge_hidden_dim=12, readout_hidden_dim=12, readout_layers=2,
                   dropout=0.0, in_feat_dropout=0.0, propagation_depth=3, towers=5, divide_input_first=False,
                   divide_input_last=True, aggregators=['mean', 'max', 'min', 'std'],
                   scalers=['identity', 'amplification', 'attenuation'], readout_aggregators=['mean', 'max', 'min', 'sum'],
                   pretrans_layers=1, posttrans_layers=1, residual=True, avg_d=1.4, device='cpu')
PNA_SIMPLE_KW = dict(target_dim=4, hidden_dim=24, last_layer_dim=24, mid_batch_norm=True, last_batch_norm=True,
                     readout_batchnorm=True, readout_hidden_dim=12, readout_layers=2, dropout=0.0, in_feat_dropout=0.0,
                     propagation_depth=2, aggregators=['mean', 'max', 'min', 'std'],
                     scalers=['identity', 'amplification', 'attenuation'], readout_aggregators=['min', 'max', 'mean'],
                     posttrans_layers=1, residual=True, avg_d=1.4, batch_norm_momentum=0.1)


def gen_pna_original(dgl):
    """G8: the original-PNA variants of reference models/pna_original.py (config shape of
    configs/pna_original.yml: towers 5, graph_norm, divide_input_first False / last True)."""
    from models.pna_original import PNAOriginal, PNAOriginalSimple
    mols = synth.make_dataset(6, seed=42)
    out = mols_to_npz(mols)
    for tag, cls, kw in (('orig', PNAOriginal, PNA_ORIG_KW), ('simple', PNAOriginalSimple, PNA_SIMPLE_KW)):
        torch.manual_seed(321)
        model = cls(**kw)
        make_trained_like(model, 23)
        with torch.no_grad():       # nn.Linear mixing / MLPReadout layers are not FCLayers: scale them as well
            for n_, p in model.named_parameters():
                if n_.endswith('mixing_network.weight') or ('FC_layers' in n_ and n_.endswith('weight')):
                    p.mul_(2.0)
        model.train()
        out.update(sd_np(model, f'{tag}/sd'))
        g2, _ = dgl_graphs(dgl, mols)
        snorm = torch.cat([torch.full((m.n_atoms, 1), 1.0 / float(m.n_atoms)) for m in mols]).sqrt()
        z = model(g2, snorm) if tag == 'orig' else model(g2)
        out[f'{tag}/out'] = z.detach().numpy()
        out[f'{tag}/node_emb'] = g2.ndata['feat'].detach().numpy()
        c = torch.from_numpy(det_fill(tuple(z.shape), f'cot_{tag}'))
        out[f'{tag}/cot'] = c.numpy()
        (z * c).sum().backward()
        out.update({f'{tag}/grad/{k}': p.grad.detach().numpy().copy() for k, p in model.named_parameters()
                    if p.grad is not None})
        out.update(sd_np(model, f'{tag}/sd_after'))
    np.savez_compressed(os.path.join(HERE, 'pna_original.npz'), **out)
    print('pna_original.npz', len(out), 'arrays')


if __name__ == '__main__':
    torch.set_num_threads(4)
    dgl, PNA, PNALayer, Net3D, NTXent, NTXentMP = import_reference()
    gen_pna_layer(dgl, PNALayer)
    gen_models_small(dgl, PNA, Net3D, NTXent)
    gen_ntxent(NTXent, NTXentMP)
    gen_full_config(dgl, PNA, Net3D, NTXent)
    gen_pna_original(dgl)
    for f in sorted(os.listdir(HERE)):
        if f.endswith('.npz'):
            print(f, os.path.getsize(os.path.join(HERE, f)) // 1024, 'KiB')
