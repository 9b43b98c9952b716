"""-m gpu: option combinations of the plugin modules that no yml of the reference uses, sampled (fixed seeds) from the constructor
arguments of models/pna.py:95-114 and models/net3d.py:15-18 and compared with the oracle, which restates the reference's forward for
any of them: outputs, node embeddings, running statistics and every parameter gradient.  The yml configurations have tests of
their own (test_gpu_models.py); this one is for the code paths beside them - hidden sizes that are not multiples of 4, blocks
without BatchNorm, one / three pretrans layers, no residual, activations other than ReLU, every scaler subset."""
import importlib
import random

import numpy as np
import pytest
import torch

from helpers import rel_err, synth
from oracle import pna3d_oracle as O
from test_gpu_models import _det_load, _lonely, _star, make_batch, param_grads

pytestmark = pytest.mark.gpu


def _pna_cfg(rng):
    hidden = rng.choice([16, 18, 20, 33])
    return dict(hidden_dim=hidden, target_dim=rng.choice([5, 8]), propagation_depth=rng.choice([1, 2, 3]),
                aggregators=rng.choice([['mean'], ['sum', 'std'], ['mean', 'sum', 'std', 'var'], ['var', 'mean']]),
                scalers=rng.choice([['identity'], ['identity', 'amplification'], ['attenuation', 'identity'],
                                    ['identity', 'amplification', 'attenuation'], ['amplification']]),
                readout_aggregators=rng.choice([['mean'], ['sum', 'mean'], ['mean', 'sum']]),
                readout_batchnorm=rng.choice([True, False]), readout_hidden_dim=rng.choice([hidden, 12]),
                readout_layers=rng.choice([1, 2, 3]), residual=rng.choice([True, False]),
                activation=rng.choice(['relu', 'relu', 'silu', 'tanh', 'leakyrelu']), last_activation=rng.choice(['none', 'none', 'relu']),
                mid_batch_norm=rng.choice([True, False]), last_batch_norm=rng.choice([True, False]),
                pretrans_layers=rng.choice([1, 2, 3]), posttrans_layers=rng.choice([1, 2]),
                batch_norm_momentum=rng.choice([0.1, 0.93]), dropout=0.0)


def _net3d_cfg(rng):
    hidden = rng.choice([20, 16, 10])
    return dict(hidden_dim=hidden, target_dim=rng.choice([5, 8]), propagation_depth=rng.choice([1, 2, 3]),
                batch_norm=rng.choice([True, False]), readout_batchnorm=rng.choice([True, False]),
                readout_aggregators=rng.choice([['mean'], ['sum', 'mean'], ['mean', 'sum']]), readout_layers=rng.choice([1, 2]),
                readout_hidden_dim=rng.choice([None, 12]), node_wise_output_layers=rng.choice([0, 1, 2]),
                fourier_encodings=rng.choice([0, 4]), reduce_func=rng.choice(['sum', 'mean']),
                update_net_layers=rng.choice([1, 2]), message_net_layers=rng.choice([1, 2, 3]),
                activation=rng.choice(['SiLU', 'ReLU']), batch_norm_momentum=rng.choice([0.1, 0.93]))


def _oracle(fwd, cfg, graph, sd, cot, dtype):
    """The oracle's forward + backward in `dtype` (fp32: the reference's arithmetic; fp64: what both are measured against)."""
    P = O.require_grad({k: (v.clone().to(dtype) if v.is_floating_point() else v.clone()) for k, v in sd.items()})
    g = {k: (v.to(dtype) if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in graph.items()}
    res = fwd(g, P, cfg, True)
    (res[0] * cot.to(dtype)).sum().backward()
    return res, P


def _compare(module, got_out, got_sd, o32, P32, o64, P64, what):
    """Outputs and running statistics within 1e-4 of the fp32 oracle; every parameter gradient no further from the fp64 oracle
    than 4x the fp32 oracle's own distance to it + 2e-4 of the tensor + 5e-6 of the largest gradient (some sampled
    configurations - BatchNorm over 7 rows, three ReLU blocks in a row - are badly conditioned in fp32 whoever computes them).
    Piecewise-linear activations add 1.5e-2 of the tensor: an activation within fp32 rounding of 0 is cut by one implementation
    and passed by the other, a whole gradient element either way - the fp32 oracle's own distance to the fp64 one moves between
    2e-4 and 9e-3 with the CPU it runs on for exactly this reason (measured on this configuration set)."""
    import os
    acts = [str(what.get(k, '')).lower() for k in ('activation', 'last_activation')]
    gate = 1.5e-2 if any(a in ('relu', 'leakyrelu') for a in acts) else 0.0
    assert rel_err(got_out.detach().cpu(), o32.detach()) < 1e-4, what
    names = [k for k in O.trainable(P64) if P64[k].grad is not None]
    scale = max(P64[k].grad.abs().max().item() for k in names)
    got = param_grads(module)
    for k in names:
        g64 = P64[k].grad
        e_ref = (P32[k].grad.double() - g64).abs().max().item()
        e_hip = (got[k].detach().cpu().double() - g64).abs().max().item()
        bound = 4 * e_ref + (2e-4 + gate) * g64.abs().max().item() + 5e-6 * scale
        if os.environ.get('I3D_TEST_VERBOSE') and e_hip > 0.3 * bound:
            print(f'{k}: hip {e_hip:.3e} oracle32 {e_ref:.3e} bound {bound:.3e} max {g64.abs().max().item():.3e}')
        assert e_hip <= bound, (k, e_hip, e_ref, bound, what)
    for k, v in P32.items():        # the oracle updates its running statistics in place
        if 'running' in k:
            assert rel_err(got_sd[k].cpu(), v.detach()) < 1e-4, (k, what)


@pytest.mark.parametrize('seed', range(40))
def test_pna_option_combinations_vs_oracle(seed):
    amd = importlib.import_module('3dinfomax_amd')
    rng = random.Random(1000 + seed)
    kw = _pna_cfg(rng)
    mols = synth.make_dataset(rng.choice([9, 24]), seed=200 + seed)
    if seed % 3 == 0:       # ragged batches: an isolated atom (in-degree 0, a graph without edges), a hub, a two-atom molecule
        mols = [_lonely(seed)] + mols[:4] + [_star(9 + seed % 7, seed), _star(1, seed + 1)] + mols[4:]
    pna = amd.PNA(avg_d=1.0, device='cuda:0', **kw)
    _det_load(pna, f'fuzz{seed}')
    sd = {k: v.clone() for k, v in pna.state_dict().items()}
    og2, _ = O.graphs_from_molecules(mols)
    cot = torch.from_numpy(np.random.default_rng(seed).standard_normal((len(mols), kw['target_dim'])).astype(np.float32))
    (o32, emb32), P32 = _oracle(O.pna_forward, O.pna_config(**kw), og2, sd, cot, torch.float32)
    (o64, _), P64 = _oracle(O.pna_forward, O.pna_config(**kw), og2, sd, cot, torch.float64)
    pna.cuda().train()
    g2, _ = make_batch(amd, mols)
    out = pna(g2)
    assert rel_err(g2.ndata['feat'].cpu(), emb32.detach()) < 1e-4, kw
    (out * cot.cuda()).sum().backward()
    _compare(pna, out, pna.state_dict(), o32, P32, o64, P64, kw)
    pna.eval()                                            # eval mode: running statistics of the step above
    with torch.no_grad():
        g2e, _ = make_batch(amd, mols)
        ref_e, _ = O.pna_forward(og2, {k: v.detach() for k, v in P32.items()}, O.pna_config(**kw), False)
        assert rel_err(pna(g2e).cpu(), ref_e) < 1e-4, kw


@pytest.mark.parametrize('seed', range(24))
def test_net3d_option_combinations_vs_oracle(seed):
    amd = importlib.import_module('3dinfomax_amd')
    rng = random.Random(2000 + seed)
    kw = _net3d_cfg(rng)
    mols = synth.make_dataset(rng.choice([7, 16]), seed=300 + seed)
    net = amd.Net3D(node_dim=0, edge_dim=1, avg_d=1.0, **kw)
    _det_load(net, f'fuzz3d{seed}')
    sd = {k: v.clone() for k, v in net.state_dict().items()}
    _, og3 = O.graphs_from_molecules(mols)
    cot = torch.from_numpy(np.random.default_rng(seed).standard_normal((len(mols), kw['target_dim'])).astype(np.float32))
    (o32, _), P32 = _oracle(O.net3d_forward, O.net3d_config(**kw), og3, sd, cot, torch.float32)
    (o64, _), P64 = _oracle(O.net3d_forward, O.net3d_config(**kw), og3, sd, cot, torch.float64)
    net.cuda().train()
    _, g3 = make_batch(amd, mols)
    out = net(g3)
    (out * cot.cuda()).sum().backward()
    _compare(net, out, net.state_dict(), o32, P32, o64, P64, kw)


def _orig_cfg(rng):
    towers = rng.choice([1, 2, 3, 5])
    hidden = towers * rng.choice([4, 6, 7])
    return dict(target_dim=rng.choice([1, 4]), hidden_dim=hidden, last_layer_dim=hidden, mid_batch_norm=rng.choice([True, False]),
                last_batch_norm=rng.choice([True, False]), graph_norm=rng.choice([True, False]), readout_batchnorm=True,
                edge_hidden_dim=rng.choice([8, 12, 7]), readout_hidden_dim=12, readout_layers=2, dropout=0.0, in_feat_dropout=0.0,
                propagation_depth=rng.choice([1, 2, 3]), towers=towers, divide_input_first=rng.choice([True, False]),
                divide_input_last=rng.choice([True, False]), edge_feat=rng.choice([True, True, False]),
                aggregators=rng.choice([['mean', 'sum', 'std'], ['mean'], ['sum', 'var'], ['mean', 'std', 'sum', 'var']]),
                scalers=rng.choice([['identity', 'amplification', 'attenuation'], ['identity'], ['amplification', 'identity']]),
                readout_aggregators=rng.choice([['mean', 'sum'], ['mean']]), pretrans_layers=rng.choice([1, 2]),
                posttrans_layers=rng.choice([1, 2]), residual=rng.choice([True, False]), avg_d=rng.choice([1.0, 1.4]), device='cpu')


@pytest.mark.parametrize('stack', [True, False])
@pytest.mark.parametrize('seed', range(16))
def test_tower_variant_option_combinations_vs_oracle(seed, stack, monkeypatch):
    """PNAOriginal (reference models/pna_original.py:119-319) over its constructor's options - towers 1-5, divided / undivided
    inputs, with and without edge features, graph norm, BatchNorm placement, one or two pre / posttrans layers - through the
    stacked-tower path (where the structure allows it; the model falls back by itself otherwise) and one tower after the other."""
    amd = importlib.import_module('3dinfomax_amd')
    po = importlib.import_module('3dinfomax_amd.pna_original')
    monkeypatch.setattr(po, 'TOWER_STACK', stack)
    rng = random.Random(3000 + seed)
    kw = _orig_cfg(rng)
    mols = synth.make_dataset(rng.choice([8, 20]), seed=400 + seed)
    if seed % 3 == 0:
        mols = [_lonely(seed)] + mols[:4] + [_star(9 + seed % 7, seed), _star(1, seed + 1)] + mols[4:]
    model = amd.PNAOriginal(**kw)
    _det_load(model, f'fuzzo{seed}')
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    og2, _ = O.graphs_from_molecules(mols)
    snorm = O.snorm_n(og2['batch_num_nodes'])
    cot = torch.from_numpy(np.random.default_rng(seed).standard_normal((len(mols), kw['target_dim'])).astype(np.float32))

    def fwd(g, P, cfg, training):
        return O.pna_original_forward(g, snorm.to(P['output.FC_layers.0.weight'].dtype), P, cfg, training)
    (o32, emb32), P32 = _oracle(fwd, kw, og2, sd, cot, torch.float32)
    (o64, _), P64 = _oracle(fwd, kw, og2, sd, cot, torch.float64)
    model.cuda().train()
    g2, _ = make_batch(amd, mols)
    out = model(g2, snorm.cuda())
    assert rel_err(g2.ndata['feat'].cpu(), emb32.detach()) < 1e-4, kw
    (out * cot.cuda()).sum().backward()
    _compare(model, out, model.state_dict(), o32, P32, o64, P64, dict(kw, activation='relu'))


def _std_loss(x):          # reference commons/losses.py:962-964
    return torch.mean(torch.relu(1 - torch.sqrt(x.var(dim=0) + 1e-04)))


def _cov_loss(x):          # reference commons/losses.py:954-959
    b, d = x.size()
    x = x - x.mean(dim=0)
    cov = (x.T @ x) / (b - 1)
    off = cov.flatten()[:-1].view(d - 1, d + 1)[:, 1:].flatten()
    return off.pow(2).sum() / d


def _uniformity_loss(x1, x2, t=2):      # reference commons/losses.py:946-951
    u1 = torch.pdist(x1, p=2).pow(2).mul(-t).exp().mean().log()
    u2 = torch.pdist(x2, p=2).pow(2).mul(-t).exp().mean().log()
    return (u1 + u2) / 2


@pytest.mark.parametrize('seed', range(20))
def test_ntxent_shapes_and_options_vs_reference_formula(seed):
    """NT-Xent / NTXentMultiplePositives (reference commons/losses.py:135-163, 206-258) over sampled batch sizes (none a multiple of
    the kernels' tiles), embedding widths, temperatures, conformer counts, with / without normalisation and with the variance /
    covariance / uniformity regularisers switched on: loss and both gradients against the reference's formulas in fp32."""
    losses = importlib.import_module('3dinfomax_amd.losses')
    rng = random.Random(4000 + seed)
    B, dim = rng.choice([2, 3, 7, 33, 65, 130, 257, 500]), rng.choice([8, 20, 37, 64, 256])
    conf = rng.choice([1, 1, 2, 3, 4])
    norm = rng.choice([True, True, False])
    tau = rng.choice([0.1, 0.5, 1.0]) if norm else 0.5
    # (the reference's covariance / uniformity terms only take 2-D embeddings: with conformers they raise there, :254-257)
    regs = dict(variance_reg=rng.choice([0, 0, 0.7]), covariance_reg=rng.choice([0, 0, 0.3]) if conf == 1 else 0,
                uniformity_reg=rng.choice([0, 0, 0.5]) if conf == 1 else 0)
    if conf > 1:
        regs['conformer_variance_reg'] = rng.choice([0, 0.4])
    gen = torch.Generator().manual_seed(seed)
    scale = 1.0 if norm else 0.15
    if regs['uniformity_reg']:
        scale = 0.1          # (exp(-2 |x - y|^2) of unit-variance rows in 37+ dimensions underflows: log(0) in the reference too)
    z1 = (torch.randn(B, dim, generator=gen) * scale).requires_grad_(True)
    z2 = (torch.randn(B * conf, dim, generator=gen) * scale).requires_grad_(True)
    ref = O.ntxent(z1, z2, tau, norm) if conf == 1 else O.ntxent_multiple_positives(z1, z2, tau, norm)
    z2v = z2 if conf == 1 else z2.view(B, conf, dim)          # (:230: the regularisers see [batch, conformers, dim])
    if regs.get('conformer_variance_reg') and conf > 1:
        ref = ref + regs['conformer_variance_reg'] * torch.mean(torch.relu(1 - torch.sqrt(z2v.var(dim=1) + 1e-04)))
    if regs['variance_reg'] and B > 1:
        ref = ref + regs['variance_reg'] * (_std_loss(z1) + _std_loss(z2v))
    if regs['covariance_reg'] and B > 1:
        ref = ref + regs['covariance_reg'] * (_cov_loss(z1) + _cov_loss(z2v))
    if regs['uniformity_reg'] and B > 1:
        ref = ref + regs['uniformity_reg'] * _uniformity_loss(z1, z2v)
    ref.backward()
    a, b = z1.detach().cuda().requires_grad_(True), z2.detach().cuda().requires_grad_(True)
    mod = (losses.NTXent if conf == 1 else losses.NTXentMultiplePositives)(norm=norm, tau=tau, **regs)
    loss = mod(a, b)
    loss.backward()
    what = (B, dim, conf, norm, tau, regs)
    assert abs(loss.item() - ref.item()) < 2e-5 * max(1.0, abs(ref.item())), what
    assert rel_err(a.grad.cpu(), z1.grad) < 5e-5 and rel_err(b.grad.cpu(), z2.grad) < 5e-5, what
