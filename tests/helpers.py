"""Shared test helpers: fixture loading, molecule reconstruction, tolerances."""
import importlib
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
amd = importlib.import_module('3dinfomax_amd')
synth = importlib.import_module('3dinfomax_amd.synth')

PNA_YML = dict(target_dim=256, hidden_dim=200, mid_batch_norm=True, last_batch_norm=True, readout_batchnorm=True,
               batch_norm_momentum=0.93, readout_hidden_dim=200, readout_layers=2, dropout=0.0, propagation_depth=7,
               aggregators=['mean', 'max', 'min', 'std'], scalers=['identity', 'amplification', 'attenuation'],
               readout_aggregators=['min', 'max', 'mean'], pretrans_layers=2, posttrans_layers=1, residual=True)
NET3D_YML = dict(target_dim=256, hidden_dim=20, hidden_edge_dim=20, node_wise_output_layers=0, message_net_layers=1,
                 update_net_layers=1, reduce_func='mean', fourier_encodings=4, propagation_depth=1, dropout=0.0,
                 batch_norm=True, readout_batchnorm=True, batch_norm_momentum=0.93, readout_hidden_dim=20,
                 readout_layers=1, readout_aggregators=['min', 'max', 'mean'])
PNA_SMALL = dict(PNA_YML, hidden_dim=16, target_dim=8, propagation_depth=2, readout_hidden_dim=16)
NET3D_SMALL = dict(NET3D_YML, target_dim=8)


def load(name):
    return np.load(os.path.join(GOLDEN, name))


def mols_from_npz(z, prefix='mol'):
    n_atoms, n_edges = z[f'{prefix}_n_atoms'], z[f'{prefix}_n_edges']
    mols, a0, e0 = [], 0, 0
    for n, e in zip(n_atoms, n_edges):
        mols.append(synth.Molecule(int(n), z[f'{prefix}_src'][e0:e0 + e], z[f'{prefix}_dst'][e0:e0 + e],
                                   z[f'{prefix}_atom_feat'][a0:a0 + n], z[f'{prefix}_bond_feat'][e0:e0 + e],
                                   z[f'{prefix}_coords'][a0:a0 + n]))
        a0 += n
        e0 += e
    return mols


def sd_from_npz(z, tag):
    pre = tag + '/'
    return {k[len(pre):]: torch.from_numpy(z[k].copy()) for k in z.files if k.startswith(pre)}


def rel_err(a, b):
    """max |a-b| / max |b|  (the '1e-4 relative fp32' of BASELINE.json:north_star)."""
    a = torch.as_tensor(a, dtype=torch.float64)
    b = torch.as_tensor(b, dtype=torch.float64)
    denom = b.abs().max().clamp(min=1e-30)
    return ((a - b).abs().max() / denom).item()


def grads_close(got, ref, rtol, what='', gate_floor=0.0):
    """Compare a dict of gradients with the reference's.  The absolute floor is tied to the largest gradient in the whole
    set.  Gradients that are analytically zero (a bias directly in front of a BatchNorm: the column sum of a BatchNorm
    input gradient) are pure rounding noise in both implementations - recognisable by a reference value below 1e-4 of the
    scale - and only have to stay noise-sized (5e-5 of the scale; the value moves with every change of a summation order).
    gate_floor (of the scale): room for ONE ReLU gate cut on one side and passed on the other (an activation within fp32
    rounding of 0) - a whole gradient element either way, on whichever small tensor it happens to hit."""
    scale = max(float(np.abs(np.asarray(v)).max()) for v in ref.values())
    for k, v in ref.items():
        a = torch.as_tensor(got[k], dtype=torch.float64).cpu()
        b = torch.as_tensor(v, dtype=torch.float64)
        err = (a - b).abs().max().item()
        bmax = b.abs().max().item()
        bound = rtol * bmax + ((5e-5 if bmax < 1e-4 * scale else 5e-6) + gate_floor) * scale
        if os.environ.get('I3D_TEST_VERBOSE') and err > 0.2 * bound:
            print(f'grads_close {what}{k}: err {err:.3e} bound {bound:.3e} max {bmax:.3e} scale {scale:.3e}')
        assert err <= bound, f'{what}{k}: err {err:.3e} > {bound:.3e}'


def close(a, b, rtol, atol=0.0):
    a = torch.as_tensor(a, dtype=torch.float64).cpu()
    b = torch.as_tensor(b, dtype=torch.float64).cpu()
    return (a - b).abs().max().item() <= rtol * b.abs().max().item() + atol


def grads_close_l2(got, ref, rtol, what='', floor=5e-6):
    """Per-parameter relative L2 error.  Used where the batch is big enough that fp32 rounding can flip the
    arg-max/arg-min of a near-tie (two different atoms whose feature value agrees to ~1e-6) in a max/min aggregator
    or readout: the routing of that one gradient element then differs between two correct fp32 implementations
    (max-norm error O(1e-2) on a few entries) while the gradient as a whole still agrees (L2 error << 1e-2)."""
    scale = max(float(np.linalg.norm(np.asarray(v).ravel())) for v in ref.values())
    for k, v in ref.items():
        a = torch.as_tensor(got[k], dtype=torch.float64).cpu().flatten()
        b = torch.as_tensor(v, dtype=torch.float64).flatten()
        err = (a - b).norm().item()
        bound = rtol * b.norm().item() + floor * scale      # floor: analytically-zero gradients (a shift a later BatchNorm removes)
        if os.environ.get('I3D_TEST_VERBOSE') and err > 0.2 * bound:
            print(f'grads_close_l2 {what}{k}: rel {err / max(b.norm().item(), 1e-30):.3f} err {err:.3e} bound {bound:.3e} norm {b.norm().item():.3e} scale {scale:.3e}')
        assert err <= bound or os.environ.get('I3D_TEST_VERBOSE') == 'noassert', f'{what}{k}: L2 err {err:.3e} > {bound:.3e}'


def _segment_first_argext(values, ptr, largest):
    """[n_segments, F] position (inside its segment) of the first maximum / minimum of every column; values [rows, F],
    ptr [n_segments + 1]; empty segments get 0"""
    from oracle import pna3d_oracle as O
    n = ptr.shape[0] - 1
    deg = ptr[1:] - ptr[:-1]
    out = torch.zeros(n, values.shape[1], dtype=torch.long)
    for D in sorted(set(deg.tolist())):
        if D == 0:
            continue
        segs = torch.nonzero(deg == D).flatten()
        rows = ptr[segs][:, None] + torch.arange(D)[None, :]
        out[segs] = O.first_argext(values[rows], largest)
    return out


def hip_routing(pna, g2, out):
    """The mailbox / graph positions the HIP kernels of this forward pass route their max / min gradients to, in the form
    oracle.pna_forward(route=...) takes: per layer from the messages exactly as the aggregation kernels read them
    (pna_native.debug_messages: same expression, same build flags), for the readout from the node embeddings the forward
    left on the graph.  Call it BEFORE backward() (the forward context is released there)."""
    import importlib
    native = importlib.import_module('3dinfomax_amd.pna_native')
    idx = g2.index()
    in_ptr, graph_ptr = idx.in_ptr.cpu().long(), idx.graph_ptr.cpu().long()
    route = {}
    for l, layer in enumerate(pna.node_gnn.mp_layers):
        f_msg = list(layer.pretrans.fully_connected)[-1].out_dim
        msg = native.debug_messages(out, l, idx.num_edges, f_msg).cpu()
        route[f'layer{l}'] = {'max': _segment_first_argext(msg, in_ptr, True), 'min': _segment_first_argext(msg, in_ptr, False)}
    emb = g2.ndata['feat'].detach().cpu()
    route['readout'] = {'max': _segment_first_argext(emb, graph_ptr, True), 'min': _segment_first_argext(emb, graph_ptr, False)}
    return route


def routing_flips(route, capture, n_layers):
    """(positions where the oracle's own first extremum differs from the routed one, positions compared) over all layers and
    the readout, and the largest gap - relative to the tensor's scale - between the oracle's values at the two positions of a
    flip: a flip is legitimate only as a NEAR-TIE (two candidates that agree to fp32 rounding)."""
    flips = total = 0
    worst = 0.0
    for l in range(n_layers):
        cap = capture[f'layer{l}']
        e = cap['e'].detach()
        own = cap['argext']
        for kind in ('max', 'min'):
            d = own[kind] != route[f'layer{l}'][kind]
            flips += int(d.sum())
            total += d.numel()
    own = capture.get('readout', {})
    for kind in ('max', 'min'):
        if kind in own:
            d = own[kind] != route['readout'][kind]
            flips += int(d.sum())
            total += d.numel()
    return flips, total
