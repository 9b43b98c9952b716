"""CPU ORACLE - TEST INFRASTRUCTURE ONLY, never part of the product path.

A plain-PyTorch (CPU, fp32, autograd) restatement of the 3DInfomax pre-training hot path:
PNA (reference models/pna.py), Net3D (reference models/net3d.py), the FCLayer/MLP towers
(reference models/base_layers.py), Atom/BondEncoder (reference commons/mol_encoder.py),
fourier_encode_dist (reference commons/utils.py:103-110) and NT-Xent (reference
commons/losses.py:126-163, 206-258), op-for-op in the reference's order - including DGL's
degree-bucketed reduce and the materialised concats - so that it doubles as the CPU baseline.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file.

PARITY PINNING: the reference holds no tests or golden vectors for this path,
so this oracle is pinned against the *reference itself*, imported in the build container by
tests/golden/gen_golden.py (DGL/ogb stubbed), through the committed fixtures in
tests/golden/*.npz - see tests/test_oracle_golden.py.

Everything is functional: parameters/buffers come in one dict keyed exactly like the
reference modules' state_dict, graphs as plain index tensors.
"""
import math
from typing import Dict, List

import torch
import torch.nn.functional as F

EPS = 1e-5  # reference models/pna.py:14

ATOM_FEATURE_DIMS = [119, 5, 12, 12, 10, 6, 6, 2, 2]   # ogb constant, reference commons/mol_encoder.py:6
BOND_FEATURE_DIMS = [5, 6, 2]                          # ogb constant, reference commons/mol_encoder.py:7


# ------------------------------------------------------------------------------------------
# towers: reference models/base_layers.py
# ------------------------------------------------------------------------------------------
def _act(name):
    """reference models/base_layers.py:9-20 (get_activation)."""
    if name is None:
        return None
    name = name.lower()
    return {'relu': F.relu, 'silu': F.silu, 'none': None, 'sigmoid': torch.sigmoid, 'tanh': torch.tanh,
            'leakyrelu': F.leaky_relu}[name]


def fc_layer(x, P, prefix, activation, batch_norm, momentum, training):
    """Linear -> activation -> (dropout=0) -> BatchNorm1d.  reference models/base_layers.py:100-111."""
    h = F.linear(x, P[prefix + '.linear.weight'], P[prefix + '.linear.bias'])
    act = _act(activation)
    if act is not None:
        h = act(h)
    if batch_norm:
        h = F.batch_norm(h, P[prefix + '.batch_norm.running_mean'], P[prefix + '.batch_norm.running_var'],
                         P[prefix + '.batch_norm.weight'], P[prefix + '.batch_norm.bias'],
                         training, momentum, 1e-5)
        if training:
            P[prefix + '.batch_norm.num_batches_tracked'] += 1
    return h


def mlp(x, P, prefix, layers, mid_activation, last_activation, mid_batch_norm, last_batch_norm, momentum,
        training):
    """reference models/base_layers.py:119-147 (MLP: hidden FCLayers then a last FCLayer)."""
    if layers <= 1:
        return fc_layer(x, P, f'{prefix}.fully_connected.0', last_activation, last_batch_norm, momentum, training)
    for l in range(layers - 1):
        x = fc_layer(x, P, f'{prefix}.fully_connected.{l}', mid_activation, mid_batch_norm, momentum, training)
    return fc_layer(x, P, f'{prefix}.fully_connected.{layers - 1}', last_activation, last_batch_norm, momentum,
                    training)


def embedding_sum(idx, P, prefix, n_tables):
    """Sum of per-column embedding lookups.  reference commons/mol_encoder.py:34-42, 65-73."""
    out = 0
    for k in range(n_tables):
        out = out + F.embedding(idx[:, k], P[f'{prefix}.{k}.weight'])
    return out


# ------------------------------------------------------------------------------------------
# DGL message-passing semantics (third-party `dgl`, unpinned in environment.yml:12; restated
# from its documented behaviour)
# ------------------------------------------------------------------------------------------
def _csr_by_dst(dst, n):
    deg = torch.bincount(dst, minlength=n)
    order = torch.sort(dst, stable=True)[1]
    rowptr = torch.zeros(n + 1, dtype=torch.long)
    rowptr[1:] = torch.cumsum(deg, 0)
    return deg, order, rowptr


def degree_bucketed_reduce(msgs, dst, n, reduce_fn, out_dim, with_nodes=False):
    """update_all with a UDF reduce: one call per distinct in-degree D>0, mailbox [n_D, D, F] ordered by
    edge id; isolated nodes get zeros.  with_nodes: reduce_fn(mailbox, D, nodes) (test routing, see pna_reduce)."""
    deg, order, rowptr = _csr_by_dst(dst, n)
    out = torch.zeros(n, out_dim, dtype=msgs.dtype)
    for D in sorted(set(deg.tolist())):
        if D == 0:
            continue
        nodes = torch.nonzero(deg == D).flatten()
        eids = order[rowptr[nodes][:, None] + torch.arange(D)[None, :]]
        out = out.index_copy(0, nodes, reduce_fn(msgs[eids], D, nodes) if with_nodes else reduce_fn(msgs[eids], D))
    return out


def first_argext(h, largest=True):
    """index of the FIRST maximum / minimum along dim -2 (what torch.max / torch.min(dim) route their gradient to on CPU,
    SURVEY.md Appendix A, and what the HIP kernels pick)"""
    ext = h.max(dim=-2, keepdim=True)[0] if largest else h.min(dim=-2, keepdim=True)[0]
    return (h == ext).to(torch.uint8).argmax(dim=-2)


def routed_ext(h, idx):
    """max / min along dim -2 with the gradient routed to the GIVEN position idx [..., F] (test infrastructure: the GPU
    parity tests pass the positions the HIP kernels picked, so that a near-tie that fp32 rounding resolves differently on
    the two sides does not move whole gradient rows; the forward value differs from the true extremum by the tie's gap)"""
    return h.gather(-2, idx.unsqueeze(-2)).squeeze(-2)


def segment_readout(x, batch_num_nodes, op, route=None, capture=None):
    """dgl.readout_nodes: per-graph segment reduce in batch order.  route: {'max': [B, F], 'min': [B, F]} positions inside
    each graph the max / min gradients are routed to (tests); capture: receives this side's own first-extremum positions."""
    outs, start, own = [], 0, []
    for b, n in enumerate(batch_num_nodes):
        seg = x[start:start + n]
        start += n
        if op in ('max', 'min') and capture is not None:
            own.append(first_argext(seg, op == 'max'))
        if op in ('max', 'min') and route is not None and op in route:
            outs.append(routed_ext(seg, route[op][b]))
            continue
        outs.append({'sum': lambda s: s.sum(0), 'mean': lambda s: s.mean(0),
                     'max': lambda s: s.max(0)[0], 'min': lambda s: s.min(0)[0]}[op](seg))
    if own:
        capture[op] = torch.stack(own, 0)
    return torch.stack(outs, 0)


# ------------------------------------------------------------------------------------------
# PNA: reference models/pna.py
# ------------------------------------------------------------------------------------------
def aggregate(h, name, route=None):
    """reference models/pna.py:17-50; h is a mailbox [n, D, F].  route (tests): [n, F] positions for max / min."""
    if name == 'mean':
        return torch.mean(h, dim=-2)
    if name == 'max':
        return routed_ext(h, route) if route is not None else torch.max(h, dim=-2)[0]
    if name == 'min':
        return routed_ext(h, route) if route is not None else torch.min(h, dim=-2)[0]
    if name == 'sum':
        return torch.sum(h, dim=-2)
    if name in ('var', 'std'):
        h_mean_squares = torch.mean(h * h, dim=-2)
        h_mean = torch.mean(h, dim=-2)
        var = torch.relu(h_mean_squares - h_mean * h_mean)
        return var if name == 'var' else torch.sqrt(var + EPS)
    raise ValueError(name)


def scale(h, name, D, avg_d_log=1.0):
    """reference models/pna.py:57-68.  D is a python int, np.log -> python float."""
    if name == 'identity':
        return h
    if name == 'amplification':
        return h * (math.log(D + 1) / avg_d_log)
    if name == 'attenuation':
        return h * (avg_d_log / math.log(D + 1))
    raise ValueError(name)


def pna_reduce(mailbox, D, aggregators, scalers, avg_d_log=1.0, route=None):
    """reference models/pna.py:221-235 (reduce_func).  route (tests): {'max': [n, F], 'min': [n, F]} mailbox positions."""
    h = torch.cat([aggregate(mailbox, a, route.get(a) if route is not None else None) for a in aggregators], dim=-1)
    if len(scalers) > 1:   # reference quirk: a single scaler is never applied (:232)
        h = torch.cat([scale(h, s, D, avg_d_log) for s in scalers], dim=-1)
    return h


def pna_layer(h, ef, src, dst, P, prefix, cfg, training, capture=None, route=None, x=None):
    """reference models/pna.py:199-252 (PNALayer.forward / pretrans_edges).  route (tests): {'max': [N, F], 'min': [N, F]}
    mailbox positions the max / min gradients of every node are routed to; capture receives this side's own choices."""
    n = h.shape[0]
    z = torch.cat([h[src], h[dst], ef], dim=-1)                          # :249
    if cfg.get('pairwise_distances', False):                             # :243-245: squared distance of the end points' coordinates
        z = torch.cat([z, torch.sum((x[src] - x[dst]) ** 2, dim=-1)[:, None]], dim=-1)
    e = mlp(z, P, f'{prefix}.pretrans', cfg['pretrans_layers'], cfg['activation'], cfg['last_activation'],
            cfg['mid_batch_norm'], cfg['last_batch_norm'], cfg['batch_norm_momentum'], training)   # :252
    n_out = len(cfg['aggregators']) * (len(cfg['scalers']) if len(cfg['scalers']) > 1 else 1) * h.shape[1]
    own = {'max': torch.zeros(n, e.shape[1], dtype=torch.long), 'min': torch.zeros(n, e.shape[1], dtype=torch.long)}

    def reduce_fn(mb, D, nodes):
        if capture is not None:
            own['max'][nodes] = first_argext(mb.detach(), True)
            own['min'][nodes] = first_argext(mb.detach(), False)
        r = {k: v[nodes].clamp(max=D - 1) for k, v in route.items()} if route is not None else None
        return pna_reduce(mb, D, cfg['aggregators'], cfg['scalers'], route=r)
    agg = degree_bucketed_reduce(e, dst, n, reduce_fn, n_out, with_nodes=True)  # :206
    if capture is not None:
        capture['e'] = e
        capture['agg'] = agg
        capture['argext'] = own
    hcat = torch.cat([h, agg], dim=-1)                                    # :207
    out = mlp(hcat, P, f'{prefix}.posttrans', cfg['posttrans_layers'], cfg['activation'], cfg['last_activation'],
              cfg['mid_batch_norm'], cfg['last_batch_norm'], cfg['batch_norm_momentum'], training)   # :209
    if cfg.get('residual', True):
        out = out + h                                                     # :210-211
    return out


def pna_forward(graph, P, cfg, training=True, capture=None, route=None):
    """reference models/pna.py:131-135 (PNA.forward) and :161-166 (PNAGNN.forward).

    graph: dict(src, dst [E] int64; atom_feat [N,9]; bond_feat [E,3]; batch_num_nodes list[int]).
    Returns (out [B,target_dim], node embeddings [N,F])."""
    h = embedding_sum(graph['atom_feat'], P, 'node_gnn.atom_encoder.atom_embedding_list', graph['atom_feat'].shape[1])
    ef = embedding_sum(graph['bond_feat'], P, 'node_gnn.bond_encoder.bond_embedding_list', graph['bond_feat'].shape[1])
    for l in range(cfg['propagation_depth']):
        cap = None
        if capture is not None:
            cap = capture.setdefault(f'layer{l}', {})
        h = pna_layer(h, ef, graph['src'], graph['dst'], P, f'node_gnn.mp_layers.{l}', cfg, training, cap,
                      route.get(f'layer{l}') if route is not None else None, x=graph.get('x'))
    rcap = capture.setdefault('readout', {}) if capture is not None else None
    r = torch.cat([segment_readout(h, graph['batch_num_nodes'], op, route.get('readout') if route is not None else None, rcap)
                   for op in cfg['readout_aggregators']], dim=-1)
    out = mlp(r, P, 'output', cfg.get('readout_layers', 2), 'relu', 'none', cfg.get('readout_batchnorm', True), False,
              cfg['batch_norm_momentum'], training)                       # :127-129
    return out, h


def pna_config(**kw):
    """Defaults of reference models/pna.py:95-114 overlaid with kw (e.g. the yml's model_parameters)."""
    cfg = dict(readout_batchnorm=True, readout_layers=2, residual=True, activation='relu', last_activation='none',
               mid_batch_norm=False, last_batch_norm=False, propagation_depth=5, posttrans_layers=1,
               pretrans_layers=1, batch_norm_momentum=0.1)
    cfg.update(kw)
    return cfg


# ------------------------------------------------------------------------------------------
# Net3D: reference models/net3d.py, commons/utils.py:103-110
# ------------------------------------------------------------------------------------------
def fourier_encode_dist(x, num_encodings=4):
    """reference commons/utils.py:103-110 (include_self=True)."""
    x = x.unsqueeze(-1)
    orig = x
    scales = 2 ** torch.arange(num_encodings, dtype=x.dtype)
    x = x / scales
    x = torch.cat([x.sin(), x.cos()], dim=-1)
    x = torch.cat((x, orig), dim=-1)
    return x.squeeze()


def net3d_config(**kw):
    """Defaults of reference models/net3d.py:15-18 overlaid with kw."""
    cfg = dict(batch_norm=False, node_wise_output_layers=2, readout_batchnorm=True, batch_norm_momentum=0.1,
               reduce_func='sum', propagation_depth=4, readout_layers=2, readout_hidden_dim=None, fourier_encodings=0,
               activation='SiLU', update_net_layers=2, message_net_layers=2)
    cfg.update(kw)
    return cfg


def net3d_forward(graph, P, cfg, training=True):
    """reference models/net3d.py:57-75 (Net3D.forward) with Net3DLayer :108-125.

    graph: dict(src, dst [E3] int64; d [E3,1] fp32; num_nodes; batch_num_nodes)."""
    n = graph['num_nodes']
    src, dst = graph['src'], graph['dst']
    mom, bn, act = cfg['batch_norm_momentum'], cfg['batch_norm'], cfg['activation']
    h = P['node_embedding'][None, :].expand(n, -1)                                           # :61
    d = graph['d']
    if cfg['fourier_encodings'] > 0:
        d = fourier_encode_dist(d, cfg['fourier_encodings'])                                 # :63-64
    d = F.silu(mlp(d, P, 'edge_input', 1, act, act, bn, bn, mom, training))                  # :80-81
    deg = torch.bincount(dst, minlength=n)
    for l in range(cfg['propagation_depth']):
        pre = f'mp_layers.{l}'
        m_in = torch.cat([h[src], h[dst], d], dim=-1)                                        # :113-114
        m = mlp(m_in, P, f'{pre}.message_network', cfg['message_net_layers'], act, act, bn, bn, mom, training)
        d = d + m                                                                            # :116
        w = torch.sigmoid(F.linear(m, P[f'{pre}.soft_edge_network.weight'], P[f'{pre}.soft_edge_network.bias']))
        msg = m * w                                                                          # :117-118
        m_sum = torch.zeros(n, msg.shape[1], dtype=msg.dtype).index_add(0, dst, msg)
        if cfg['reduce_func'] == 'mean':                                                     # :95-96 (fn.mean)
            m_sum = m_sum / deg.clamp(min=1).to(msg.dtype)[:, None]
        h_new = mlp(m_sum + h, P, f'{pre}.update_network', cfg['update_net_layers'], act, 'None', bn, bn, mom,
                    training)                                                                # :120-125
        h = h_new + h
    if cfg['node_wise_output_layers'] > 0:                                                   # :70-71
        h = mlp(h, P, 'node_wise_output_network', cfg['node_wise_output_layers'], act, 'None', bn, bn, mom, training)
    r = torch.cat([segment_readout(h, graph['batch_num_nodes'], op) for op in cfg['readout_aggregators']], dim=-1)
    out = mlp(r, P, 'output', cfg['readout_layers'], 'relu', 'none', cfg['readout_batchnorm'], False, mom, training)
    return out, h


# ------------------------------------------------------------------------------------------
# losses: reference commons/losses.py
# ------------------------------------------------------------------------------------------
def ntxent(z1, z2, tau=0.5, norm=True):
    """reference commons/losses.py:143-155 (NTXent.forward, regularisers off)."""
    sim = torch.einsum('ik,jk->ij', z1, z2)
    if norm:
        sim = sim / (torch.einsum('i,j->ij', z1.norm(dim=1), z2.norm(dim=1)) + 1e-8)
    sim = torch.exp(sim / tau)
    pos = torch.diagonal(sim)
    return -torch.log(pos / (sim.sum(dim=1) - pos)).mean()


def ntxent_multiple_positives(z1, z2, tau=0.5, norm=True):
    """reference commons/losses.py:225-247 (NTXentMultiplePositives.forward, regularisers off)."""
    b, dim = z1.shape
    z2 = z2.view(b, -1, dim)
    sim = torch.einsum('ik,juk->iju', z1, z2)
    if norm:
        sim = sim / torch.einsum('i,ju->iju', z1.norm(dim=1), z2.norm(dim=2))   # no epsilon (:239)
    sim = torch.exp(sim / tau).sum(dim=2)
    pos = torch.diagonal(sim)
    return -torch.log(pos / (sim.sum(dim=1) - pos)).mean()


# ------------------------------------------------------------------------------------------
# parameter construction (reference init): models/base_layers.py:89,93-98, commons/mol_encoder.py:26-27,
# models/net3d.py:31-32
# ------------------------------------------------------------------------------------------
def _fc_params(P, prefix, in_dim, out_dim, batch_norm, gen):
    w = torch.empty(out_dim, in_dim)
    torch.nn.init.xavier_uniform_(w, gain=1 / in_dim, generator=gen)
    P[prefix + '.linear.weight'] = w
    P[prefix + '.linear.bias'] = torch.zeros(out_dim)
    if batch_norm:
        P[prefix + '.batch_norm.weight'] = torch.ones(out_dim)
        P[prefix + '.batch_norm.bias'] = torch.zeros(out_dim)
        P[prefix + '.batch_norm.running_mean'] = torch.zeros(out_dim)
        P[prefix + '.batch_norm.running_var'] = torch.ones(out_dim)
        P[prefix + '.batch_norm.num_batches_tracked'] = torch.zeros((), dtype=torch.long)


def _mlp_params(P, prefix, in_dim, hidden, out_dim, layers, mid_bn, last_bn, gen):
    if layers <= 1:
        _fc_params(P, f'{prefix}.fully_connected.0', in_dim, out_dim, last_bn, gen)
        return
    _fc_params(P, f'{prefix}.fully_connected.0', in_dim, hidden, mid_bn, gen)
    for l in range(1, layers - 1):
        _fc_params(P, f'{prefix}.fully_connected.{l}', hidden, hidden, mid_bn, gen)
    _fc_params(P, f'{prefix}.fully_connected.{layers - 1}', hidden, out_dim, last_bn, gen)


def init_pna_params(cfg, seed=0) -> Dict[str, torch.Tensor]:
    """Parameter dict with the reference's state_dict keys/shapes and init distributions (not its RNG stream)."""
    gen = torch.Generator().manual_seed(seed)
    P: Dict[str, torch.Tensor] = {}
    Fh = cfg['hidden_dim']
    n_agg = len(cfg['aggregators']) * (len(cfg['scalers']) if len(cfg['scalers']) > 1 else 1)
    # NB reference quirk: posttrans in_dim uses len(aggregators)*len(scalers)+1 (models/pna.py:193)
    post_in = (len(cfg['aggregators']) * len(cfg['scalers']) + 1) * Fh
    assert post_in == (n_agg + 1) * Fh or len(cfg['scalers']) == 1
    for l in range(cfg['propagation_depth']):
        pre = f'node_gnn.mp_layers.{l}'
        _mlp_params(P, f'{pre}.pretrans', 3 * Fh, Fh, Fh, cfg['pretrans_layers'], cfg['mid_batch_norm'],
                    cfg['last_batch_norm'], gen)
        _mlp_params(P, f'{pre}.posttrans', post_in, Fh, Fh, cfg['posttrans_layers'], cfg['mid_batch_norm'],
                    cfg['last_batch_norm'], gen)
    for k, dim in enumerate(ATOM_FEATURE_DIMS):
        w = torch.empty(dim, Fh)
        torch.nn.init.xavier_uniform_(w, generator=gen)
        P[f'node_gnn.atom_encoder.atom_embedding_list.{k}.weight'] = w
    for k, dim in enumerate(BOND_FEATURE_DIMS):
        w = torch.empty(dim, Fh)
        torch.nn.init.xavier_uniform_(w, generator=gen)
        P[f'node_gnn.bond_encoder.bond_embedding_list.{k}.weight'] = w
    rh = cfg.get('readout_hidden_dim') or Fh
    _mlp_params(P, 'output', Fh * len(cfg['readout_aggregators']), rh, cfg['target_dim'], cfg.get('readout_layers', 2),
                cfg.get('readout_batchnorm', True), False, gen)
    return P


def init_net3d_params(cfg, seed=0) -> Dict[str, torch.Tensor]:
    gen = torch.Generator().manual_seed(seed)
    P: Dict[str, torch.Tensor] = {}
    H, bn = cfg['hidden_dim'], cfg['batch_norm']
    edge_in = 1 if cfg['fourier_encodings'] == 0 else 2 * cfg['fourier_encodings'] + 1
    _mlp_params(P, 'edge_input', edge_in, H, H, 1, bn, bn, gen)
    P['node_embedding'] = torch.randn(H, generator=gen)
    for l in range(cfg['propagation_depth']):
        _mlp_params(P, f'mp_layers.{l}.message_network', 3 * H, H, H, cfg['message_net_layers'], bn, bn, gen)
        _mlp_params(P, f'mp_layers.{l}.update_network', H, H, H, cfg['update_net_layers'], bn, bn, gen)
        bound = 1 / math.sqrt(H)
        P[f'mp_layers.{l}.soft_edge_network.weight'] = (torch.rand(1, H, generator=gen) * 2 - 1) * bound
        P[f'mp_layers.{l}.soft_edge_network.bias'] = (torch.rand(1, generator=gen) * 2 - 1) * bound
    if cfg['node_wise_output_layers'] > 0:
        _mlp_params(P, 'node_wise_output_network', H, H, H, cfg['node_wise_output_layers'], bn, bn, gen)
    rh = cfg.get('readout_hidden_dim') or H
    _mlp_params(P, 'output', H * len(cfg['readout_aggregators']), rh, cfg['target_dim'], cfg['readout_layers'],
                cfg['readout_batchnorm'], False, gen)
    return P


def trainable(P: Dict[str, torch.Tensor]) -> List[str]:
    """Names of the entries that are nn.Parameters in the reference (everything but BN buffers)."""
    return [k for k in P if not (k.endswith('running_mean') or k.endswith('running_var')
                                 or k.endswith('num_batches_tracked'))]


def require_grad(P):
    for k in trainable(P):
        P[k] = P[k].detach().clone().requires_grad_(True)
    return P


# ------------------------------------------------------------------------------------------
# graph dict helpers + one full training step (the unit bench.py's cpu_baseline times)
# ------------------------------------------------------------------------------------------
def graphs_from_molecules(mols, coords_list=None):
    """Batch `synth.Molecule`s into the two graph dicts the oracle forwards consume
    (block-diagonal batching, DGL `dgl.batch` semantics; complete-graph edge order of
    reference datasets/qm9_dataset.py:215-217)."""
    import numpy as np
    srcs, dsts, af, bf, s3, d3, dd, bnn = [], [], [], [], [], [], [], []
    off = 0
    for i, m in enumerate(mols):
        srcs.append(m.src + off)
        dsts.append(m.dst + off)
        af.append(m.atom_feat)
        bf.append(m.bond_feat)
        n = m.n_atoms
        ar = np.arange(n)
        a = np.repeat(ar, n - 1)
        b = np.concatenate([np.concatenate([ar[:j], ar[j + 1:]]) for j in range(n)]) if n > 1 else ar[:0]
        xyz = m.coords if coords_list is None else coords_list[i]
        diff = (xyz[a] - xyz[b]).astype(np.float32)
        dd.append(np.sqrt((diff ** 2).sum(-1, dtype=np.float32))[:, None].astype(np.float32))
        s3.append(a + off)
        d3.append(b + off)
        bnn.append(n)
        off += n
    t = lambda x, dt: torch.from_numpy(np.concatenate(x)).to(dt)
    g2 = dict(src=t(srcs, torch.long), dst=t(dsts, torch.long), atom_feat=t(af, torch.long),
              bond_feat=t(bf, torch.long), batch_num_nodes=bnn, num_nodes=off,
              x=torch.from_numpy(np.concatenate([(m.coords if coords_list is None else coords_list[i]).astype(np.float32)
                                                 for i, m in enumerate(mols)])))      # ndata['x'] (pairwise_distances=True)
    g3 = dict(src=t(s3, torch.long), dst=t(d3, torch.long), d=t(dd, torch.float32), batch_num_nodes=bnn,
              num_nodes=off)
    return g2, g3


def train_step(g2, g3, P2, cfg2, P3, cfg3, optim, tau=0.1):
    """One pre-training step: reference trainer/self_supervised_trainer.py:24-29 +
    trainer/trainer.py:116-124 (forward both nets, NT-Xent, backward, Adam step, zero_grad)."""
    z2, _ = pna_forward(g2, P2, cfg2, training=True)
    z3, _ = net3d_forward(g3, P3, cfg3, training=True)
    loss = ntxent(z2, z3, tau=tau)
    loss.backward()
    optim.step()
    optim.zero_grad()
    return loss.detach()


# ------------------------------------------------------------------------------------------
# Original-PNA variants: reference models/pna_original.py
# ------------------------------------------------------------------------------------------
def mlp_readout(x, P, prefix, L=2):
    """reference models/base_layers.py:149-164 (MLPReadout: L halving Linear+ReLU layers, then Linear)."""
    for l in range(L):
        x = F.relu(F.linear(x, P[f'{prefix}.FC_layers.{l}.weight'], P[f'{prefix}.FC_layers.{l}.bias']))
    return F.linear(x, P[f'{prefix}.FC_layers.{L}.weight'], P[f'{prefix}.FC_layers.{L}.bias'])


def pna_original_reduce(mailbox, D, aggregators, scalers, avg_d):
    """reference models/pna_original.py:231-236 / 418-423: scalers are ALWAYS applied, with the real avg_d."""
    h = torch.cat([aggregate(mailbox, a) for a in aggregators], dim=1)
    return torch.cat([scale(h, s, D, avg_d) for s in scalers], dim=1)


def pna_original_forward(graph, snorm_n, P, cfg, training=True):
    """reference models/pna_original.py:138-146 (PNAOriginal.forward), :181-194 (PNAGNNOriginal.forward),
    :241-261 (PNATower.forward), :296-312 (PNALayer.forward).  gru_enable / use_3d / dropout are off."""
    src, dst = graph['src'], graph['dst']
    n = graph['num_nodes']
    mom = 0.1   # MLP default batch_norm_momentum: pna_original never passes one
    h = embedding_sum(graph['atom_feat'], P, 'node_gnn.embedding_h.atom_embedding_list', graph['atom_feat'].shape[1])
    e = None
    if cfg.get('edge_feat', True):
        e = embedding_sum(graph['bond_feat'], P, 'node_gnn.embedding_e.bond_embedding_list', graph['bond_feat'].shape[1])
    L, towers = cfg['propagation_depth'], cfg.get('towers', 1)
    n_blocks = len(cfg['aggregators']) * len(cfg['scalers'])
    for l in range(L):
        last = l == L - 1
        in_dim = cfg['hidden_dim']
        out_dim = cfg['last_layer_dim'] if last else cfg['hidden_dim']
        divide = cfg.get('divide_input_last', True) if last else cfg.get('divide_input_first', True)
        it = in_dim // towers if divide else in_dim
        outs = []
        for t in range(towers):
            pre = f'node_gnn.layers.{l}.towers.{t}'
            ht = h[:, t * it:(t + 1) * it] if divide else h
            z = torch.cat([ht[src], ht[dst]] + ([e] if e is not None else []), dim=1)              # :221-225
            if cfg.get('use_3d', False):                                                           # :224-226
                z = torch.cat([z, torch.norm(graph['x'][src] - graph['x'][dst], dim=-1)[:, None]], dim=1)
            msg = mlp(z, P, f'{pre}.pretrans', cfg['pretrans_layers'], 'relu', 'none', False, False, mom, training)
            agg = degree_bucketed_reduce(msg, dst, n, lambda mb, D: pna_original_reduce(
                mb, D, cfg['aggregators'], cfg['scalers'], cfg['avg_d']), n_blocks * it)            # :250
            ho = mlp(torch.cat([ht, agg], dim=1), P, f'{pre}.posttrans', cfg['posttrans_layers'], 'relu', 'none',
                     cfg['mid_batch_norm'], cfg['last_batch_norm'], mom, training)                 # :251-254
            if cfg.get('graph_norm', False):
                ho = ho * snorm_n                                                                   # :257-258
            outs.append(ho)
        h_cat = torch.cat(outs, dim=1)
        h_out = F.leaky_relu(F.linear(h_cat, P[f'node_gnn.layers.{l}.mixing_network.weight'],
                                      P[f'node_gnn.layers.{l}.mixing_network.bias']))               # :308
        if cfg.get('residual', False) and in_dim == out_dim:
            h_out = h + h_out
        if cfg.get('gru_enable', False) and not last:                                               # :190-193, :64-84
            h_out = gru_step(h, h_out, P, 'node_gnn.gru.gru')
        h = h_out
    r = torch.cat([segment_readout(h, graph['batch_num_nodes'], op) for op in cfg['readout_aggregators']], dim=-1)
    return mlp_readout(r, P, 'output'), h


def gru_step(x, h0, P, prefix):
    """one step of torch.nn.GRU (gate order r | z | n), reference models/pna_original.py:64-84: input x, hidden state h0"""
    gi = F.linear(x, P[f'{prefix}.weight_ih_l0'], P[f'{prefix}.bias_ih_l0'])
    gh = F.linear(h0, P[f'{prefix}.weight_hh_l0'], P[f'{prefix}.bias_hh_l0'])
    H = h0.shape[1]
    r = torch.sigmoid(gi[:, :H] + gh[:, :H])
    z = torch.sigmoid(gi[:, H:2 * H] + gh[:, H:2 * H])
    n = torch.tanh(gi[:, 2 * H:] + r * gh[:, 2 * H:])
    return (1 - z) * n + z * h0


def pna_original_simple_forward(graph, P, cfg, training=True):
    """reference models/pna_original.py:341-348, :374-382, :425-444 (PNASimpleLayer: copy_u messages, no edge MLP,
    ReLU after posttrans)."""
    src, dst = graph['src'], graph['dst']
    n = graph['num_nodes']
    mom = 0.1
    h = embedding_sum(graph['atom_feat'], P, 'node_gnn.embedding_h.atom_embedding_list', graph['atom_feat'].shape[1])
    L = cfg['propagation_depth']
    n_blocks = len(cfg['aggregators']) * len(cfg['scalers'])
    for l in range(L):
        in_dim = cfg['hidden_dim']
        out_dim = cfg['last_layer_dim'] if l == L - 1 else cfg['hidden_dim']
        agg = degree_bucketed_reduce(h[src], dst, n, lambda mb, D: pna_original_reduce(
            mb, D, cfg['aggregators'], cfg['scalers'], cfg['avg_d']), n_blocks * in_dim)
        ho = mlp(agg, P, f'node_gnn.layers.{l}.posttrans', cfg['posttrans_layers'], 'relu', 'none',
                 cfg['mid_batch_norm'], cfg['last_batch_norm'], mom, training)
        ho = F.relu(ho)
        if cfg.get('residual', False) and in_dim == out_dim:
            ho = h + ho
        h = ho
    r = torch.cat([segment_readout(h, graph['batch_num_nodes'], op) for op in cfg['readout_aggregators']], dim=-1)
    out = mlp(r, P, 'output', cfg['readout_layers'], 'relu', 'none', cfg['readout_batchnorm'], False,
              cfg['batch_norm_momentum'], training)
    return out, h


def snorm_n(batch_num_nodes):
    """reference datasets/custom_collate.py:46-47: per-node 1/sqrt(graph size), [N,1]."""
    return torch.cat([torch.full((n, 1), 1.0 / float(n)) for n in batch_num_nodes]).sqrt()
