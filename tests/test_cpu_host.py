"""CPU-side tests (no GPU): the C-ABI library loads and exports every declared symbol, the plugin surface keeps
the reference's constructor kwargs / state_dict keys, the graph index and the synthetic generators are right,
and the product path refuses to run without the HIP kernels (no silent fallback)."""
import importlib
import os

import numpy as np
import pytest
import torch

from helpers import NET3D_SMALL, NET3D_YML, PNA_SMALL, PNA_YML, load, mols_from_npz, sd_from_npz, synth

amd = importlib.import_module('3dinfomax_amd')
L = importlib.import_module('3dinfomax_amd._lib')
graph = importlib.import_module('3dinfomax_amd.graph')


def test_build_entry_compiles_and_library_exports_every_declared_symbol():
    import __graft_entry__ as ge
    ge.build()
    lib = L.load()
    declared = L.declared_symbols()
    assert len(declared) >= 30
    for name in declared:
        assert hasattr(lib, name), f'{name} declared in include/infomax3d_hip.h but not exported'
    assert set(declared) == set(L._SIGNATURES), 'ctypes signature table out of sync with the header'
    assert lib.i3d_abi_version() == 1
    # argument validation happens on the host before any launch: no GPU needed
    rc = lib.i3d_gemm_f32(0, 0, -1, 4, 4, None, 4, None, 4, None, 4, None, 0, None)
    assert rc == -1 and b'negative dimension' in lib.i3d_last_error()


def test_state_dict_surface_matches_reference():
    z = load('models_small.npz')
    pna = amd.PNA(avg_d=1.0, device='cpu', **PNA_SMALL)
    net = amd.Net3D(node_dim=0, edge_dim=1, avg_d=1.0, **NET3D_SMALL)
    ref2, ref3 = sd_from_npz(z, 'init/pna_sd'), sd_from_npz(z, 'init/net3d_sd')
    assert list(pna.state_dict().keys()) == list(ref2.keys())
    assert list(net.state_dict().keys()) == list(ref3.keys())
    for k, v in pna.state_dict().items():
        assert tuple(v.shape) == tuple(ref2[k].shape), k
    for k, v in net.state_dict().items():
        assert tuple(v.shape) == tuple(ref3[k].shape), k
    pna.load_state_dict(ref2, strict=True)
    net.load_state_dict(ref3, strict=True)
    # full yml size: parameter counts measured on the reference classes
    full = amd.PNA(avg_d=1.0, device='cpu', **PNA_YML)
    assert sum(p.numel() for p in full.parameters()) == 4982056
    full3 = amd.Net3D(node_dim=0, edge_dim=1, avg_d=1.0, **NET3D_YML)
    assert sum(p.numel() for p in full3.parameters()) == 17617
    # optimizer grouping relies on the 'batch_norm' substring (reference trainer/self_supervised_trainer.py:79-82)
    assert any('batch_norm' in k for k, _ in full.named_parameters())
    # init distribution of FCLayer: xavier_uniform with gain 1/in_dim (reference models/base_layers.py:93-98)
    w = full.node_gnn.mp_layers[0].posttrans.fully_connected[0].linear.weight
    bound = (1 / 2600) * (6 / (2600 + 200)) ** 0.5
    assert w.abs().max().item() <= bound * 1.0001 and w.abs().max().item() > 0.9 * bound
    import inspect
    assert 'class PNA' in inspect.getsource(type(full))       # trainer/trainer.py:266-270 snapshots the source


def test_plugin_alias_and_star_import():
    ns = {}
    exec('from infomax3d_amd import *', ns)
    for name in ('PNA', 'Net3D', 'NTXent', 'NTXentMultiplePositives', 'contrastive_collate', 'conformer_collate',
                 'PNALayer', 'PNAGNN', 'PNA_AGGREGATORS', 'PNA_SCALERS'):
        assert name in ns, name
    loss = ns['NTXent'](tau=0.1, norm=True, uniformity_reg=0, variance_reg=0, covariance_reg=0)
    assert loss.tau == 0.1


def test_graph_index_against_bruteforce():
    rng = np.random.default_rng(0)
    n, E = 40, 150
    src, dst = rng.integers(0, n, E), rng.integers(0, n, E)
    dst[dst == 7] = 8                      # node 7 has no in-edges
    g = graph.BatchedMolGraph(torch.from_numpy(src), torch.from_numpy(dst), n, torch.tensor([15, 25]))
    idx = g.index()
    perm = idx.perm.numpy()
    assert idx.num_nodes == n and idx.num_edges == E and idx.num_graphs == 2
    assert np.all(np.diff(dst[perm]) >= 0)
    for v in range(n):
        seg = perm[idx.in_ptr[v]:idx.in_ptr[v + 1]]
        assert np.array_equal(seg, np.nonzero(dst == v)[0])            # stable: edge-id order inside a mailbox
        out = idx.out_epos.numpy()[idx.out_ptr[v]:idx.out_ptr[v + 1]]
        assert np.array_equal(np.sort(perm[out]), np.nonzero(src == v)[0])
    assert np.array_equal(idx.src_s.numpy(), src[perm]) and np.array_equal(idx.dst_s.numpy(), dst[perm])
    assert np.array_equal(idx.inv_perm.numpy()[perm], np.arange(E))
    assert idx.graph_ptr.tolist() == [0, 15, 40]
    assert idx.in_ptr[8] == idx.in_ptr[7]


def test_batching_matches_dgl_semantics_and_reference_edge_order():
    mols = synth.make_dataset(5, seed=9)
    g2 = amd.batch([amd.bond_graph(m) for m in mols])
    g3 = amd.batch([amd.complete_graph(m) for m in mols])
    off = np.cumsum([0] + [m.n_atoms for m in mols])
    assert g2.batch_num_nodes().tolist() == [m.n_atoms for m in mols]
    s, d = g2.edges()
    assert np.array_equal(s.numpy(), np.concatenate([m.src + o for m, o in zip(mols, off)]))
    assert np.array_equal(d.numpy(), np.concatenate([m.dst + o for m, o in zip(mols, off)]))
    assert g2.ndata['feat'].dtype == torch.int64 and g2.ndata['feat'].shape == (off[-1], 9)
    assert g2.edata['feat'].shape[1] == 3
    # complete graph: src = repeat_interleave(arange(n), n-1) (reference datasets/qm9_dataset.py:215-217)
    n0 = mols[0].n_atoms
    s3, d3 = g3.edges()
    assert np.array_equal(s3[:n0 * (n0 - 1)].numpy(), np.repeat(np.arange(n0), n0 - 1))
    assert g3.edata['d'].shape == (sum(m.n_atoms * (m.n_atoms - 1) for m in mols), 1)
    a, b = amd.contrastive_collate([(amd.bond_graph(m), amd.complete_graph(m)) for m in mols])
    assert a[0].number_of_nodes() == b[0].number_of_nodes() == off[-1]


def test_synthetic_molecules_are_qm9_shaped_and_seeded():
    a = synth.make_dataset(300, seed=4)
    b = synth.make_dataset(300, seed=4)
    assert all(np.array_equal(x.src, y.src) and np.array_equal(x.coords, y.coords) for x, y in zip(a, b))
    n_atoms = np.array([m.n_atoms for m in a])
    assert 14 < n_atoms.mean() < 22 and n_atoms.max() <= 30
    for m in a[:50]:
        deg = np.bincount(m.dst, minlength=m.n_atoms)
        assert deg.min() >= 1 and deg.max() <= 4                       # valence caps: degrees in 1..4
        assert np.array_equal(m.src[0::2], m.dst[1::2]) and np.array_equal(m.dst[0::2], m.src[1::2])
        assert np.array_equal(m.bond_feat[0::2], m.bond_feat[1::2])
        assert (m.atom_feat < np.array(synth.ATOM_FEATURE_DIMS)).all() and (m.atom_feat >= 0).all()
    q = synth.make_dataset(20, seed=1, kind='qmugs')
    assert max(np.bincount(m.dst).max() for m in q) <= 6


def test_product_path_has_no_cpu_fallback():
    """Without a GPU the plugin must fail loudly instead of computing on the CPU."""
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    mols = synth.make_dataset(2, seed=0)
    pna = amd.PNA(avg_d=1.0, device='cpu', **PNA_SMALL)
    g2 = amd.batch([amd.bond_graph(m) for m in mols])
    with pytest.raises((AssertionError, RuntimeError)):
        pna(g2)
    with pytest.raises((AssertionError, RuntimeError)):
        amd.NTXent(tau=0.1)(torch.randn(4, 8), torch.randn(4, 8))
