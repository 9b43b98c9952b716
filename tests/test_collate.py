"""the collate functions of the plugin surface against tests/golden/collate.npz, which was
produced by calling the reference's OWN datasets/custom_collate.py functions (tests/golden/gen_golden_host.py); and the
duck-typed `DGLGraph` input path (graph.as_batched_graph) driven by the stub DGL up to index construction."""
import importlib
import os
import sys

import numpy as np
import pytest
import torch

from helpers import GOLDEN, load, mols_from_npz, synth

amd = importlib.import_module('3dinfomax_amd')
G = importlib.import_module('3dinfomax_amd.graph')


def _items(z):
    mols = mols_from_npz(z)
    return mols, [amd.bond_graph(m) for m in mols], [amd.complete_graph(m) for m in mols]


def _same_graph(bg, z, tag):
    s, d = bg.edges()
    assert np.array_equal(s.numpy(), z[f'{tag}/src']) and np.array_equal(d.numpy(), z[f'{tag}/dst'])
    assert bg.number_of_nodes() == int(z[f'{tag}/n'])
    assert np.array_equal(bg.batch_num_nodes().numpy(), z[f'{tag}/bnn'])
    for k in [f for f in z.files if f.startswith(f'{tag}/ndata/')]:
        assert np.array_equal(bg.ndata[k.split('/')[-1]].numpy(), z[k]), k
    for k in [f for f in z.files if f.startswith(f'{tag}/edata/')]:
        a, b = bg.edata[k.split('/')[-1]].numpy(), z[k]
        assert a.shape == b.shape and np.allclose(a, b, rtol=0, atol=0), k


def test_graph_collate_matches_reference():
    z = load('collate.npz')
    mols, g2, _ = _items(z)
    t = [torch.from_numpy(r) for r in z['targets']]
    (bg,), targets = amd.graph_collate(list(zip(g2, t)))
    _same_graph(bg, z, 'graph_collate/g')
    assert targets.dtype == torch.float32 and np.array_equal(targets.numpy(), z['graph_collate/targets'])
    t1 = [torch.tensor(float(v)) for v in z['targets1']]
    (_,), targets1 = amd.graph_collate(list(zip(g2, t1)))
    assert tuple(targets1.shape) == tuple(z['graph_collate/targets1'].shape) == (len(mols), 1)
    assert np.array_equal(targets1.numpy(), z['graph_collate/targets1'])


def test_s_norm_collates_match_reference():
    z = load('collate.npz')
    mols, g2, g3 = _items(z)
    t = [torch.from_numpy(r) for r in z['targets']]
    (bg, snorm), targets = amd.s_norm_graph_collate(list(zip(g2, t)))
    _same_graph(bg, z, 's_norm_graph_collate/g')
    assert snorm.dtype == torch.float32 and np.array_equal(snorm.numpy(), z['s_norm_graph_collate/snorm_n'])
    assert np.array_equal(targets.numpy(), z['s_norm_graph_collate/targets'])
    (bg, snorm), (bg3,) = amd.s_norm_contrastive_collate(list(zip(g2, g3)))
    _same_graph(bg, z, 's_norm_contrastive_collate/g')
    _same_graph(bg3, z, 's_norm_contrastive_collate/g3')
    assert np.array_equal(snorm.numpy(), z['s_norm_contrastive_collate/snorm_n'])


def test_contrastive_and_conformer_collate_match_reference():
    z = load('collate.npz')
    mols, g2, g3 = _items(z)
    (bg,), (bg3,) = amd.contrastive_collate(list(zip(g2, g3)))
    _same_graph(bg, z, 'contrastive_collate/g')
    _same_graph(bg3, z, 'contrastive_collate/g3')
    t = [torch.from_numpy(r) for r in z['targets']]
    out = amd.contrastive_collate(list(zip(g2, g3, t)))
    assert len(out) == 3 and np.array_equal(out[2].numpy(), z['contrastive_collate/targets'])
    # conformers: the generator's noise (seed 5) on top of the molecule coordinates, three graphs per molecule
    rng = np.random.default_rng(5)
    confs = []
    for m, c0 in zip(mols, g3):
        cs = [c0]
        for _ in range(2):
            noise = rng.normal(0, 0.05, size=m.coords.shape).astype(np.float32)
            cs.append(amd.complete_graph(m, coords=m.coords + noise))
        confs.append(amd.batch(cs))
    (bg,), (bgc,) = amd.conformer_collate(list(zip(g2, confs)))
    _same_graph(bg, z, 'conformer_collate/g')
    _same_graph(bgc, z, 'conformer_collate/g3')
    assert bgc.batch_num_nodes().shape[0] == 3 * len(mols)          # flattened, DGL semantics


@pytest.fixture()
def stub_dgl():
    """the stand-in DGL of tests/golden/_stubs (restated DGL semantics) as `dgl`"""
    path = os.path.join(GOLDEN, '_stubs')
    sys.path.insert(0, path)
    saved = {k: sys.modules.pop(k) for k in list(sys.modules) if k == 'dgl' or k.startswith('dgl.')}
    try:
        yield importlib.import_module('dgl')
    finally:
        sys.path.remove(path)
        for k in [k for k in sys.modules if k == 'dgl' or k.startswith('dgl.')]:
            del sys.modules[k]
        sys.modules.update(saved)


def test_dgl_like_graphs_are_accepted_by_duck_typing(stub_dgl):
    """the reference's own datasets yield `dgl.DGLGraph`s; collates and models take them as they are."""
    dgl = stub_dgl
    mols = synth.make_dataset(4, seed=9)
    dg2, dg3 = [], []
    for m in mols:
        g = dgl.graph((torch.from_numpy(m.src), torch.from_numpy(m.dst)), num_nodes=m.n_atoms)
        g.ndata['feat'] = torch.from_numpy(m.atom_feat)
        g.edata['feat'] = torch.from_numpy(m.bond_feat)
        dg2.append(g)
        s, d = synth.complete_graph_edges(m.n_atoms)
        c = dgl.graph((torch.from_numpy(s), torch.from_numpy(d)), num_nodes=m.n_atoms)
        c.edata['d'] = torch.from_numpy(synth.pairwise_distances(m.coords, s, d))
        dg3.append(c)
    # (1) the rebinding `from infomax3d_amd import *` makes the plugin's collate see DGLGraph items
    (bg,), (bg3,) = amd.contrastive_collate(list(zip(dg2, dg3)))
    ref = amd.batch([amd.bond_graph(m) for m in mols])
    ref3 = amd.batch([amd.complete_graph(m) for m in mols])
    for a, b in ((bg, ref), (bg3, ref3)):
        assert torch.equal(a.edges()[0], b.edges()[0]) and torch.equal(a.edges()[1], b.edges()[1])
        assert torch.equal(a.batch_num_nodes(), b.batch_num_nodes())
        for k in a.ndata:
            assert torch.equal(a.ndata[k], b.ndata[k])
        assert set(a.edata) == set(b.edata)
        for k in a.edata:
            assert torch.equal(a.edata[k], b.edata[k])
    # (2) a graph batched by DGL itself (the reference's collate left in place) goes through as_batched_graph
    dbg = dgl.batch(dg2)
    wrapped = G.as_batched_graph(dbg)
    assert wrapped is G.as_batched_graph(dbg)                     # cached on the DGL object
    assert wrapped.ndata is dbg.ndata and wrapped.edata is dbg.edata      # frames shared: side effects reach the caller
    idx, ridx = wrapped.index(), ref.index()
    for f in ('in_ptr', 'perm', 'src_s', 'dst_s', 'out_ptr', 'out_epos', 'graph_ptr', 'inv_perm'):
        assert torch.equal(getattr(idx, f), getattr(ridx, f)), f
    assert idx.degree_groups()[2] == ridx.degree_groups()[2]
    wrapped.ndata['feat'] = torch.zeros(wrapped.number_of_nodes(), 3)      # what PNAGNN.forward does (models/pna.py:162)
    assert dbg.ndata['feat'].shape == (wrapped.number_of_nodes(), 3)
