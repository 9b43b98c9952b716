"""Row f1: the vectorised batch assembly (3dinfomax_amd/dataset.py + csrc/batch.hip) must produce exactly what
batching per-molecule graphs produces (graph.batch = dgl.batch semantics): bit-exact integer arrays, distances to 1 ulp."""
import importlib

import numpy as np
import pytest
import torch

from helpers import synth

amd = importlib.import_module('3dinfomax_amd')
dataset = importlib.import_module('3dinfomax_amd.dataset')

INDEX_FIELDS = ('in_ptr', 'perm', 'src_s', 'dst_s', 'out_ptr', 'out_epos', 'graph_ptr', 'inv_perm')


def _check_index(a, b):
    assert (a.num_nodes, a.num_edges, a.num_graphs) == (b.num_nodes, b.num_edges, b.num_graphs)
    for f in INDEX_FIELDS:
        assert torch.equal(getattr(a, f).cpu(), getattr(b, f).cpu()), f


def test_bond_graph_assembly_equals_per_molecule_batching():
    mols = synth.make_dataset(200, seed=12) + synth.make_dataset(20, seed=3, kind='qmugs')
    ds = dataset.FlatMolDataset(mols)
    rng = np.random.default_rng(0)
    for _ in range(3):
        ids = rng.permutation(len(mols))[:64]
        g2, xyz, gp, n, bnn = ds.assemble_2d(ids, 'cpu')
        ref = amd.batch([amd.bond_graph(mols[i]) for i in ids])
        assert torch.equal(g2.edges()[0], ref.edges()[0]) and torch.equal(g2.edges()[1], ref.edges()[1])
        assert torch.equal(g2.ndata['feat'], ref.ndata['feat']) and torch.equal(g2.edata['feat'], ref.edata['feat'])
        assert torch.equal(g2.batch_num_nodes(), ref.batch_num_nodes())
        _check_index(g2.index(), ref.index())
        assert np.array_equal(xyz.numpy(), np.concatenate([mols[i].coords for i in ids]))


@pytest.mark.gpu
def test_complete_graphs_built_on_device_equal_reference_construction():
    assert torch.cuda.is_available()
    mols = synth.make_dataset(96, seed=5) + synth.make_dataset(8, seed=9, kind='qmugs')
    ds = dataset.FlatMolDataset(mols)
    ids = np.random.default_rng(1).permutation(len(mols))[:80]
    (g2,), (g3,) = ds.assemble(ids, torch.device('cuda:0'))
    ref3 = amd.batch([amd.complete_graph(mols[i]) for i in ids])
    assert torch.equal(g3.edges()[0].cpu(), ref3.edges()[0]) and torch.equal(g3.edges()[1].cpu(), ref3.edges()[1])
    _check_index(g3.index(), ref3.index())
    d, dref = g3.edata['d'].cpu(), ref3.edata['d']
    assert d.shape == dref.shape
    assert torch.allclose(d, dref, rtol=3e-7, atol=1e-7)         # same fp32 formula, 1 ulp
    ref2 = amd.batch([amd.bond_graph(mols[i]) for i in ids])
    _check_index(g2.index(), ref2.index())
    # and the models consume the assembled batch unchanged
    pna = amd.PNA(hidden_dim=16, target_dim=8, aggregators=['mean', 'max', 'min', 'std'],
                  scalers=['identity', 'amplification', 'attenuation'], readout_aggregators=['min', 'max', 'mean'],
                  propagation_depth=2, mid_batch_norm=True, last_batch_norm=True, pretrans_layers=2).cuda()
    net = amd.Net3D(node_dim=0, edge_dim=1, hidden_dim=20, target_dim=8, readout_aggregators=['min', 'max', 'mean'],
                    batch_norm=True, node_wise_output_layers=0, reduce_func='mean', fourier_encodings=4,
                    propagation_depth=1, readout_layers=1, update_net_layers=1, message_net_layers=1).cuda()
    za, zb = pna(g2), net(g3)
    g2r, g3r = ref2.to('cuda:0'), ref3.to('cuda:0')
    assert torch.equal(za, pna(g2r))
    assert torch.allclose(zb, net(g3r), rtol=1e-5, atol=1e-6)


def test_batch_stream_through_dataloader_workers_matches_direct_assembly():
    """BatchStream: the numpy half of the assembly in DataLoader worker processes; the batches are the seeded sequence,
    bit for bit what FlatMolDataset.assemble_host gives in-process, and the device half accepts them."""
    mols = synth.make_dataset(90, seed=4)
    ds = dataset.FlatMolDataset(mols)
    stream = dataset.BatchStream(ds, batch_size=16, steps=7, seed=3)
    loader = torch.utils.data.DataLoader(stream, batch_size=None, num_workers=2, prefetch_factor=2)
    got = list(loader)
    assert len(got) == 7
    for i, hb in enumerate(got):
        ref = stream[i]
        norm = lambda v: [list(x) if isinstance(x, (list, tuple)) else x for x in v]
        assert list(hb['dims']) == list(ref['dims']) and norm(hb['cuts']) == norm(ref['cuts']) and norm(hb['groups']) == norm(ref['groups'])
        for k in ('i32', 'i64', 'f32', 'n', 'rows', 'tiles'):
            assert torch.equal(hb[k], ref[k]), (i, k)
    # batches 0.. of one epoch partition a permutation; the next epoch reshuffles
    first_epoch = torch.cat([got[i]['n'] for i in range(5)])
    assert first_epoch.numel() == 80
    g2, xyz, gp, n, bnn = dataset.host_batch_to_device(got[0], 'cpu')
    direct = ds.assemble_2d(np.random.default_rng(3).permutation(90)[:16], 'cpu')[0]
    assert torch.equal(g2.ndata['feat'], direct.ndata['feat']) and torch.equal(g2.index().in_ptr, direct.index().in_ptr)
