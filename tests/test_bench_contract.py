"""bench.py's bookkeeping that needs no GPU: the PMC lookup behind `roofline.traffic` (a kernel
that gained a template parameter left the driver's line with `traffic: null`)."""
import importlib
import importlib.util
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
amd = importlib.import_module('3dinfomax_amd')
lk = importlib.import_module('tools.pmc_lookup')


def _bench():
    spec = importlib.util.spec_from_file_location('bench_module', os.path.join(ROOT, 'bench.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_kernel_names_match_on_base_name_and_leading_template_arguments():
    assert lk.split_kernel_name('pna_aggregate_bwd_kernel<4, 2, false>') == ('pna_aggregate_bwd_kernel', ('4', '2', 'false'))
    assert lk.split_kernel_name('wgrad_reduce_kernel') == ('wgrad_reduce_kernel', ())
    summary = {'k<2>': [dict(grid=256, dispatches=1, traffic_MB=1.0)],
               'k<2, false>': [dict(grid=512, dispatches=3, traffic_MB=2.0)],
               'k<2,true,7>': [dict(grid=512, dispatches=1, traffic_MB=4.0)],
               'k<20>': [dict(grid=512, dispatches=1, traffic_MB=100.0)],
               'kk<2>': [dict(grid=512, dispatches=1, traffic_MB=100.0)]}
    assert lk.traffic_bytes(summary, 'k', (2,)) == int(1e6 * (1.0 + 6.0 + 4.0) / 5)
    assert lk.traffic_bytes(summary, 'k', (2,), grids={512}) == int(1e6 * (6.0 + 4.0) / 4)
    assert lk.traffic_bytes(summary, 'k', (2, 'false')) == 2_000_000
    with pytest.raises(lk.PmcLookupError):
        lk.traffic_bytes(summary, 'k', (3,))
    with pytest.raises(lk.PmcNotCovered):           # the kernel is there, this workload's grid is not: another workload's pass
        lk.traffic_bytes(summary, 'k', (2,), grids={1024})
    with pytest.raises(lk.PmcLookupError):          # a kernel the file does not hold at all
        lk.traffic_bytes(summary, 'renamed_kernel', (2,), grids={512})


def test_committed_pmc_summary_has_rows_for_the_default_bench_workload():
    """The file bench.py reads (latest profiles/rNN_step_k4_pmc.json) must hold rows of the kernels the step launches on the
    batches bench.py builds (seeds 1000..1003, batch 512), within 5 % of the algorithmic bytes."""
    bench = _bench()
    path = bench.latest_pmc_summary()
    assert path is not None and os.path.exists(path)
    summary = lk.load(path)
    F = bench.PNA_KW['hidden_dim']
    for seed in range(1000, 1004):
        mols = amd.synth.make_dataset(512, seed=seed)
        N = sum(m.n_atoms for m in mols)
        E = sum(len(m.src) for m in mols)
        grids = {lk.k4_grid(N, F)}
        fwd = lk.traffic_bytes(summary, 'pna_aggregate_fwd_kernel', (2,), grids)
        bwd = lk.traffic_bytes(summary, 'pna_aggregate_bwd_kernel', (4, 2), grids)
        alg_fwd = 4.0 * E * F + 4.0 * N * 4 * F + 4.0 * (N + 1)
        alg_bwd = 4.0 * N * 4 * F + 2 * 4.0 * E * F + 4.0 * (N + 1)
        assert 0.95 < fwd / alg_fwd < 1.08, (seed, fwd, alg_fwd)
        assert 0.95 < bwd / alg_bwd < 1.08, (seed, bwd, alg_bwd)
