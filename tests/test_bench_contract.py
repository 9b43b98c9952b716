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


def test_bench_starts_its_own_ranks_from_a_plain_python_command():
    """the driver's N = 1 command shape (`python bench.py --gpus N ...`) must also start at N > 1 - bench.py
    re-executes itself under torch.distributed.run on a free port when no launcher set WORLD_SIZE - and the line must say how
    many ranks the communicator saw.  --dry-run --backend gloo: launch path + process group only, no GPU."""
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '3', '--warmup', '1',
                        '--backend', 'gloo', '--dry-run'], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, r.stdout          # ONE line, from rank 0
    out = json.loads(lines[0])
    assert out['n_gpus'] == 2 and out['ranks_seen'] == 2 and out['dry_run'] is True
    assert sorted(x['rank'] for x in out['ranks']) == [0, 1] and len({x['pid'] for x in out['ranks']}) == 2
    assert 'torch.distributed.run' in r.stderr          # it says what it launched


def test_bench_under_a_launcher_does_not_launch_again():
    """The driver's own N > 1 command (torch.distributed.run around bench.py) keeps working: WORLD_SIZE set -> no re-exec."""
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr',
                        '127.0.0.1', '--master-port', '29641', os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--backend', 'gloo',
                        '--dry-run'], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith('{')][0])
    assert out['ranks_seen'] == 2 and 'without a launcher' not in r.stderr
