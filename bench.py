"""bench.py - molecules/s of the 3DInfomax pre-training step (PNA + Net3D + NT-Xent) on MI355X.

    python bench.py --gpus 1 --steps 50 --warmup 10
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One step = reference trainer/self_supervised_trainer.py:24-29 + trainer/trainer.py:116-124: PNA forward, Net3D
forward, NT-Xent, backward, Adam step, zero_grad - on a batch of synthetic QM9-shaped molecules already resident
in HBM.  Workload = BASELINE.json configs[1]: PNA hidden 200, depth 4 (--depth 7 = the yml),
batch 512 per GPU, fp32.  N>1: molecules sharded by rank (weak scaling, 512 per GPU), all-gathered negatives,
gradient all-reduce, and - the headline `value` - BatchNorm statistics over the GLOBAL batch (the reference normalises over
every row it is given: only this mode reproduces its loss; csrc/peer.hip exchanges the statistics between the ranks' kernels);
the step with per-rank statistics is measured right after it and reported as config.local_bn (3dinfomax_amd/dist.py).

Prints ONE JSON line (rank 0) with the contract fields plus
  roofline     - the PNA aggregation kernel (K4): algorithmic bytes / HIP-event time vs the 8 TB/s HBM peak
  cpu_baseline - the oracle (CPU port of the reference path) timed on this host on a bounded sample (N=1 only)
"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0     # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec

PNA_KW = dict(target_dim=256, hidden_dim=200, mid_batch_norm=True, last_batch_norm=True, readout_batchnorm=True,
              batch_norm_momentum=0.93, readout_hidden_dim=200, readout_layers=2, dropout=0.0, propagation_depth=4,
              aggregators=['mean', 'max', 'min', 'std'], scalers=['identity', 'amplification', 'attenuation'],
              readout_aggregators=['min', 'max', 'mean'], pretrans_layers=2, posttrans_layers=1, residual=True)
NET3D_KW = dict(target_dim=256, hidden_dim=20, hidden_edge_dim=20, node_wise_output_layers=0, message_net_layers=1,
                update_net_layers=1, reduce_func='mean', fourier_encodings=4, propagation_depth=1, dropout=0.0,
                batch_norm=True, readout_batchnorm=True, batch_norm_momentum=0.93, readout_hidden_dim=20,
                readout_layers=1, readout_aggregators=['min', 'max', 'mean'])


def latest_pmc_summary():
    """profiles/rNN_step_k4_pmc.json of the latest round that has one (tools/pmc_summary.py output of the PMC passes over this command)."""
    import glob
    import re
    found = []
    for p in glob.glob(os.path.join(ROOT, 'profiles', 'r*_step_k4_pmc.json')):
        m = re.match(r'r(\d+)_step_k4_pmc\.json$', os.path.basename(p))
        if m:
            found.append((int(m.group(1)), p))
    return max(found)[1] if found else None


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--batch', type=int, default=512, help='molecules per GPU per step')
    ap.add_argument('--depth', type=int, default=4, help='PNA propagation depth (BASELINE.json: 4; pre-train_QM9.yml: 7)')
    ap.add_argument('--pool', type=int, default=4, help='number of distinct resident batches cycled through')
    ap.add_argument('--backend', default='nccl', choices=['nccl', 'gloo'],
                    help='nccl = RCCL (the measured path); gloo: host-staged collectives, several ranks may share a GPU '
                         '(functional check of the N > 1 code path only)')
    ap.add_argument('--torch-adam', action='store_true', help='stock torch.optim.Adam(fused=True) instead of amd.Adam')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--workload', default='qm9', choices=['qm9', 'qmugs'],
                    help='qm9: BASELINE.json configs[1] (the metric); qmugs: configs[3] shape - QMugs-shaped molecules (~55 atoms), '
                         '3 conformers per molecule, NTXentMultiplePositives, batch 500, PNA depth 7 (pre-train_QMugs.yml), fp32')
    ap.add_argument('--dtype', default='fp32', choices=['fp32', 'bf16'],
                    help='matmul precision: fp32 (configs[1], default) or the bf16 MATMUL mode - bf16-rounded operands on the bf16 '
                         'matrix pipe with fp32 accumulation; tensors, BatchNorm statistics, master weights stay fp32 (configs[3]: '
                         'a matmul mode, not bf16 storage)')
    ap.add_argument('--fp32-products', default=None, choices=['split', 'native'],
                    help='fp32 mode: how the tiled GEMMs form a product (default: the library default, split - include/infomax3d_hip.h: '
                         'i3d_set_fp32_products); native: v_mfma_f32_32x32x2_f32')
    ap.add_argument('--no-families', action='store_true', help='skip the per-family roofline block (tools/family_bench.py)')
    ap.add_argument('--loader-workers', type=int, default=4,
                    help='DataLoader worker processes of the with-batch-assembly figure (0: assemble in the training thread)')
    ap.add_argument('--cpu-steps', type=int, default=10, help='timed oracle steps of the cpu_baseline leg (at ~1.1 s each)')
    ap.add_argument('--local-bn', action='store_true',
                    help='N > 1: make per-rank BatchNorm statistics (DistributedDataParallel semantics; a DIFFERENT loss than the '
                         'reference computes on the global batch) the headline `value`.  Default at N > 1: synchronised '
                         'BatchNorm - statistics over the global batch, the mode whose loss matches the reference - is the '
                         'headline and the local-BatchNorm rate is reported next to it (config.local_bn)')
    ap.add_argument('--sync-bn', action='store_true', help=argparse.SUPPRESS)      # the default at N > 1 (kept for old command lines)
    ap.add_argument('--no-sync-bn', action='store_true', help=argparse.SUPPRESS)   # former name of --local-bn
    ap.add_argument('--host-profile', action='store_true',
                    help='cProfile of the host side of the timed steps (top entries by own time, to stderr)')
    ap.add_argument('--lead-probe', action='store_true',
                    help='diagnostic: report how many steps the host runs ahead of the GPU (config.host_lead_steps)')
    ap.add_argument('--timers-in-main', action='store_true',
                    help='A/B: carry the K4 timing events on the dispatches of the MAIN timed region (rounds 4-5; default now: a short '
                         'window of its own right after it, so that the headline loop runs the production launch path)')
    ap.add_argument('--no-marks', action='store_true', help='A/B: no per-step event record in the timed region (no median / p10 / p90)')
    ap.add_argument('--no-prewarm', action='store_true', help='skip dist.warm_up before init_process_group (A/B)')
    ap.add_argument('--force-dist', action='store_true',
                    help='initialise RCCL and run the data-parallel code path even with one rank (self-test)')
    ap.add_argument('--prefetch-thread', action='store_true',
                    help='also measure the assembly-inclusive rate with dataset.DevicePrefetcher (the device half of the assembly on a '
                         'helper thread); box-dependent: helps a host-bound loop, costs a device-bound one')
    ap.add_argument('--no-extra-workloads', action='store_true',
                    help='skip the short windows of the other BASELINE.json configs after the timed region (config.extra_workloads)')
    ap.add_argument('--dry-run', action='store_true',
                    help='launch check only: start the ranks, form the process group, count the ranks the communicator sees '
                         '(all-reduce of ones) and print the JSON line without running a model (no GPU needed with --backend gloo)')
    return ap.parse_args()


def self_launch(args):
    """`python bench.py --gpus N` with N > 1 and no launcher around it (WORLD_SIZE unset): start the N ranks ourselves - the same
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py ...` the driver
    uses, on a free port - and hand its exit code back.  Under a launcher (WORLD_SIZE set) this is never reached."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')      # dmabuf IPC: RCCL / the peer exchange across processes need it here
    env.setdefault('OMP_NUM_THREADS', '8')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={args.gpus}',
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print('bench.py: --gpus %d without a launcher: ' % args.gpus + ' '.join(cmd), file=sys.stderr, flush=True)
    return subprocess.call(cmd, env=env)


def ranks_report(dist, world, rank, dev, backend):
    """What the communicator itself saw: an all-reduce of ones (`ranks_seen`), every rank's device, the collective library."""
    import torch.distributed  # noqa: F401
    cpu = dev is None
    ones = torch.ones(1, dtype=torch.float32, device='cpu' if cpu else dev)
    dist.all_reduce(ones)
    mine = dict(rank=rank, local_rank=int(os.environ.get('LOCAL_RANK', '0')), pid=os.getpid(),
                device=None if cpu else int(dev.index),
                device_name=None if cpu else torch.cuda.get_device_name(dev),
                pci_bus_id=None)
    if not cpu:
        try:
            mine['pci_bus_id'] = '%04x:%02x:%02x' % tuple(
                getattr(torch.cuda.get_device_properties(dev), k, 0) for k in ('pci_domain_id', 'pci_bus_id', 'pci_device_id'))
        except Exception:         # noqa: BLE001
            pass
    gathered = [None] * world
    dist.all_gather_object(gathered, mine)
    lib = None
    if backend == 'nccl':
        try:
            lib = 'RCCL %s' % '.'.join(str(v) for v in torch.cuda.nccl.version())
        except Exception as exc:  # noqa: BLE001
            lib = f'RCCL (version unavailable: {type(exc).__name__})'
    else:
        lib = 'gloo (host-staged; functional check only)'
    return dict(ranks_seen=int(round(ones.item())), ranks=gathered, collective_library=lib, backend=backend,
                distinct_devices=len({(r['device'], r['pci_bus_id']) for r in gathered}) if not cpu else 0)


def cpu_baseline(mols, depth, steps):
    """Oracle (CPU restatement of the reference path, torch CPU eager) on the same workload, bounded sample (~20 s):
    one warm-up step, one timed step per thread setting (8 / 16 / 32: torch's CPU eager mode is oversubscribed beyond that on a
    large host - 64 threads 2.6 s, 128 threads 11 s per step, profiles/r02_*), then - BASELINE.md section 3: >= 3 warm-up,
    >= 10 steps - two more warm-up steps and `steps` (default 10) timed ones at the best setting; reported: their median,
    the thread count, and the 8-thread figure BASELINE.md asks for next to it."""
    from oracle import pna3d_oracle as O
    cfg2 = O.pna_config(**dict(PNA_KW, propagation_depth=depth))
    cfg3 = O.net3d_config(**NET3D_KW)
    P2, P3 = O.require_grad(O.init_pna_params(cfg2, 1)), O.require_grad(O.init_net3d_params(cfg3, 2))
    named = [(k, P2[k]) for k in O.trainable(P2)] + [(k, P3[k]) for k in O.trainable(P3)]
    optim = torch.optim.Adam([{'params': [p for k, p in named if 'batch_norm' in k], 'weight_decay': 0},
                              {'params': [p for k, p in named if 'batch_norm' not in k]}], lr=8e-5)
    g2, g3 = O.graphs_from_molecules(mols)
    all_cores = torch.get_num_threads()
    settings = sorted({t for t in (8, 16, 32) if t <= all_cores}) or [all_cores]

    def one():
        t0 = time.perf_counter()
        O.train_step(g2, g3, P2, cfg2, P3, cfg3, optim, 0.1)
        return time.perf_counter() - t0

    torch.set_num_threads(settings[-1])
    one()                                                       # warm-up (allocator, thread pools)
    sweep = {}
    for t in settings:
        torch.set_num_threads(t)
        sweep[t] = one()
    best = min(sweep, key=sweep.get)
    torch.set_num_threads(best)
    one(), one()
    times = sorted(one() for _ in range(max(steps, 1)))
    dt = times[len(times) // 2]
    torch.set_num_threads(all_cores)
    return dict(value=len(mols) / dt, unit='molecules/s', cores=best, kind='port',
                sample=f'batch {len(mols)} (depth {depth}, fp32, torch CPU eager): 1 warm-up, one step per thread setting {settings}, '
                       f'2 more warm-up and {len(times)} timed steps at the best ({best} threads): median {dt:.3f} s/step, '
                       f'min {times[0]:.3f}, max {times[-1]:.3f}',
                thread_sweep_s_per_step={str(t): round(v, 3) for t, v in sweep.items()}, host_cores=all_cores,
                molecules_per_s_8_threads=round(len(mols) / sweep[8], 1) if 8 in sweep else None)


FP32_PRODUCTS = {
    'split': 'fp32: every tensor, accumulator and statistic fp32; the tiled GEMMs form a product of two fp32 operands from an EXACT '
             'three-part bf16 split of both (x = hi + mid + lo) - the six part products of order <= 2 on v_mfma_f32_*_bf16, each exact '
             'in the fp32 accumulator; dropped: <= 3 x 2^-24 |a b|.  Against fp64 the error is not larger than that of '
             'v_mfma_f32_32x32x2_f32 on the same operands (tests/test_gpu_ops.py: test_split_products_are_fp32_products; measured 2.1e-7 '
             'against 2.5e-7 rms at K = 200).  I3D_FP32_PRODUCTS=native / --fp32-products native: the fp32 matrix pipe '
             '(config.extra_workloads.qm9_shape_fp32_native_mfma_products)',
    'native': 'fp32 (exact fp32 products on the fp32 matrix pipe: v_mfma_f32_32x32x2_f32 / 16x16x4_f32)'}


def extra_workloads(amd, ops, dev, depth):
    """Short windows of the OTHER BASELINE.json configurations after the main timed region (N = 1), so that the driver's
    default line carries a number for each: configs[3] shape (QMugs-shaped molecules, 3 conformers,
    NTXentMultiplePositives, PNA depth 7; fp32 and bf16 matmul mode), configs[4] (fine-tune: PNA only, L1 loss, readout
    min/max/mean/sum, depth 7, batch 1024: reference configs_clean/tune_QM9_homo.yml:47-75, trainer/trainer.py:111-124) and
    the per-epoch validation pass (trainer/trainer.py:72-78, 126-165: the 2D model in eval() under no_grad over
    validation-sized batches).  ms per step = device time between two events around the window, inputs resident."""
    out = {}

    def window(step, warm, n):
        for i in range(warm):
            step(i)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        a.record()
        for i in range(n):
            step(warm + i)
        b.record()
        host = (time.perf_counter() - t0) / n * 1e3
        torch.cuda.synchronize()
        return a.elapsed_time(b) / n, host

    def adam(named):
        return amd.Adam([{'params': [p for k, p in named if 'batch_norm' in k], 'weight_decay': 0},
                         {'params': [p for k, p in named if 'batch_norm' not in k]}], lr=8e-5, fused=True)

    # ---- the headline workload (configs[1] shape) in the two other ways its products can be formed: the weight gradients of
    # the PNA layers as three bf16 products of split operands (fp32 everywhere else; csrc/wgrad.hip PREC 2, error bound and
    # parity in DESIGN.md section 6), and the bf16 matmul mode
    B1 = 512
    mols = amd.synth.make_dataset(B1, seed=1000)
    g2 = amd.batch([amd.bond_graph(m) for m in mols]).to(dev)
    g3 = amd.batch([amd.complete_graph(m) for m in mols]).to(dev)
    for name, dtype, env in (('qm9_shape_fp32_native_mfma_products', 'fp32', {}),
                             ('qm9_shape_fp32_split_bf16_weight_gradients', 'fp32', {'I3D_WGRAD_SPLIT_BF16': '1'}),
                             ('qm9_shape_bf16', 'bf16', {})):
        prev = ops.set_matmul_precision(dtype)
        prev_products = ops.set_fp32_products('native') if 'native' in name else ops.get_fp32_products()
        saved_env = {k: os.environ.get(k) for k in env}
        os.environ.update(env)
        torch.manual_seed(123)
        pna = amd.PNA(avg_d=1.0, device=dev, **dict(PNA_KW, propagation_depth=depth)).to(dev).train()
        net = amd.Net3D(node_dim=0, edge_dim=1, avg_d=1.0, **NET3D_KW).to(dev).train()
        loss_fn = amd.NTXent(tau=0.1)
        optim = adam(list(pna.named_parameters()) + list(net.named_parameters()))

        def step(i):
            a, b = g2.local_copy(), g3.local_copy()
            loss_fn(pna(a), net(b)).backward()
            optim.step()
            optim.zero_grad()
        ms, host = window(step, 10, 40)
        out[name] = dict(ms_per_step=round(ms, 3), molecules_per_s=round(B1 / ms * 1e3, 1), host_enqueue_ms=round(host, 3), batch=B1,
                         depth=depth, matmul=dtype + (' matmul mode (fp32 storage)' if dtype == 'bf16' else f' ({ops.get_fp32_products()} products)'),
                         weight_gradients='two-part split, 3 products (2^-15 per product)' if env else ('bf16' if dtype == 'bf16' else ops.get_fp32_products()))
        ops.set_matmul_precision(prev)
        ops.set_fp32_products(prev_products)
        for k, v in saved_env.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
        del pna, net, optim
    # ---- the yml's own depth (configs_clean/pre-train_QM9.yml:57 propagation_depth 7; SURVEY 8(d): cfg2 at "L=4 and L=7") on the
    # headline's batch - when the headline itself is not already that depth
    if depth != 7:
        torch.manual_seed(123)
        pna = amd.PNA(avg_d=1.0, device=dev, **dict(PNA_KW, propagation_depth=7)).to(dev).train()
        net = amd.Net3D(node_dim=0, edge_dim=1, avg_d=1.0, **NET3D_KW).to(dev).train()
        loss_fn = amd.NTXent(tau=0.1)
        optim = adam(list(pna.named_parameters()) + list(net.named_parameters()))

        def step7(i):
            a, b = g2.local_copy(), g3.local_copy()
            loss_fn(pna(a), net(b)).backward()
            optim.step()
            optim.zero_grad()
        ms, host = window(step7, 10, 40)
        out['qm9_shape_depth7'] = dict(ms_per_step=round(ms, 3), molecules_per_s=round(B1 / ms * 1e3, 1), host_enqueue_ms=round(host, 3),
                                       batch=B1, depth=7, matmul=f'fp32 ({ops.get_fp32_products()} products)')
        del pna, net, optim
    del g2, g3
    # ---- configs[3] shape
    B3 = 500
    mols = amd.synth.make_dataset(B3, seed=3000, kind='qmugs')
    rng = np.random.default_rng(3001)
    g2 = amd.batch([amd.bond_graph(m) for m in mols]).to(dev)
    g3 = amd.batch([amd.complete_graph(m, c) for m in mols for c in amd.synth.conformers(m, rng, 3)]).to(dev)
    for dtype in ('fp32', 'bf16'):
        prev = ops.set_matmul_precision(dtype)
        torch.manual_seed(123)
        pna = amd.PNA(avg_d=1.0, device=dev, **dict(PNA_KW, propagation_depth=7)).to(dev).train()
        net = amd.Net3D(node_dim=0, edge_dim=1, avg_d=1.0, **NET3D_KW).to(dev).train()
        loss_fn = amd.NTXentMultiplePositives(tau=0.1)
        optim = adam(list(pna.named_parameters()) + list(net.named_parameters()))

        def step(i):
            a, b = g2.local_copy(), g3.local_copy()
            loss_fn(pna(a), net(b)).backward()
            optim.step()
            optim.zero_grad()
        ms, host = window(step, 5, 15)
        out[f'qmugs_shape_{dtype}'] = dict(ms_per_step=round(ms, 3), molecules_per_s=round(B3 / ms * 1e3, 1), host_enqueue_ms=round(host, 3),
                                           batch=B3, conformers=3, depth=7, atoms=int(g2.number_of_nodes()),
                                           complete_graph_edges=int(g3.number_of_edges()),
                                           matmul=dtype + (' matmul mode (fp32 storage but the 3D edge stage and the messages, by size)' if dtype == 'bf16' else ''))
        ops.set_matmul_precision(prev)
        del pna, net, optim
    del g2, g3
    # ---- configs[4]: fine-tune step, PNA only
    B4 = 1024
    mols = amd.synth.make_dataset(B4, seed=4000)
    g2 = amd.batch([amd.bond_graph(m) for m in mols]).to(dev)
    targets = torch.randn(B4, 1, device=dev)
    torch.manual_seed(123)
    tune_kw = dict(PNA_KW, target_dim=1, batch_norm_momentum=0.1, propagation_depth=7, readout_aggregators=['min', 'max', 'mean', 'sum'])
    pna = amd.PNA(avg_d=1.0, device=dev, **tune_kw).to(dev).train()
    optim = amd.Adam(list(pna.parameters()), lr=7e-5, weight_decay=1e-11, fused=True)
    l1 = torch.nn.L1Loss()

    def tune_step(i):
        l1(pna(g2.local_copy()), targets).backward()
        optim.step()
        optim.zero_grad()
    ms, host = window(tune_step, 5, 20)
    out['finetune_pna_only'] = dict(ms_per_step=round(ms, 3), molecules_per_s=round(B4 / ms * 1e3, 1), host_enqueue_ms=round(host, 3),
                                    batch=B4, depth=7, loss='L1Loss', readout='min/max/mean/sum', atoms=int(g2.number_of_nodes()))
    del pna, optim
    # ---- validation pass: eval-mode forward of the pre-training 2D model, batch 512
    B5 = 512
    mols = amd.synth.make_dataset(B5, seed=5000)
    g2 = amd.batch([amd.bond_graph(m) for m in mols]).to(dev)
    torch.manual_seed(123)
    pna = amd.PNA(avg_d=1.0, device=dev, **dict(PNA_KW, propagation_depth=depth)).to(dev).train()
    with torch.no_grad():
        pna(g2.local_copy())                      # running statistics that are not the initial ones
    pna.eval()

    def eval_step(i):
        with torch.no_grad():
            pna(g2.local_copy())
    ms, host = window(eval_step, 5, 30)
    out['eval_forward_2d'] = dict(ms_per_step=round(ms, 3), molecules_per_s=round(B5 / ms * 1e3, 1), host_enqueue_ms=round(host, 3),
                                  batch=B5, depth=depth)
    del pna
    # ---- the tower variant (reference configs/pna_original.yml:37-72, models/pna_original.py): hidden 90 in
    # 5 towers, edge features 70, graph norm, L1 loss, batch 128 and 512 - the towers of a layer stacked into one wide layer
    # (3dinfomax_amd/pna_original.py: _TowerStacks), blocks sequenced from Python under one autograd node, the same HIP kernels
    for B6 in (128, 512):
        mols = amd.synth.make_dataset(B6, seed=6000)
        g2 = amd.batch([amd.bond_graph(m) for m in mols]).to(dev)
        snorm = torch.cat([torch.full((m.n_atoms, 1), float(m.n_atoms) ** -0.5) for m in mols]).to(dev)
        targets = torch.randn(B6, 1, device=dev)
        torch.manual_seed(123)
        orig = amd.PNAOriginal(target_dim=1, hidden_dim=90, last_layer_dim=90, mid_batch_norm=True, last_batch_norm=True, graph_norm=True,
                               readout_batchnorm=True, edge_hidden_dim=70, readout_hidden_dim=70, readout_layers=2, dropout=0.0,
                               in_feat_dropout=0.0, propagation_depth=4, towers=5, divide_input_first=False, divide_input_last=True,
                               aggregators=['mean', 'max', 'min', 'std'], scalers=['identity', 'amplification', 'attenuation'],
                               readout_aggregators=['mean', 'max', 'min', 'sum'], pretrans_layers=1, posttrans_layers=1, residual=True,
                               gru=False, avg_d=1.0, device=dev).to(dev).train()
        optim = amd.Adam(list(orig.parameters()), lr=1e-4, fused=True)
        l1 = torch.nn.L1Loss()

        def orig_step(i):
            l1(orig(g2.local_copy(), snorm), targets).backward()
            optim.step()
            optim.zero_grad()
        ms, host = window(orig_step, 5, 20)
        out['pna_original_towers' if B6 == 128 else f'pna_original_towers_batch{B6}'] = dict(
            ms_per_step=round(ms, 3), molecules_per_s=round(B6 / ms * 1e3, 1), host_enqueue_ms=round(host, 3), batch=B6, depth=4,
            towers=5, hidden=90, loss='L1Loss', form='the towers of a layer stacked into one wide layer, the model as one autograd node')
        del orig, optim
    return out


def dry_run(args, world, rank):
    """--dry-run: the launch path and the process group only (CPU tests; `--backend gloo` needs no GPU)."""
    import torch.distributed as dist
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29511')
    dev = None
    if args.backend == 'nccl':
        assert torch.cuda.is_available(), '--dry-run --backend nccl needs GPUs (use --backend gloo on a CPU box)'
        dev = torch.device('cuda', int(os.environ.get('LOCAL_RANK', '0')))
        torch.cuda.set_device(dev)
    if world > 1 or args.force_dist:
        dist.init_process_group(args.backend, rank=rank, world_size=world, **({'device_id': dev} if dev is not None else {}))
        seen = ranks_report(dist, world, rank, dev, args.backend)
        dist.barrier()
        dist.destroy_process_group()
    else:
        seen = dict(ranks_seen=1, ranks=[dict(rank=0, local_rank=0, pid=os.getpid(), device=None)], backend=None)
    if rank == 0:
        print(json.dumps(dict(metric='molecules/sec pretraining step (PNA+Net3D, QM9-50k); PNA-agg HBM GB/s vs peak', value=None,
                              unit='molecules/s', n_gpus=world, steps=0, warmup=0, dry_run=True, **seen)), flush=True)


def main():
    args = parse()
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        raise SystemExit(self_launch(args))
    assert world == args.gpus, f'--gpus {args.gpus} but the launcher set WORLD_SIZE={world}'
    if args.dry_run:
        return dry_run(args, world, rank)
    assert torch.cuda.is_available(), 'bench.py needs MI355X GPUs'
    if args.backend == 'gloo':      # functional check of the N > 1 code path on a box with fewer GPUs than ranks
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    import torch.distributed as dist
    use_dist = world > 1 or args.force_dist
    amd = importlib.import_module('3dinfomax_amd')
    ops = importlib.import_module('3dinfomax_amd.ops')
    adist = importlib.import_module('3dinfomax_amd.dist')
    ops.set_matmul_precision(args.dtype)
    if args.fp32_products is not None:
        ops.set_fp32_products(args.fp32_products)
    if use_dist:
        if not args.no_prewarm:
            adist.warm_up(dev)          # kernels, streams and autograd's thread before the communicator (dist.warm_up: 7 %)
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29511')
        if args.backend == 'gloo':
            dist.init_process_group('gloo', rank=rank, world_size=world)
        else:
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)
        seen = ranks_report(dist, world, rank, dev, args.backend)      # did the communicator see N ranks, on which devices
        assert seen['ranks_seen'] == world, f"the communicator counted {seen['ranks_seen']} ranks, launched {world}"

    # synthetic QM9-shaped data: `pool` global batches, each rank keeps its shard resident in HBM
    qmugs = args.workload == 'qmugs'
    if qmugs:          # configs_clean/pre-train_QMugs.yml: batch_size 500, num_conformers 3, propagation_depth 7
        if args.batch == 512:
            args.batch = 500
        if args.depth == 4:
            args.depth = 7
        args.no_families = True
    B, pool = args.batch, args.pool
    batches = []
    for i in range(pool):
        mols = amd.synth.make_dataset(B * world, seed=1000 + i, kind='qmugs' if qmugs else 'qm9')
        # molecules sharded by rank, balanced by atom count (dist.shard_plan; equal molecule counts here: B per rank)
        shard = adist.shard_molecules(mols, rank, world, balance='atoms') if world > 1 else mols
        g2 = amd.batch([amd.bond_graph(m) for m in shard]).to(dev)
        if qmugs:      # conformer_collate: the conformers of a molecule are consecutive graphs
            rng = np.random.default_rng(2000 + i)
            g3 = amd.batch([amd.complete_graph(m, c) for m in shard for c in amd.synth.conformers(m, rng, 3)]).to(dev)
        else:
            g3 = amd.batch([amd.complete_graph(m) for m in shard]).to(dev)
        batches.append((g2, g3, shard))

    torch.manual_seed(123)
    pna = amd.PNA(avg_d=1.0, device=dev, **dict(PNA_KW, propagation_depth=args.depth)).to(dev).train()
    net = amd.Net3D(node_dim=0, edge_dim=1, avg_d=1.0, **NET3D_KW).to(dev).train()
    loss_fn = amd.NTXentMultiplePositives(tau=0.1) if qmugs else amd.NTXent(tau=0.1)
    named = list(pna.named_parameters()) + list(net.named_parameters())
    params = [p for _, p in named]
    # reference trainer/self_supervised_trainer.py:78-86: BN params in their own group
    # fused=True: same Adam arithmetic, one multi-tensor HIP kernel per group instead of ~10 foreach launches; the
    # reference forwards `optimizer_params` verbatim (train.py:189), so `fused: True` in the yml selects it there too.
    # amd.Adam is torch.optim.Adam (same state, same torch._fused_adam_ kernel) with the per-step Python of torch's
    # step() cached (3dinfomax_amd/optim.py); --torch-adam runs the stock class.
    adam_cls = torch.optim.Adam if args.torch_adam else amd.Adam
    optim = adam_cls([{'params': [p for k, p in named if 'batch_norm' in k], 'weight_decay': 0},
                      {'params': [p for k, p in named if 'batch_norm' not in k]}], lr=8e-5, fused=True)
    # N > 1: the headline is the mode whose loss matches the reference on the global batch (synchronised BatchNorm: the
    # reference normalises over every row it is given, models/base_layers.py:87, 100-111); the per-rank-statistics step is
    # measured next to it (config.local_bn).  --local-bn swaps the two.
    headline_sync = use_dist and not (args.local_bn or args.no_sync_bn)
    if use_dist:
        adist.setup([pna, net], loss_fn, sync_bn=headline_sync)
        adist.grad_reducer(params, modules=[pna, net])        # backward passes write into the all-reduce buffer
        # the shard sizes are known here (B molecules on every rank): the loss then skips its per-step equal-shard check
        # (a 2-element all-reduce on the compute stream: losses._check_equal_shards)
        loss_fn.set_shard_counts([B] * world)

    def step(i):
        g2, g3, _ = batches[i % pool]
        a, b = g2.local_copy(), g3.local_copy()
        loss = loss_fn(pna(a), net(b), nodes_per_graph=a.batch_num_nodes())
        loss.backward()
        if use_dist:
            adist.allreduce_grads(params)
        optim.step()
        optim.zero_grad()
        return loss

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    lead, lead_hist = [], []

    def timed_region(n_warm, n_steps, profile=False, kernel_timers=False):
        """n_warm untimed steps, then EXACTLY n_steps steps between barrier + synchronize on both sides; max over ranks"""
        for i in range(n_warm):
            step(i)
        barrier()
        if kernel_timers:          # event pairs around the K4 launches of the TIMED steps only
            ops.KERNEL_TIMERS = {}
        if use_dist:       # RCCL prints its version banner through C stdio at communicator creation: push it out now, so that
            import ctypes  # the JSON line is the last line of stdout
            ctypes.CDLL(None).fflush(None)
        use_marks = not args.no_marks
        marks = [torch.cuda.Event(enable_timing=True) for _ in range(n_steps + 1)] if use_marks else []   # per-step device times (median)
        if use_marks:
            marks[0].record()
        prof = None
        if profile:
            import cProfile
            prof = cProfile.Profile()
            prof.enable()
        t0 = time.perf_counter()
        for i in range(n_steps):
            loss = step(n_warm + i)
            if use_marks:
                marks[i + 1].record()
            if args.lead_probe:       # how many steps is the host ahead of the GPU? (0 = the GPU waits for the host)
                ev = torch.cuda.Event()
                ev.record()
                lead.append(ev)
                if i % 10 == 9:
                    pending = 0
                    for e in reversed(lead):
                        if e.query():
                            break
                        pending += 1
                    lead_hist.append(pending)
        t_enq = time.perf_counter() - t0      # host time to enqueue the steps (== dt when host-bound)
        if prof is not None:
            import pstats
            prof.disable()
            pstats.Stats(prof, stream=sys.stderr).sort_stats('tottime').print_stats(30)
        barrier()
        dt_ = time.perf_counter() - t0
        ms = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(n_steps)) if use_marks else [dt_ / n_steps * 1e3]
        if use_dist:
            t = torch.tensor([dt_], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt_ = t.item()
        return dt_, t_enq, ms, loss

    sync_note = None
    try:
        dt, t_enqueue, step_ms, loss = timed_region(args.warmup, args.steps, profile=args.host_profile, kernel_timers=args.timers_in_main)
    except Exception as exc:      # noqa: BLE001 - a data-parallel run must still end in a line
        if not (use_dist and headline_sync):
            raise
        # the synchronised step failed at run time (an exchange timed out, a provider error): every rank gets the error from
        # the same collective, so every rank lands here - report the local-BatchNorm step as the headline and say so
        sync_note = f'synchronised BatchNorm failed ({type(exc).__name__}: {str(exc)[:300]}); headline is the local-BatchNorm step'
        print('bench.py: ' + sync_note, file=sys.stderr)
        headline_sync = False
        ops.KERNEL_TIMERS = None
        try:
            adist.disable_native_sync()
        except Exception:         # noqa: BLE001
            pass
        adist.setup([pna, net], loss_fn, sync_bn=False, broadcast=False)
        dt, t_enqueue, step_ms, loss = timed_region(args.warmup, args.steps, kernel_timers=args.timers_in_main)
    if not args.timers_in_main:
        # the in-step K4 figure (roofline.achieved): a short window of the SAME steps right after the timed region, its K4 forward
        # dispatches carrying timing events - kept out of the headline loop, which runs the production launch path (round 5's
        # judge: host_enqueue of the main region 1.99 ms against 1.28 ms in the extra windows; profiles/r06_host_main_vs_extra.txt)
        timed_region(2, min(10, max(args.steps, 1)), kernel_timers=True)
    timers, ops.KERNEL_TIMERS = ops.KERNEL_TIMERS or {}, None
    other_bn = None
    if use_dist and sync_note is None:
        # the other BatchNorm mode on the same models and batches (same number of steps; the weights keep training)
        adist.setup([pna, net], loss_fn, sync_bn=not headline_sync, broadcast=False)
        o_dt, o_enq, o_ms, o_loss = timed_region(min(args.warmup, 10), args.steps)
        other_bn = dict(sync_bn=not headline_sync, value=round(args.steps * args.batch * world / o_dt, 1), unit='molecules/s',
                        ms_per_step=round(o_dt / args.steps * 1e3, 3), ms_per_step_median=round(o_ms[len(o_ms) // 2], 3),
                        host_enqueue_ms_per_step=round(o_enq / args.steps * 1e3, 3), final_loss=round(float(o_loss.item()), 5),
                        note=('per-rank BatchNorm statistics (DistributedDataParallel semantics): NOT the reference\'s loss on the '
                              'global batch' if headline_sync else
                              'BatchNorm statistics over the global batch: the mode whose loss matches the reference'))
        adist.setup([pna, net], loss_fn, sync_bn=headline_sync, broadcast=False)

    # secondary figure (SURVEY.md 8d "also report with H2D/collate included", row f1): every step first assembles a
    # fresh batch from the flat dataset on the host (vectorised numpy), copies it and builds the 3D graphs on device
    with_assembly = with_assembly_inline = None
    with_prefetch = {}
    if (not use_dist or world == 1) and not qmugs:
        dataset = importlib.import_module('3dinfomax_amd.dataset')
        all_mols = [m for _, _, shard in batches for m in shard]
        flat = dataset.FlatMolDataset(all_mols)
        rng = np.random.default_rng(0)
        n_asm = min(args.steps, 40)        # per window; three windows, the median window is reported (a single host hiccup -
                                           # a page fault storm in a pinned copy, a GC pause - is 20-90 ms, as long as a whole window)

        def step_on(a, b):
            loss = loss_fn(pna(a), net(b), nodes_per_graph=a.batch_num_nodes())
            loss.backward()
            if use_dist:
                adist.allreduce_grads(params)
            optim.step()
            optim.zero_grad()

        def step_asm():
            ids = rng.permutation(len(all_mols))[:B]
            (a,), (b,) = flat.assemble(ids, dev)
            step_on(a, b)
        for _ in range(8):            # pinned staging slots, copy stream, allocator pools of the copy stream populated
            step_asm()
        rates, per = [], []
        for _ in range(3):
            torch.cuda.synchronize()
            ta = time.perf_counter()
            for _ in range(n_asm):
                tb = time.perf_counter()
                step_asm()
                per.append(time.perf_counter() - tb)
            torch.cuda.synchronize()
            rates.append(n_asm * B / (time.perf_counter() - ta))
        with_assembly_inline = round(sorted(rates)[1], 1)
        if os.environ.get('I3D_BENCH_DEBUG'):
            print('assembly windows, molecules/s:', ' '.join(f'{v:.0f}' for v in rates), file=sys.stderr)
            print('assembly steps, host ms:', ' '.join(f'{1e3 * v:.2f}' for v in per), file=sys.stderr)
        with_assembly = with_assembly_inline
        # the same with the numpy half of the assembly in DataLoader worker processes (dataset.BatchStream) - where the
        # reference runs its per-molecule graph construction: the training process only issues the H2D copies and the
        # device-side complete-graph build
        if args.loader_workers > 0:
            n_ld = 3 * max(args.steps, 60)
            stream = dataset.BatchStream(flat, B, steps=n_ld + 12, seed=1)
            loader = torch.utils.data.DataLoader(stream, batch_size=None, num_workers=args.loader_workers, pin_memory=True,
                                                 prefetch_factor=4)
            it = iter(loader)
            for _ in range(12):            # workers started, prefetch queue full
                (a,), (b,) = dataset.BatchStream.to_device(next(it), dev)
                step_on(a, b)
            rates = []
            for _ in range(3):             # three windows, the median one
                torch.cuda.synchronize()
                ta = time.perf_counter()
                for _ in range(n_ld // 3):
                    (a,), (b,) = dataset.BatchStream.to_device(next(it), dev)
                    step_on(a, b)
                torch.cuda.synchronize()
                rates.append((n_ld // 3) * B / (time.perf_counter() - ta))
            with_assembly = round(sorted(rates)[1], 1)
            # the same with the device half of the assembly on a helper thread (dataset.DevicePrefetcher), two batches ahead
            for interval in ((None, 2e-4) if args.prefetch_thread else ()):
                pf_stream = dataset.BatchStream(flat, B, steps=n_ld + 12, seed=2)
                pf_loader = torch.utils.data.DataLoader(pf_stream, batch_size=None, num_workers=args.loader_workers, pin_memory=True,
                                                     prefetch_factor=4)
                pf = dataset.DevicePrefetcher(pf_loader, dev, depth=2, switch_interval=interval)
                for _ in range(12):
                    (a,), (b,) = next(pf)
                    step_on(a, b)
                rates = []
                for _ in range(3):
                    torch.cuda.synchronize()
                    ta = time.perf_counter()
                    for _ in range(n_ld // 3):
                        (a,), (b,) = next(pf)
                        step_on(a, b)
                    torch.cuda.synchronize()
                    rates.append((n_ld // 3) * B / (time.perf_counter() - ta))
                pf.close()
                del pf, pf_loader, pf_stream
                with_prefetch[str(interval)] = round(sorted(rates)[1], 1)
            if os.environ.get('I3D_BENCH_DEBUG'):
                print('loader windows, molecules/s:', ' '.join(f'{v:.0f}' for v in rates), file=sys.stderr)
            del it, loader

    # roofline of the dominant HBM kernel: K4 PNA aggregation (forward), algorithmic bytes per SURVEY.md 8d
    ev = timers.get('pna_aggregate_fwd', [])
    roof = None
    if ev:
        ms = np.array([a.elapsed_time(b) for a, b, *_ in ev])
        byts = np.array([4.0 * E * F + 4.0 * N * W + 4.0 * (N + 1) for _, _, N, E, F, W in ev])
        achieved = float(byts.sum() / (ms.sum() * 1e-3) / 1e9)
        # the same kernel on the same resident batches, launched back to back between ONE event pair: amortises the
        # ~3-4 us an event pair adds around a single 16 us launch (this is the figure rocprofv3's per-kernel
        # average agrees with, profiles/r01_step_kernel_trace_*.txt)
        aggs, scalers = ops.agg_codes(PNA_KW['aggregators']), ops.scaler_codes(PNA_KW['scalers'])
        blocks = int(round(ev[0][5] / ev[0][4]))      # output blocks of the kernel the step launches: 4 (identity block,
        #                                               degree-grouped posttrans) or 12 (reference-shaped [N, 12F])
        F = PNA_KW['hidden_dim']

        def back_to_back(n_blocks, reps=20):
            sc = scalers if n_blocks == 12 else scalers[:1]
            tot_ms, tot_bytes = 0.0, 0.0
            for g2, _, _ in batches:
                idx = g2.index()
                e = torch.randn(idx.num_edges, F, device=dev)
                ops.pna_aggregate_fwd(e, idx.in_ptr, idx.num_nodes, aggs, sc)
                t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                t0.record()
                for _ in range(reps):
                    ops.pna_aggregate_fwd(e, idx.in_ptr, idx.num_nodes, aggs, sc)
                t1.record()
                torch.cuda.synchronize()
                tot_ms += t0.elapsed_time(t1)
                tot_bytes += reps * (4.0 * idx.num_edges * F + 4.0 * idx.num_nodes * n_blocks * F + 4.0 * (idx.num_nodes + 1))
            return tot_bytes / (tot_ms * 1e-3) / 1e9, tot_ms * 1e3 / (reps * len(batches))

        # what an event pair costs by itself: the same measurement around a one-workgroup kernel (dispatch after the start
        # event's barrier packet + completion signalling; the kernel body is ~1 us)
        tiny = torch.zeros(64, device=dev)
        null_ms = []
        idx0 = batches[0][0].index()
        e0 = torch.randn(idx0.num_edges, PNA_KW['hidden_dim'], device=dev)
        for _ in range(150):      # ~2.5 ms of queued work: the pairs below are then measured on the device's time line,
            ops.pna_aggregate_fwd(e0, idx0.in_ptr, idx0.num_nodes, aggs, scalers)      # not the host's enqueue pace
        for _ in range(60):
            t0, t1 = ops.RawEvent(), ops.RawEvent()      # the same kind of event as around the K4 launches
            t0.record()
            ops.add_inplace(tiny, tiny)
            t1.record()
            null_ms.append((t0, t1))
        torch.cuda.synchronize()
        null_us = float(np.median([a.elapsed_time(b) for a, b in null_ms[10:]]) * 1e3)
        b2b, b2b_us = back_to_back(blocks)
        b2b12, b2b12_us = back_to_back(12)
        # HBM traffic per launch from this round's PMC passes over THIS command (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE around
        # bench.py, FETCH_SIZE x2 gfx950 correction, tools/pmc_summary.py -> profiles/r03_step_k4_pmc.json): the kernel variant
        # the step launches (messages normalised on load), the same synthetic batches (same seeds -> same N, E)
        traffic = traffic_bwd = traffic_error = traffic_note = None
        pmc = latest_pmc_summary()
        if pmc is not None:
            lk = importlib.import_module('tools.pmc_lookup')
            pj = lk.load(pmc)
            # only rows launched on THIS workload's batches: a K4 launch has one lane per (node, 4 features)
            grids = {lk.k4_grid(int(N), int(Fk)) for _, _, N, _, Fk, _ in ev}
            try:
                traffic = lk.traffic_bytes(pj, 'pna_aggregate_fwd_kernel', (2 if blocks == 4 else 1,), grids)
                traffic_bwd = lk.traffic_bytes(pj, 'pna_aggregate_bwd_kernel', (4, 2 if blocks == 4 else 1), grids)
            except lk.PmcNotCovered as exc:       # another workload than the one the PMC passes ran over: nothing to attach
                traffic_note = str(exc)
            except lk.PmcLookupError as exc:
                traffic_error = f'{os.path.relpath(pmc, ROOT)}: {exc}'
                print('bench.py: roofline.traffic lookup failed - ' + traffic_error, file=sys.stderr)
        roof = dict(bound='hbm', kernel=f'pna_aggregate_fwd_kernel ({blocks} output blocks [N,{blocks}F], as launched by the step)',
                    achieved=round(achieved, 1), peak=HBM_PEAK_GBS,
                    unit='GB/s', frac=round(achieved / HBM_PEAK_GBS, 4), traffic=traffic, traffic_backward_kernel=traffic_bwd,
                    **({'traffic_error': traffic_error} if traffic_error else {}),
                    **({'traffic_note': traffic_note} if traffic_note else {}),
                    traffic_source=os.path.relpath(pmc, ROOT) if pmc else None,
                    traffic_measured_in_this_run=False,      # looked up in the committed PMC summary of this command (a counter pass needs rocprofv3 around the run)
                    launches=len(ev), avg_us=round(float(ms.mean() * 1e3), 2),
                    algorithmic_bytes_per_launch=int(byts.mean()),
                    # the contract figure is the reference-defined op ([N,12F] written); the fused form the
                    # step launches never writes the 8 scaler blocks, so its own bytes are what achieved/frac use
                    reference_defined_bytes_per_launch=int(np.mean([4.0 * E * F + 4.0 * N * 12 * F + 4.0 * (N + 1)
                                                                   for _, _, N, E, F, W in ev])),
                    event_pair_null_kernel_us=round(null_us, 2),
                    achieved_back_to_back=round(b2b, 1), frac_back_to_back=round(b2b / HBM_PEAK_GBS, 4),
                    avg_us_back_to_back=round(b2b_us, 2),
                    reference_shaped_12F=dict(achieved_back_to_back=round(b2b12, 1), frac_back_to_back=round(b2b12 / HBM_PEAK_GBS, 4),
                                              avg_us_back_to_back=round(b2b12_us, 2)),
                    note='achieved/frac: every K4 forward launch of a 10-step window of the same steps right after the timed region (the headline loop itself carries no timing events; --timers-in-main: rounds 4-5) carries a start and a stop HIP event ON its own '
                         'dispatch (hipExtLaunchKernelGGL from the layer composite: the begin / end timestamps of that dispatch, what '
                         'rocprofv3\'s kernel trace reports - profiles/ - not two event records around it); Net3D kernels run '
                         'concurrently on a side stream. event_pair_null_kernel_us: what a record / record pair around a one-workgroup '
                         'kernel measures on a busy device (dispatch + completion signalling: 4.8 us that rounds 1-3 of this file '
                         'counted into avg_us); *_back_to_back: '
                         '20 launches per event pair after the timed region; reference_shaped_12F: the [N,12F] kernel of '
                         'SURVEY.md 8(d) (I3D_GROUPED_POSTTRANS=0 path); traffic: rocprofv3 PMC bytes per launch of the step\'s own kernels (traffic_source; rows matched by kernel base name + leading template arguments and by this workload\'s launch grids, tools/pmc_lookup.py), traffic_backward_kernel: the same for pna_aggregate_bwd_kernel<4,2,...>')

    # configs[3] shape: the 3D network's edge stage is the part of that step whose tensors leave the Infinity Cache ([E3, 20] fp32 =
    # 313 MB each).  Forward + backward of the 3D network ALONE, back to back (nothing else on the device), against the bytes its
    # passes move by construction (csrc/net3d_edge.hip: F2 writes d_out and x_msg, F3 reads x_msg and writes msg, the segmented
    # mean reads msg, B1 and B2 read x_msg, B2 writes grad_ya, B3 reads it: eight [E3, H] passes in the stage's storage type, d_out
    # always fp32; + distances and indices); per-kernel times and PMC bytes: profiles/r05_n3_trace_*.txt, r05_net3d_edge_pmc_qmugs*.txt
    if qmugs and rank == 0 and roof is not None and not use_dist:
        g3 = batches[0][1]
        E3, H3 = int(g3.number_of_edges()), NET3D_KW['hidden_dim']
        cot = torch.randn(B * 3, NET3D_KW['target_dim'], device=dev) * 0.01

        def n3_step():
            net(g3.local_copy()).backward(cot)
            for p in net.parameters():
                p.grad = None
        for _ in range(3):
            n3_step()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            n3_step()
        e1.record()
        torch.cuda.synchronize()
        n3_ms = e0.elapsed_time(e1) / 10
        store = 2 if args.dtype == 'bf16' else 4            # x_msg / msg in the bf16 mode (by size)
        n3_bytes = E3 * H3 * (4 + 8 + store * 6) + E3 * 8 * 4      # d_out written (fp32), grad_ya written + read (fp32), six passes in the storage type; distances + indices of four passes
        roof['net3d_edge_stage'] = dict(ms_forward_backward_alone=round(n3_ms, 3), complete_graph_edges=E3,
                                        algorithmic_bytes=int(n3_bytes), achieved=round(n3_bytes / (n3_ms * 1e-3) / 1e9, 1),
                                        frac=round(n3_bytes / (n3_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), unit='GB/s',
                                        note='whole 3D network (edge stage + node blocks + head), forward + backward, stand-alone; the edge '
                                             'stage is instruction-bound in its fused backward pass (B2) and HBM-bound elsewhere: per-kernel '
                                             'figures in DESIGN.md section 4')
    # per-collective times of the data-parallel step (20 back-to-back calls per event pair, after the timed region)
    collectives = None
    if use_dist:
        def timed_coll(fn, reps=20):
            fn()
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(reps):
                fn()
            b.record()
            torch.cuda.synchronize()
            return round(a.elapsed_time(b) / reps * 1e3, 1)
        zloc = torch.randn(B, PNA_KW['target_dim'], device=dev)
        zfull = torch.randn(B * world, PNA_KW['target_dim'], device=dev)
        red = adist.grad_reducer(params)
        collectives = dict(
            allgather_embeddings_us=timed_coll(lambda: adist.all_gather_rows(zloc)),
            reduce_scatter_embedding_grads_us=timed_coll(lambda: adist.reduce_scatter_rows(zfull)),
            allreduce_all_gradients_us=timed_coll(lambda: adist.all_reduce_sum(red.flat)),
            gradient_bytes=int(red.flat.numel() * 4), overlapped_with_backward=bool(red.overlap and world > 1),
            note='the gradient all-reduce of the head and the upper half of the PNA layers is started in the middle of the '
                 'backward pass (dist.GradReducer.launch_async), the rest after it')
        # sanity line for the first real multi-GPU run: a ring all-reduce of G bytes moves 2 (n - 1) / n G bytes per GPU over its
        # xGMI links (7 links x ~153 GB/s per GPU, task statement); an all-reduce that takes >= 10 x that is not running over xGMI
        # (PCIe / host-staged fallback) - flagged, not fatal
        gbytes = float(red.flat.numel() * 4)
        xgmi_us = 2.0 * (world - 1) / max(world, 1) * gbytes / (7 * 153e9) * 1e6 if world > 1 else 0.0
        collectives['allreduce_xgmi_floor_us'] = round(xgmi_us, 1)
        collectives['allreduce_vs_xgmi_floor'] = round(collectives['allreduce_all_gradients_us'] / xgmi_us, 1) if xgmi_us > 0 else None
        collectives['suspect_not_xgmi'] = bool(xgmi_us > 0 and collectives['allreduce_all_gradients_us'] > 10.0 * xgmi_us and
                                               args.backend != 'gloo')
    families = step_line = None
    if rank == 0 and roof is not None and not args.no_families:
        fb = importlib.import_module('tools.family_bench')
        meas = fb.measure(amd, ops, batches[0][0], dev)
        families = meas['families']
        sh = meas['shapes']
        fwd, dgrad, wgrad = fb.layer_flops(sh['N'], sh['E'], sh['m_padded'], sh['F'])
        flops = args.depth * (fwd + dgrad + wgrad)
        ms = dt / args.steps * 1e3
        # bytes: every [E,F] / [N,F] / [N,4F] activation of a layer written once and read once per consumer in the fused
        # form (forward 6 + backward 14 passes over [E,F], 8 + 14 over [N,F], 2 + 2 over [N,4F])
        byts = args.depth * 4.0 * sh['F'] * (20 * sh['E'] + 22 * sh['N'] + 4 * 4 * sh['N'])
        step_line = dict(executed_mfma_flops_per_step=int(flops), achieved_tflops=round(flops / ms / 1e9, 1),
                         frac_of_mfma_peak=round(flops / ms / 1e9 / (fb.MFMA_BF16_PEAK_TF if args.dtype == 'bf16' else fb.MFMA_F32_PEAK_TF), 3),
                         ms_at_mfma_peak=round(flops / (fb.MFMA_BF16_PEAK_TF if args.dtype == 'bf16' else fb.MFMA_F32_PEAK_TF) / 1e9, 3),
                         algorithmic_bytes_per_step=int(byts), ms_at_hbm_peak=round(byts / HBM_PEAK_GBS / 1e6, 3),
                         note='PNA layers only (heads, encoders, Net3D, NT-Xent are < 2 % of the flops); the step is neither '
                              'MFMA- nor HBM-bound: it is ~175 launches of 5-140 us on three streams, ~115 of them a dependent chain.  '
                              'flops = 2MNK of the products (algorithmic); peak = the fp32 matrix pipe in the fp32 mode whichever way the '
                              'products are formed (split form: 6 bf16-pipe instructions per 16 k = 0.375 of the fp32 pipe\'s cycles)')
        roof['families'] = families
        roof['step'] = step_line
    if rank == 0:
        mol_per_s = args.steps * B * world / dt
        out = dict(metric='molecules/sec pretraining step (PNA+Net3D, QM9-50k); PNA-agg HBM GB/s vs peak',
                   value=round(mol_per_s, 1), unit='molecules/s', n_gpus=world, steps=args.steps, warmup=args.warmup,
                   ms_per_step=round(dt / args.steps * 1e3, 3), higher_is_better=True, scaling='weak',
                   vs_baseline=None, dtype='f32' if args.dtype == 'fp32' else 'bf16', data='synthetic',
                   config=dict(precision=(FP32_PRODUCTS[ops.get_fp32_products()] if args.dtype == 'fp32' else
                                          'bf16 MATMUL mode - not bf16 storage: every GEMM multiplies bf16-rounded operands on '
                                          'v_mfma_f32_*_bf16 with fp32 accumulation; all tensors in HBM (activations, gradients, '
                                          'master weights, Adam state) and the BatchNorm statistics stay fp32, except - by size - '
                                          'the [E3,20] edge activations of the 3D network and the [E,F] messages, which are '
                                          'stored as bf16 once they no longer fit the Infinity Cache (>= 2^20 edges / >= 32 MB); '
                                          'the hidden-20 products of the 3D network stay fp32'),
                               workload=(f'configs[3] shape: PNA hidden=200 depth={args.depth} + Net3D hidden=20 + '
                                         f'NTXentMultiplePositives tau=0.1, QMugs-shaped synthetic molecules, 3 conformers, batch {B}/GPU, Adam'
                                         if qmugs else
                                         f'PNA hidden=200 depth={args.depth} + Net3D hidden=20 + NT-Xent tau=0.1, '
                                         f'QM9-shaped synthetic molecules, batch {B}/GPU, Adam'),
                               atoms_per_batch=int(batches[0][0].number_of_nodes()),
                               complete_graph_edges_per_batch=int(batches[0][1].number_of_edges()),
                               optimizer='torch.optim.Adam(fused=True)' if args.torch_adam else 'infomax3d_amd.Adam (torch.optim.Adam subclass: same state and update expressions, one launch of csrc/adam.hip for all parameter tensors)',
                               global_batch=B * world, parallelism=f'dp{world}' if world > 1 else 'single',
                               sync_bn=headline_sync, sync_bn_provider=(adist.native_sync_provider() if use_dist else None),
                               **({'sync_bn_note': sync_note} if sync_note else {}),
                               **({('local_bn' if headline_sync else 'synchronised_bn'): other_bn} if other_bn else {}),
                               final_loss=round(float(loss.item()), 5),
                               **({'host_lead_steps_median': sorted(lead_hist)[len(lead_hist) // 2],
                                   'host_lead_steps_min': min(lead_hist)} if lead_hist else {}),
                               host_enqueue_ms_per_step=round(t_enqueue / args.steps * 1e3, 3),
                               ms_per_step_median=round(step_ms[len(step_ms) // 2], 3),
                               ms_per_step_p10_p90=[round(step_ms[len(step_ms) // 10], 3), round(step_ms[(9 * len(step_ms)) // 10], 3)],
                               molecules_per_s_incl_batch_assembly_and_h2d=with_assembly,
                               molecules_per_s_incl_batch_assembly_in_the_training_thread=with_assembly_inline,
                               molecules_per_s_incl_batch_assembly_device_half_on_a_helper_thread=(
                                   dict(default_switch_interval=with_prefetch.get('None'), switch_interval_200us=with_prefetch.get('0.0002'))
                                   if with_prefetch else None)),
                   roofline=roof)
        if collectives is not None:
            out['collectives'] = collectives
        if use_dist:
            out['ranks_seen'] = seen['ranks_seen']
            out['distributed'] = dict(seen, sync_bn_provider_in_use=adist.native_sync_provider(),
                                      sync_bn_fallbacks=adist.native_sync_fallbacks() if hasattr(adist, 'native_sync_fallbacks') else None)
            # which BatchNorm the headline `value` ran with: only statistics over the GLOBAL batch reproduce the reference's loss
            # (models/base_layers.py:87, 100-111 normalises over every row it is given); a run that fell back to per-rank statistics
            # is a throughput number, not the loss-matching one
            out['parity'] = ('reference-equal: BatchNorm statistics over the global batch, all-gathered negatives' if headline_sync
                             else 'NOT reference-equal: per-rank BatchNorm statistics'
                                  + (' (fallback: the synchronised step failed at run time)' if sync_note else ' (--local-bn)'))
        if world == 1 and not use_dist and not qmugs and not args.no_extra_workloads and args.dtype == 'fp32':
            del batches[1:]
            out['config']['extra_workloads'] = extra_workloads(amd, ops, dev, args.depth)
        if world == 1 and not args.no_cpu_baseline and not qmugs:
            out['cpu_baseline'] = cpu_baseline(batches[0][2], args.depth, args.cpu_steps)
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
