"""GPU microbenchmark: the weight gradients of ONE PNA layer's backward at the step's shapes (batch 512 QM9-shaped) -
round 2's launches (5 split-K GEMMs + row-segment GEMM + slice reductions + fold-back) against ONE i3d_wgrad_multi call.
    python tools/wgrad_bench.py [--nodes 8409 --edges 16638] [--units 128,192,256,320]
"""
import argparse
import importlib
import os
import subprocess
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--nodes', type=int, default=8409)
    ap.add_argument('--edges', type=int, default=16638)
    ap.add_argument('--hidden', type=int, default=200)
    ap.add_argument('--reps', type=int, default=50)
    ap.add_argument('--units', default='')
    a = ap.parse_args()
    if a.units:
        for u in a.units.split(','):
            env = dict(os.environ, I3D_WGRAD_UNITS=u)
            subprocess.run([sys.executable, __file__, '--nodes', str(a.nodes), '--edges', str(a.edges), '--hidden',
                            str(a.hidden), '--reps', str(a.reps)], env=env, check=True)
        return
    ops = importlib.import_module('3dinfomax_amd.ops')
    dev = torch.device('cuda:0')
    N, E, F = a.nodes, a.edges, a.hidden
    A4, S, V = 4 * F, 3, 64
    r = lambda *s: torch.randn(*s, device=dev)
    dpost, h, agg, dpre2, x1, dP, dpre1 = r(N, F), r(N, F), r(N, A4), r(E, F), r(E, F), r(N, 2 * F), r(E, F)
    onehot = torch.zeros(E, V, device=dev)
    onehot[torch.arange(E), torch.randint(0, 60, (E,))] = 1.0
    # QM9-like in-degree groups: ~half hydrogens (D = 1), the rest D = 2..4
    counts = [int(N * 0.52), int(N * 0.05), int(N * 0.13), 0]
    counts[3] = N - sum(counts)
    rows, starts, o = [], [], 0
    perm = torch.randperm(N).tolist()
    for c in counts:
        starts.append(len(rows))
        rows += perm[o:o + c] + [-1] * ((c + 63) // 64 * 64 - c)
        o += c
    deg_rows = torch.tensor(rows, dtype=torch.int32, device=dev)
    coef = [1.0, 0.7, 1.4, 1.0, 1.1, 0.9, 1.0, 1.4, 0.7, 1.0, 1.6, 0.6]
    aff, row = r(3 * F), r(F)
    ldw = F + S * A4
    gW_post, gW2, gW1, gQ = r(F, ldw), r(F, F), r(F, 3 * F), r(V, F)
    gWD = torch.empty(len(counts), F, A4, device=dev)
    problems = [dict(A=dpost, B=h)]
    for s0, c in zip(starts, counts):
        problems.append(dict(A=dpost, B=agg, rows=deg_rows, k_begin=s0, k_count=c))
    problems += [dict(A=dpre2, B=x1), dict(A=dP, B=h), dict(A=onehot, B=dpre1)]
    G = len(counts)
    outputs = [dict(first_problem=0, C=gW_post, ldc=ldw),
               dict(kind=ops.WGRAD_COMBINE, first_problem=1, n_groups=G, C=gW_post, c_offset=F, ldc=ldw, coef=coef, n_scalers=S,
                    scaler_stride=A4),
               dict(kind=ops.WGRAD_BN, first_problem=1 + G, C=gW2, aff=aff, row=row),
               dict(first_problem=2 + G, C=gW1, ldc=3 * F, c_split=F, c_delta=F - F * 3 * F),
               dict(first_problem=3 + G, C=gQ)]

    def new():
        ops.wgrad_multi(problems, outputs)

    def old():
        ops.gemm(dpost, h, trans_a=True, out=gW_post[:, :F])
        ops.gemm_rowsubset_multi(dpost, agg, deg_rows, starts, counts, gWD)
        ops.combine_weights_bwd(gWD, gW_post, F, A4, coef, G, S)
        ops.gemm_wgrad_bn(dpre2, x1, row, aff)
        ops.gemm(dP[:, :F], h, trans_a=True, out=gW1[:, :F])
        ops.gemm(dP[:, F:], h, trans_a=True, out=gW1[:, F:2 * F])
        ops.gemm(onehot, dpre1, trans_a=True, out=gQ)

    if os.environ.get('WG_TIMING'):
        new()
        stamps(ops, dev, new)
        return
    flops = 2.0 * (N * F * F + N * F * A4 + E * F * F + N * 2 * F * F + E * V * F)
    for name, fn in (('round-2 launches', old), ('i3d_wgrad_multi', new)):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()
        for _ in range(a.reps):
            fn()
        e1.record()
        host = (time.perf_counter() - t0) / a.reps * 1e6
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / a.reps * 1e3
        print(f'{name:18s} units={os.environ.get("I3D_WGRAD_UNITS", "256"):>4s}  {us:8.1f} us / layer   '
              f'{flops / us / 1e6:6.1f} TF/s = {flops / us / 1e6 / 157.3 * 100:4.1f} % of the fp32 MFMA peak   host {host:6.1f} us')


def stamps(ops, dev, fn):
    """probe build (-DWG_TIMING, I3D_LIB_PATH): per-workgroup cycle stamps from the tail of the scratch"""
    ws = ops._gemm_workspace(dev)
    tail = ws[-(1 << 16):].view(torch.int64)
    tail.zero_()
    fn()
    torch.cuda.synchronize()
    t = tail.view(-1, 8).cpu()
    t = t[t[:, 3] != 0]
    if t.shape[0] == 0:
        return
    pro, loop, epi = (t[:, 1] - t[:, 0]).float(), (t[:, 2] - t[:, 1]).float(), (t[:, 3] - t[:, 2]).float()
    start, end = t[:, 0].float(), t[:, 3].float()
    print(f'  {t.shape[0]} workgroups; cycles (mean / max): prologue {pro.mean():.0f} / {pro.max():.0f}  K loop {loop.mean():.0f} / '
          f'{loop.max():.0f}  panel store {epi.mean():.0f} / {epi.max():.0f}; first start .. last end {end.max() - start.min():.0f}, '
          f'start spread {start.max() - start.min():.0f}')
    for pi in sorted(set(t[:, 7].tolist())):
        m = t[:, 7] == pi
        print(f'    problem {pi}: {int(m.sum())} units, K loop mean {loop[m].mean():.0f} max {loop[m].max():.0f}, '
              f'total mean {(end[m] - start[m]).mean():.0f}')


if __name__ == '__main__':
    main()
