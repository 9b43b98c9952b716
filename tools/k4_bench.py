"""Isolated run of the PNA aggregation kernel (K4) on QM9-shaped batches: back-to-back launches timed with HIP
events, algorithmic bytes per SURVEY.md 8(d).  Also the target of the rocprofv3 --pmc passes (FETCH_SIZE /
WRITE_SIZE) whose per-launch HBM traffic goes into bench.py's roofline.traffic.

    python tools/k4_bench.py [--batches 512 2048 8192] [--reps 50] [--bwd]
"""
import argparse
import importlib
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batches', type=int, nargs='+', default=[512, 2048, 8192])
    ap.add_argument('--reps', type=int, default=50)
    ap.add_argument('--feat', type=int, default=200)
    ap.add_argument('--bwd', action='store_true')
    a = ap.parse_args()
    amd = importlib.import_module('3dinfomax_amd')
    ops = importlib.import_module('3dinfomax_amd.ops')
    dev = torch.device('cuda:0')
    aggs, scalers = ops.agg_codes(['mean', 'max', 'min', 'std']), ops.scaler_codes(['identity', 'amplification', 'attenuation'])
    for B in a.batches:
        mols = amd.synth.make_dataset(B, seed=1000)
        idx = amd.batch([amd.bond_graph(m) for m in mols]).index().to(dev)
        N, E, F = idx.num_nodes, idx.num_edges, a.feat
        e = torch.randn(E, F, device=dev)
        gout = torch.randn(N, 12 * F, device=dev)
        gout4 = torch.randn(N, 4 * F, device=dev)
        fwd_bytes = 4.0 * E * F + 4.0 * N * 12 * F + 4.0 * (N + 1)
        bwd_bytes = 4.0 * N * 12 * F + 2 * 4.0 * E * F + 4.0 * (N + 1)
        fwd4_bytes = 4.0 * E * F + 4.0 * N * 4 * F + 4.0 * (N + 1)
        bwd4_bytes = 4.0 * N * 4 * F + 2 * 4.0 * E * F + 4.0 * (N + 1)
        ident = scalers[:1]
        # "12F": the reference-shaped output [N, 12F]; "4F": identity block only (degree-grouped posttrans, the step's kernel)
        for name, fn, byts in (('fwd 12F', lambda: ops.pna_aggregate_fwd(e, idx.in_ptr, N, aggs, scalers), fwd_bytes),
                               ('fwd  4F', lambda: ops.pna_aggregate_fwd(e, idx.in_ptr, N, aggs, ident), fwd4_bytes),
                               ('bwd 12F', lambda: ops.pna_aggregate_bwd(gout, e, idx.in_ptr, N, aggs, scalers), bwd_bytes),
                               ('bwd  4F', lambda: ops.pna_aggregate_bwd(gout4, e, idx.in_ptr, N, aggs, ident), bwd4_bytes)):
            if name.startswith('bwd') and not a.bwd:
                continue
            for _ in range(3):
                fn()
            t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            t0.record()
            for _ in range(a.reps):
                fn()
            t1.record()
            torch.cuda.synchronize()
            us = t0.elapsed_time(t1) * 1e3 / a.reps
            print(f'K4 {name} B={B:5d} N={N:7d} E={E:7d} F={F}: {us:8.2f} us/launch  algorithmic {byts / 1e6:8.1f} MB '
                  f'-> {byts / us * 1e-3:7.1f} GB/s = {byts / us * 1e-3 / 8000:.3f} of 8 TB/s', flush=True)


if __name__ == '__main__':
    main()
