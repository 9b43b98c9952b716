"""GPU microbenchmark of i3d_gemm_f32 on the GEMM shapes of the pre-training step (batch 512 QM9-shaped):
prints time and TFLOP/s per (shape, tile config, split-K).  Run on the MI355X box:
    python tools/gemm_bench.py [--all-cfgs]
"""
import argparse
import ctypes
import importlib
import os
import sys
from ctypes import c_void_p

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
L = importlib.import_module('3dinfomax_amd._lib')

N_NODES, N_EDGES, F = 8320, 19400, 200
SHAPES = [  # name, ta, tb, M, N, K
    ('P      X[N,F] W^T        ', 0, 1, N_NODES, F, F),
    ('Q      ef[E,F] W^T       ', 0, 1, N_EDGES, F, F),
    ('post   agg[N,12F] W^T    ', 0, 1, N_NODES, F, 12 * F),
    ('dgrad  dY[E,F] W         ', 0, 0, N_EDGES, F, F),
    ('dgrad  dY[N,F] W[F,12F]  ', 0, 0, N_NODES, 12 * F, F),
    ('wgrad  dY^T[F,N] X[N,F]  ', 1, 0, F, F, N_NODES),
    ('wgrad  dY^T[F,E] X[E,F]  ', 1, 0, F, F, N_EDGES),
    ('wgrad  dY^T[F,N] agg     ', 1, 0, F, 12 * F, N_NODES),
    ('post4  agg[N,4F] W^T     ', 0, 1, N_NODES, F, 4 * F),
    ('P2     X[N,F] [Ws|Wd]^T  ', 0, 1, N_NODES, 2 * F, F),
    ('dP     dP[N,2F] [Ws;Wd]  ', 0, 0, N_NODES, F, 2 * F),
    ('dgrad4 dY[N,F] W[F,4F]   ', 0, 0, N_NODES, 4 * F, F),
    ('net3d  d[E3,20] W^T      ', 0, 1, 140000, 20, 20),
    ('net3d  wgrad [20,20]     ', 1, 0, 20, 20, 140000),
    ('net3d  wgrad [20,13]     ', 1, 0, 20, 13, 140000),
    ('head   r[512,600] W^T    ', 0, 1, 512, 200, 600),
    ('sim    z1 z2^T           ', 0, 1, 512, 512, 256),
]


WS_BYTES = 64 << 20
_ws = None


def run(lib, ta, tb, M, N, K, cfg, splits, reps=20, scratch=True):
    dev = torch.device('cuda:0')
    A = torch.randn((K, M) if ta else (M, K), device=dev)
    B = torch.randn((N, K) if tb else (K, N), device=dev)
    C = torch.empty(M, N, device=dev)
    st = c_void_p(torch.cuda.current_stream().cuda_stream)
    global _ws
    if _ws is None:
        _ws = torch.empty(WS_BYTES, dtype=torch.uint8, device=dev)
    args = (ta, tb, M, N, K, c_void_p(A.data_ptr()), A.shape[1], c_void_p(B.data_ptr()), B.shape[1],
            c_void_p(C.data_ptr()), N, None, 0, cfg, splits, c_void_p(_ws.data_ptr()) if scratch else None,
            WS_BYTES if scratch else 0, st)
    for _ in range(3):
        assert lib.i3d_gemm_f32_ex(*args) == 0, lib.i3d_last_error()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        lib.i3d_gemm_f32_ex(*args)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    ref = (A.T if ta else A).double() @ (B.T if tb else B).double()
    err = ((C.double() - ref).abs().max() / ref.abs().max()).item()
    return us, 2.0 * M * N * K / us * 1e-6, err


def run_rowseg(lib, cfg, seg_rows, reps=20, scratch=True):
    """weight gradients of the in-degree groups (QM9-like degree histogram) in one launch"""
    dev = torch.device('cuda:0')
    torch.manual_seed(0)
    deg = torch.multinomial(torch.tensor([0.0, 0.42, 0.13, 0.17, 0.28]), N_NODES, replacement=True)
    order = torch.argsort(deg, stable=True).int()
    counts = torch.bincount(deg, minlength=5).tolist()
    starts, cnts, o = [], [], 0
    for c in counts:
        if c:
            starts.append(o)
            cnts.append(c)
        o += c
    G = len(starts)
    dY, a = torch.randn(N_NODES, F, device=dev), torch.randn(N_NODES, 4 * F, device=dev)
    rows = order.to(dev)
    out = torch.zeros(G, F, 4 * F, device=dev)
    ia = lambda v: (ctypes.c_int * len(v))(*v)
    st = c_void_p(torch.cuda.current_stream().cuda_stream)
    args = (F, 4 * F, G, ia(starts), ia(cnts), c_void_p(dY.data_ptr()), F, c_void_p(a.data_ptr()), 4 * F,
            c_void_p(rows.data_ptr()), N_NODES, c_void_p(out.data_ptr()), F * 4 * F, 4 * F, 0, cfg, seg_rows,
            c_void_p(_ws.data_ptr()) if scratch else None, WS_BYTES if scratch else 0, st)
    for _ in range(3):
        assert lib.i3d_gemm_f32_rowsubset_multi(*args) == 0, lib.i3d_last_error()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        lib.i3d_gemm_f32_rowsubset_multi(*args)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    ref = torch.stack([dY[rows[s:s + c].long()].double().T @ a[rows[s:s + c].long()].double() for s, c in zip(starts, cnts)])
    err = ((out.double() - ref).abs().max() / ref.abs().max()).item()
    return us, 2.0 * N_NODES * F * 4 * F / us * 1e-6, err


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--all-cfgs', action='store_true')
    ap.add_argument('--nodes', type=int, default=N_NODES, help='atoms of the batch (batch 4096: 67434, QMugs shape: 24300)')
    ap.add_argument('--edges', type=int, default=N_EDGES, help='directed bonds of the batch (batch 4096: 133440, QMugs: 49000)')
    ap.add_argument('--wide-sweep', action='store_true', help='with --all-cfgs: every tile configuration, more split counts')
    ap.add_argument('--no-rowseg', action='store_true')
    a = ap.parse_args()
    lib = L.load()
    shapes = [(n, ta, tb, {N_NODES: a.nodes, N_EDGES: a.edges}.get(M, M), N, {N_NODES: a.nodes, N_EDGES: a.edges}.get(K, K))
              for n, ta, tb, M, N, K in SHAPES]
    if a.nodes != N_NODES:
        shapes = [sh for sh in shapes if 'net3d' not in sh[0] and 'head' not in sh[0] and 'sim' not in sh[0] and '12F' not in sh[0]]
    for name, ta, tb, M, N, K in shapes:
        us, tf, err = run(lib, ta, tb, M, N, K, -1, 0)
        print(f'{name} M={M:6d} N={N:5d} K={K:6d}  auto: {us:8.1f} us {tf:7.1f} TF  err {err:.1e}', flush=True)
        if ta:
            us, tf, err = run(lib, ta, tb, M, N, K, -1, 0, scratch=False)
            print(f'{name} M={M:6d} N={N:5d} K={K:6d}  auto, fp32 atomics instead of scratch: {us:8.1f} us {tf:7.1f} TF', flush=True)
        if a.all_cfgs:
            wgrad = bool(ta)
            cfgs = ((3, 2, 8, 1) if wgrad else (2, 9, 11)) if not a.wide_sweep else ((3, 2, 8, 0, 5, 4, 7) if wgrad else (0, 2, 4, 5, 7, 9, 10, 11, 12))
            for cfg in cfgs:
                for splits in ((8, 16, 32, 64, 128, 256) if wgrad else (1,)):
                    us, tf, err = run(lib, ta, tb, M, N, K, cfg, splits)
                    print(f'      cfg {cfg:2d} splits {splits:2d}: {us:8.1f} us {tf:7.1f} TF  err {err:.1e}', flush=True)
    if a.no_rowseg:
        sys.exit(0)
    for cfg, seg in ((-1, 0), (3, 256), (3, 512), (3, 1024), (3, 2048), (2, 512), (2, 1024), (4, 512), (4, 1024), (4, 2048)):
        us, tf, err = run_rowseg(lib, cfg, seg)
        us2, _, _ = run_rowseg(lib, cfg, seg, scratch=False)
        print(f'rowseg wgrad 5 groups [F x 4F] cfg {cfg:2d} seg_rows {seg:4d}: {us:8.1f} us {tf:7.1f} TF  err {err:.1e}   '
              f'(atomics: {us2:.1f} us)', flush=True)
