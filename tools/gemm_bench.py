"""GPU microbenchmark of i3d_gemm_f32 on the GEMM shapes of the pre-training step (batch 512 QM9-shaped):
prints time and TFLOP/s per (shape, tile config, split-K).  Run on the MI355X box:
    python tools/gemm_bench.py [--all-cfgs]
"""
import argparse
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
    ('net3d  d[E3,20] W^T      ', 0, 1, 140000, 20, 20),
    ('head   r[512,600] W^T    ', 0, 1, 512, 200, 600),
    ('sim    z1 z2^T           ', 0, 1, 512, 512, 256),
]


def run(lib, ta, tb, M, N, K, cfg, splits, reps=20):
    dev = torch.device('cuda:0')
    A = torch.randn((K, M) if ta else (M, K), device=dev)
    B = torch.randn((N, K) if tb else (K, N), device=dev)
    C = torch.empty(M, N, device=dev)
    st = c_void_p(torch.cuda.current_stream().cuda_stream)
    args = (ta, tb, M, N, K, c_void_p(A.data_ptr()), A.shape[1], c_void_p(B.data_ptr()), B.shape[1],
            c_void_p(C.data_ptr()), N, None, 0, cfg, splits, st)
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


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--all-cfgs', action='store_true')
    a = ap.parse_args()
    lib = L.load()
    for name, ta, tb, M, N, K in SHAPES:
        us, tf, err = run(lib, ta, tb, M, N, K, -1, 0)
        print(f'{name} M={M:6d} N={N:5d} K={K:6d}  auto: {us:8.1f} us {tf:7.1f} TF  err {err:.1e}', flush=True)
        if a.all_cfgs:
            for cfg in (0, 2, 3, 4, 5):
                for splits in ((1, 2, 4) if K <= 2400 and M > 2000 else (4, 8, 16, 32)):
                    if splits > 1 and K < 1024:
                        continue
                    us, tf, err = run(lib, ta, tb, M, N, K, cfg, splits)
                    print(f'      cfg {cfg} splits {splits:2d}: {us:8.1f} us {tf:7.1f} TF  err {err:.1e}', flush=True)
