"""How the hand-written fp32 MFMA GEMM compares with the vendor library (torch.mm -> hipBLASLt / rocBLAS) on the step's
shapes.  python tools/blas_compare.py"""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module('3dinfomax_amd.ops')
dev = torch.device('cuda:0')
# (round 4: + the merged h-product [N,F]x[3F,F]^T, the grouped posttrans K = 5F, the merged data gradient [N,3F]x[3F,F])
SHAPES = [('fwd   [N,F]x[3F,F]^T (PL)', 0, 1, 9216, 600, 200), ('fwd   [N,5F]x[F,5F]^T', 0, 1, 9216, 200, 1000),
          ('dgrad [N,3F]x[3F,F]', 0, 0, 9216, 200, 600), ('fwd   [E,F]x[F,F]^T b512', 0, 1, 19100, 200, 200),
          ('fwd   [N,F]x[F,F]^T', 0, 1, 8320, 200, 200), ('fwd   [E,F]x[F,F]^T', 0, 1, 16640, 200, 200),
          ('fwd   [N,F]x[2F,F]^T (P)', 0, 1, 8320, 400, 200), ('fwd   [N,4F]x[F,4F]^T', 0, 1, 8320, 200, 800),
          ('dgrad [E,F]x[F,F]', 0, 0, 16640, 200, 200), ('dgrad [N,F]x[F,4F]', 0, 0, 8320, 800, 200),
          ('wgrad [F,F] K=N', 1, 0, 200, 200, 8320), ('wgrad [F,F] K=E', 1, 0, 200, 200, 16640),
          ('wgrad [2F,F] K=N', 1, 0, 400, 200, 8320), ('wgrad [F,4F] K=N', 1, 0, 200, 800, 8320)]


def timeit(fn, reps=30):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


for name, ta, tb, M, N, K in SHAPES:
    A = torch.randn((K, M) if ta else (M, K), device=dev)
    B = torch.randn((N, K) if tb else (K, N), device=dev)
    C = torch.empty(M, N, device=dev)
    mine = timeit(lambda: ops.gemm(A, B, trans_a=bool(ta), trans_b=bool(tb), out=C))
    At, Bt = (A.t() if ta else A), (B.t() if tb else B)
    lib = timeit(lambda: torch.mm(At, Bt, out=C))
    fl = 2.0 * M * N * K
    print(f'{name:28s} M={M:6d} N={N:4d} K={K:6d}   this repo {mine:7.1f} us {fl / mine * 1e-6:6.1f} TF   '
          f'torch.mm {lib:7.1f} us {fl / lib * 1e-6:6.1f} TF', flush=True)
