"""Time the fused-BatchNorm GEMM variants against the plain forward GEMM on the step's shapes.
    python tools/fused_gemm_bench.py"""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module('3dinfomax_amd.ops')


def timeit(fn, n=60):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


def main():
    dev = torch.device('cuda:0')
    for (M, N, K) in ((16907, 200, 200), (8511, 200, 800), (8511, 200, 200)):
        A = torch.randn(M, K, device=dev)
        W = torch.randn(N, K, device=dev) * K ** -0.5
        bias = torch.randn(N, device=dev)
        aff = torch.randn(3, K, device=dev)
        out = torch.empty(M, N, device=dev)
        fl = 2.0 * M * N * K
        res = {}
        res['plain'] = timeit(lambda: ops.gemm(A, W, trans_b=True, bias=bias, out=out))
        res['prologue'] = timeit(lambda: ops.gemm_fused(A, W, bias, aff, None, want_stats=False, out=out))
        res['stats'] = timeit(lambda: ops.gemm_fused(A, W, bias, None, None, want_stats=True, out=out))
        res['both'] = timeit(lambda: ops.gemm_fused(A, W, bias, aff, None, want_stats=True, out=out))
        print(f'M={M} N={N} K={K}: ' + '  '.join(f'{k} {v:6.1f} us ({fl / v / 1e6:5.1f} TF)' for k, v in res.items()), flush=True)
    # statistics finalisation
    for tiles, F in ((423, 200), (265, 200), (134, 200)):
        partial = torch.rand(tiles, 3, F, device=dev) + 1
        g, b = torch.ones(F, device=dev), torch.zeros(F, device=dev)
        t = timeit(lambda: ops.bn_finalize_partials(partial, tiles, F, 1e-5, 0.1, g, b))
        print(f'finalize tiles={tiles} F={F}: {t:5.1f} us (incl. 3 torch.empty)')


if __name__ == '__main__':
    main()
