"""The row-panel GEMM (csrc/panel.hip) against the tiled kernels of csrc/gemm.hip on the chain's shapes at batch 512: error against the
fp64 product and time per launch (20 launches per event pair).   python tools/panel_bench.py"""
import ctypes
import importlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ops = importlib.import_module('3dinfomax_amd.ops')
_lib = importlib.import_module('3dinfomax_amd._lib')
L = _lib.load()
dev = torch.device('cuda:0')


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        a.record()
        for _ in range(reps):
            fn()
        b.record()
        torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) / reps * 1e3)
    return best


def p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def main():
    N0, E0 = 8409, 16638
    shapes = [('fwd FC2   [E,200]x[200,200]^T', E0, 200, 200, 1), ('dgrad FC2 [E,200]x[200,200]', E0, 200, 200, 0),
              ('fwd PL    [N,200]x[600,200]^T', N0, 600, 200, 1), ('dgrad DL  [N,600]x[600,200]', N0, 200, 600, 0),
              ('fwd post  [N,800]x[200,800]^T', N0, 200, 800, 1), ('dgrad post[N,200]x[200,800]', N0, 800, 200, 0)]
    st = ops._stream()
    print(f'{"shape":34s} {"tiled us":>9s} {"panel us":>9s} {"TF panel":>9s} {"err tiled":>10s} {"err panel":>10s}')
    for name, M, N, K, trans in shapes:
        torch.manual_seed(0)
        A = torch.randn(M, K, device=dev)
        W = (torch.randn(N, K, device=dev) if trans else torch.randn(K, N, device=dev)) * K ** -0.5
        bias = torch.randn(N, device=dev)
        ref = A.double() @ (W.double().T if trans else W.double()) + bias.double()
        packed = torch.empty(L.i3d_panel_packed_bytes(N, K), dtype=torch.uint8, device=dev)
        _lib.check(L.i3d_panel_pack(p(W), W.stride(0), N, K, trans, p(packed), st), 'pack')
        C = torch.empty(M, N, device=dev)
        run = lambda: _lib.check(L.i3d_panel_gemm(M, N, K, p(A), K, p(packed), p(C), N, p(bias), 0, st), 'panel')       # noqa: E731
        run()
        torch.cuda.synchronize()
        tiled = ops.gemm(A, W, trans_b=bool(trans), bias=bias)
        scale = ref.abs().max()
        e_t = float((tiled.double() - ref).abs().max() / scale)
        e_p = float((C.double() - ref).abs().max() / scale)
        us_t = timeit(lambda: ops.gemm(A, W, trans_b=bool(trans), bias=bias))
        us_p = timeit(run)
        # accumulate form
        C2 = tiled.clone()
        _lib.check(L.i3d_panel_gemm(M, N, K, p(A), K, p(packed), p(C2), N, None, 1, st), 'panel acc')
        e_acc = float((C2.double() - (2 * ref - bias.double())).abs().max() / scale)
        print(f'{name:34s} {us_t:9.2f} {us_p:9.2f} {2.0 * M * N * K / us_p / 1e6:9.1f} {e_t:10.2e} {e_p:10.2e}   acc err {e_acc:.1e}')


def fused():
    E0, F = 16638, 200
    torch.manual_seed(0)
    A, W, bias, aff = torch.randn(E0, F, device=dev), torch.randn(F, F, device=dev) * F ** -0.5, torch.randn(F, device=dev), torch.randn(3, F, device=dev)
    pk, _, _ = ops.panel_pack(W, True)
    us_t = timeit(lambda: ops.gemm_fused(A, W, bias, aff, None))
    us_p = timeit(lambda: ops.panel_gemm_fused(A, pk, F, bias, aff, None))
    us_pn = timeit(lambda: ops.panel_gemm_fused(A, pk, F, bias, aff, None, want_stats=False))
    us_pp = timeit(lambda: ops.panel_gemm(A, pk, F, bias))
    print(f'fwd FC2 fused (BN prologue + statistics) [E,200]x[200,200]^T: tiled {us_t:.2f} us, row-panel {us_p:.2f} us '
          f'(prologue only {us_pn:.2f}, plain {us_pp:.2f})')


if __name__ == '__main__':
    fused()
    main()
