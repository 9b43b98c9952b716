"""Tile configurations of i3d_gemm_f32_ex on launches of few tiles (M = 512 .. 17 k rows, N = 64 .. 800, K = 200 / 600): where
the 32x32 tile of four 16x16 waves (cfg 8) beats the 64x64 / 64x32 tiles - the dispatch rule 'at most ~1000 32x32 tiles' of
csrc/gemm.hip.  Output: profiles/r03_small_gemm_sweep.txt.  Usage: python tools/small_gemm_sweep.py"""
import ctypes, importlib, os, sys
from ctypes import c_void_p
import torch
sys.path.insert(0, os.getcwd())
L = importlib.import_module('3dinfomax_amd._lib').load()
dev = torch.device('cuda:0')
ws = torch.empty(64 << 20, dtype=torch.uint8, device=dev)
def run(ta, tb, M, N, K, cfg, splits, reps=50):
    A = torch.randn((K, M) if ta else (M, K), device=dev); B = torch.randn((N, K) if tb else (K, N), device=dev)
    C = torch.empty(M, N, device=dev); st = c_void_p(torch.cuda.current_stream().cuda_stream)
    args = (ta, tb, M, N, K, c_void_p(A.data_ptr()), A.shape[1], c_void_p(B.data_ptr()), B.shape[1], c_void_p(C.data_ptr()), N, None, 0,
            cfg, splits, c_void_p(ws.data_ptr()), 64 << 20, st)
    for _ in range(3):
        if L.i3d_gemm_f32_ex(*args) != 0: return None
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): L.i3d_gemm_f32_ex(*args)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps
for tb in (1, 0):
  for N, K in ((200, 200), (200, 600), (256, 200), (600, 200), (800, 200), (64, 200)):
    for M in (512, 1024, 2048, 4096, 8704, 17408):
        res = [(f'c{c}', run(0, tb, M, N, K, c, 1)) for c in (-1, 9, 11, 3, 8)]
        print(f'tb{tb} M{M:6d} N{N:4d} K{K:4d} t64={((M+63)//64)*((N+63)//64):5d} ' + '  '.join(f'{k}:{v:.1f}' for k, v in res), flush=True)
