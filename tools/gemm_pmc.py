"""Run one GEMM shape under a few tile configs (for rocprofv3 --pmc passes).  python tools/gemm_pmc.py"""
import importlib, os, sys
from ctypes import c_void_p
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
L = importlib.import_module('3dinfomax_amd._lib')
lib = L.load()
dev = torch.device('cuda:0')
M, N, K = 8320, 200, 2400
A = torch.randn(M, K, device=dev); B = torch.randn(N, K, device=dev); C = torch.empty(M, N, device=dev)
st = torch.cuda.current_stream().cuda_stream
for cfg in (4, 5, 8, 2, 7):
    for _ in range(3):
        lib.i3d_gemm_f32_ex(0, 1, M, N, K, A.data_ptr(), K, B.data_ptr(), K, C.data_ptr(), N, None, 0, cfg, 1, st)
torch.cuda.synchronize()
