import importlib, sys, time, torch
sys.path.insert(0, '/root/repo')
ops = importlib.import_module('3dinfomax_amd.ops'); L = importlib.import_module('3dinfomax_amd._lib').load()
dev = torch.device('cuda:0')
a = torch.zeros(1024, device=dev); b = torch.ones(1024, device=dev)
st = torch._C._cuda_getCurrentRawStream(0)
for n in (200, 2000):
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): L.i3d_add_inplace(a.data_ptr(), b.data_ptr(), 1024, st)
    t1 = time.perf_counter() - t; torch.cuda.synchronize(); t2 = time.perf_counter() - t
    print(f'raw ctypes launch x{n}: enqueue {t1/n*1e6:.2f} us/launch, incl. drain {t2/n*1e6:.2f}')
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(2000): torch.empty(1000, 200, device=dev)
print(f'torch.empty: {(time.perf_counter()-t)/2000*1e6:.2f} us')
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(2000): a.add_(1)
t1 = time.perf_counter() - t; torch.cuda.synchronize()
print(f'torch add_: enqueue {t1/2000*1e6:.2f} us')
x = torch.randn(8320, 200, device=dev); W = torch.randn(200, 200, device=dev)
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(500): ops.gemm(x, W, trans_b=True)
t1 = time.perf_counter() - t; torch.cuda.synchronize(); t2 = time.perf_counter() - t
print(f'ops.gemm: enqueue {t1/500*1e6:.2f} us, incl. drain {t2/500*1e6:.2f}')
