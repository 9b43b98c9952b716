"""native vs split fp32 products: accuracy against fp64 and time, on the chain's GEMM shapes"""
import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module('3dinfomax_amd.ops')
dev = torch.device('cuda:0')
torch.manual_seed(0)

def timeit(fn, n=100):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3

def err(C, ref):
    d = (C.double() - ref)
    return (d.abs().max() / ref.abs().max()).item(), (d.pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()).item()

N, E, F = 9216, 19400, 200
shapes = [('P  fwd [N,F]x[2F,F]^T', N, 2 * F, F, True), ('FC2 fwd [E,F]x[F,F]^T', E, F, F, True), ('post fwd [N,4F]x[F,4F]^T', N, F, 4 * F, True),
          ('dgrad-agg [N,F]x[F,4F]', N, 4 * F, F, False), ('dgrad FC2 [E,F]x[F,F]', E, F, F, False), ('dgrad merged [E,F]x[F,2F]', E, 2 * F, F, False),
          ('post12 fwd', N, F, 12 * F, True)]
for name, M, Nn, K, tb in shapes:
    A = torch.randn(M, K, device=dev)
    B = torch.randn((Nn, K) if tb else (K, Nn), device=dev) * K ** -0.5
    ref = A.double() @ (B.double().T if tb else B.double())
    out = torch.empty(M, Nn, device=dev)
    line = f'{name:28s}'
    for mode in ('native', 'split'):
        ops.set_fp32_products(mode)
        ops.gemm(A, B, trans_b=tb, out=out)
        e = err(out, ref)
        t = timeit(lambda: ops.gemm(A, B, trans_b=tb, out=out))
        line += f' | {mode} {t:6.1f} us max {e[0]:.2e} rms {e[1]:.2e}'
    ops.set_matmul_precision('bf16')
    t = timeit(lambda: ops.gemm(A, B, trans_b=tb, out=out))
    ops.set_matmul_precision('fp32')
    line += f' | bf16 {t:6.1f} us'
    print(line, flush=True)
# fused variants
for (M, Nn, K) in ((E, F, F), (N, F, 4 * F)):
    A = torch.randn(M, K, device=dev); W = torch.randn(Nn, K, device=dev) * K ** -0.5
    bias = torch.randn(Nn, device=dev); aff = torch.randn(3, K, device=dev); out = torch.empty(M, Nn, device=dev)
    Ad = A.double(); a = aff.double()
    refp = ((Ad - a[0]) * a[1] + a[2]) @ W.double().T + bias.double()
    refs = Ad @ W.double().T + bias.double()
    for mode in ('native', 'split'):
        ops.set_fp32_products(mode)
        r = {}
        ops.gemm_fused(A, W, bias, aff, None, want_stats=True, out=out); e1 = err(out, refp)
        ops.gemm_fused(A, W, bias, None, None, want_stats=True, out=out); e2 = err(out, refs)
        r['stats'] = timeit(lambda: ops.gemm_fused(A, W, bias, None, None, want_stats=True, out=out))
        r['both'] = timeit(lambda: ops.gemm_fused(A, W, bias, aff, None, want_stats=True, out=out))
        print(f'fused M={M} N={Nn} K={K} {mode}: stats {r["stats"]:6.1f} us (rms {e2[1]:.2e})  both {r["both"]:6.1f} us (rms {e1[1]:.2e})', flush=True)
ops.set_fp32_products('native')
