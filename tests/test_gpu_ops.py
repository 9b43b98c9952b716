"""-m gpu: every HIP kernel, called through the C ABI (3dinfomax_amd/ops.py -> ctypes -> lib3dinfomax_hip.so),
against plain fp32/fp64 torch on the CPU or the oracle.  Tolerances are written next to each check:
the bar of BASELINE.json:north_star is 1e-4 relative fp32; the kernels are held to ~1e-5."""
import importlib
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from helpers import rel_err
from oracle import pna3d_oracle as O

pytestmark = pytest.mark.gpu

ops = None
DEV = None


@pytest.fixture(scope='module', autouse=True)
def _gpu():
    global ops, DEV
    assert torch.cuda.is_available(), 'these tests need the MI355X'
    ops = importlib.import_module('3dinfomax_amd.ops')
    DEV = torch.device('cuda:0')
    yield


def g(t):
    return t.to(DEV)


def rnd(*shape, seed=0, scale=1.0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale


# ---- GEMM ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('M,N,K', [(64, 200, 200), (1000, 200, 600), (777, 200, 2600), (513, 400, 200), (300, 256, 200),
                                   (5000, 20, 9), (5000, 20, 60), (33, 8, 7), (1, 1, 1), (129, 208, 17)])
@pytest.mark.parametrize('ta,tb', [(False, True), (False, False), (True, False), (True, True)])
def test_gemm_matches_fp64(M, N, K, ta, tb):
    A = rnd(*((K, M) if ta else (M, K)), seed=1)
    B = rnd(*((N, K) if tb else (K, N)), seed=2)
    bias = rnd(N, seed=3)
    ref = (A.double().T if ta else A.double()) @ (B.double().T if tb else B.double()) + bias.double()
    out = ops.gemm(g(A), g(B), trans_a=ta, trans_b=tb, bias=g(bias))
    # fp32 fmaf chain vs fp64: error ~ 1e-7 * sum|a*b|  (MI355X guide, section 3)
    tol = 3e-6 * (A.abs().max() * B.abs().max() * K).item() / max(ref.abs().max().item(), 1e-30)
    assert rel_err(out.cpu(), ref) < max(tol, 1e-5)


def _rms_rel(out, ref):
    return ((out.double() - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()).item()


@pytest.mark.parametrize('M,N,K,tb', [(9216, 400, 200, True), (9216, 200, 800, True), (9216, 800, 200, False), (19400, 200, 200, False)])
def test_split_products_are_fp32_products(M, N, K, tb):
    """fp32 mode, the default way the tiled GEMMs form a product (csrc/gemm.hip, i3d_set_fp32_products(1)): both operands split
    exactly into three bf16 parts, the six part products of order <= 2 on the bf16 matrix pipe, fp32 accumulation.  Held to:
    against the fp64 product its error is NOT larger than that of v_mfma_f32_32x32x2_f32 on the same operands (measured: 0.8x,
    2.1e-7 against 2.5e-7 rms at K = 200) - an fp32 GEMM in another summation order, not a reduced-precision one; the bf16
    matmul mode on the same operands is four orders of magnitude away.  Forward and data-gradient layouts, the fused
    (BatchNorm prologue / statistics) variant; reference op: nn.Linear, models/base_layers.py:101"""
    A, B = rnd(M, K, seed=1), rnd(*((N, K) if tb else (K, N)), seed=2) * K ** -0.5
    ref = A.double() @ (B.double().T if tb else B.double())
    res = {}
    prev = ops.get_fp32_products()
    try:
        for mode in ('native', 'split'):
            ops.set_fp32_products(mode)
            res[mode] = ops.gemm(g(A), g(B), trans_b=tb).cpu()
        ops.set_matmul_precision('bf16')
        res['bf16'] = ops.gemm(g(A), g(B), trans_b=tb).cpu()
    finally:
        ops.set_matmul_precision('fp32')
        ops.set_fp32_products(prev)
    e = {k: _rms_rel(v, ref) for k, v in res.items()}
    assert not torch.equal(res['native'], res['split'])              # (the switch does switch)
    assert e['split'] <= 1.05 * e['native'] and e['split'] < 1e-6, e
    assert rel_err(res['split'], ref) <= 1.25 * rel_err(res['native'], ref), (rel_err(res['split'], ref), rel_err(res['native'], ref))
    assert e['bf16'] > 1000 * e['split'], e


def _panel_pack(W, N, K, trans):
    import ctypes
    _lib = importlib.import_module('3dinfomax_amd._lib')
    L = _lib.load()
    packed = torch.empty(L.i3d_panel_packed_bytes(N, K), dtype=torch.uint8, device=DEV)
    _lib.check(L.i3d_panel_pack(ctypes.c_void_p(W.data_ptr()), W.stride(0), N, K, trans, ctypes.c_void_p(packed.data_ptr()), ops._stream()),
               'i3d_panel_pack')
    return packed


@pytest.mark.parametrize('M,N,K,trans', [(16638, 200, 200, 1), (16638, 200, 200, 0), (8409, 600, 200, 1), (8409, 200, 600, 0), (8409, 800, 200, 0),
                                         (1, 200, 200, 1), (63, 4, 8, 1), (65, 212, 40, 0), (300, 416, 72, 1), (129, 92, 256, 0)])
def test_row_panel_gemm_against_fp64_and_the_tiled_kernels(M, N, K, trans):
    """csrc/panel.hip: C (+)= A op(W) (+ bias) from the weight packed once into its three bf16 images, A read and split once per
    208-column block - the split-product arithmetic of the tiled kernels (nn.Linear forward and data gradient, reference
    models/base_layers.py:101): against the fp64 product not further away than the tiled split form; ragged row / column / K tails,
    the accumulate form, a wider output row pitch."""
    import ctypes
    _lib = importlib.import_module('3dinfomax_amd._lib')
    L = _lib.load()
    p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None      # noqa: E731
    A = g(rnd(M, K, seed=1))
    W = g((rnd(N, K, seed=2) if trans else rnd(K, N, seed=2)) * K ** -0.5)
    bias = g(rnd(N, seed=3))
    ref = A.double() @ (W.double().T if trans else W.double()) + bias.double()
    packed = _panel_pack(W, N, K, trans)
    ldc = N + 8
    C = torch.full((M, ldc), 7.0, device=DEV)
    _lib.check(L.i3d_panel_gemm(M, N, K, p(A), K, p(packed), p(C), ldc, p(bias), 0, ops._stream()), 'i3d_panel_gemm')
    assert torch.all(C[:, N:] == 7.0)
    prev = ops.get_fp32_products()
    try:
        ops.set_fp32_products('split')
        tiled = ops.gemm(A, W, trans_b=bool(trans), bias=bias)
    finally:
        ops.set_fp32_products(prev)
    scale = float(ref.abs().max())
    e_panel, e_tiled = float((C[:, :N].double() - ref).abs().max()) / scale, float((tiled.double() - ref).abs().max()) / scale
    assert e_panel < 2e-6 and e_panel <= 1.5 * e_tiled + 1e-7, (e_panel, e_tiled)
    C2 = C.clone()
    _lib.check(L.i3d_panel_gemm(M, N, K, p(A), K, p(packed), p(C2), ldc, None, 1, ops._stream()), 'i3d_panel_gemm')
    assert float((C2[:, :N].double() - (2 * ref - bias.double())).abs().max()) / scale < 4e-6 and torch.all(C2[:, N:] == 7.0)
    torch.cuda.synchronize()


@pytest.mark.parametrize('M,N,K,act', [(16638, 200, 200, None), (5000, 200, 200, 'relu'), (777, 64, 40, 'leakyrelu'), (31, 200, 200, None)])
def test_row_panel_gemm_fused_with_batchnorm_prologue_and_statistics(M, N, K, act):
    """i3d_panel_gemm_fused against i3d_gemm_f32_fused (csrc/gemm.hip FUSE 3): the BatchNorm-apply prologue on A, bias, activation,
    and per-tile column statistics of the stored values - 32-row tiles here, 64-row tiles there; the finalised mean / invstd (and
    aff) of the two agree to rounding, the outputs to the split form's rounding."""
    import ctypes
    _lib = importlib.import_module('3dinfomax_amd._lib')
    L = _lib.load()
    p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None      # noqa: E731
    A, W, bias, aff = g(rnd(M, K, seed=1) + 0.7), g(rnd(N, K, seed=2) * K ** -0.5), g(rnd(N, seed=3)), g(rnd(3, K, seed=4))
    gamma, beta = g(rnd(N, seed=5) * 0.2 + 1), g(rnd(N, seed=6) * 0.2)
    a = aff.double()
    acts = {None: lambda t: t, 'relu': F.relu, 'leakyrelu': F.leaky_relu}
    ref = acts[act](((A.double() - a[0]) * a[1] + a[2]) @ W.double().T + bias.double())
    prev = ops.get_fp32_products()
    try:
        ops.set_fp32_products('split')
        out_t, part_t, tiles_t = ops.gemm_fused(A, W, bias, aff, act)
    finally:
        ops.set_fp32_products(prev)
    packed = _panel_pack(W, N, K, 1)
    tiles = L.i3d_panel_stats_tiles(M)
    out_p = torch.empty(M, N, device=DEV)
    part_p = torch.full((tiles, 3, N), float('nan'), device=DEV)
    _lib.check(L.i3d_panel_gemm_fused(M, N, K, p(A), K, p(packed), p(out_p), N, p(bias), p(aff), _lib.ACT[act], p(part_p), ops._stream()),
               'i3d_panel_gemm_fused')
    scale = float(ref.abs().max())
    assert float((out_p.double() - ref).abs().max()) / scale < 2e-6
    assert float((out_p - out_t).abs().max()) / scale < 2e-6
    assert not torch.isnan(part_p).any() and float(part_p[:, 2, 0].sum()) == M
    m_t, i_t, aff_t = ops.bn_finalize_partials(part_t, tiles_t, N, 1e-5, 0.1, gamma, beta)
    m_p, i_p, aff_p = ops.bn_finalize_partials(part_p, tiles, N, 1e-5, 0.1, gamma, beta)
    ref_mean, ref_var = ref.mean(0), ref.var(0, unbiased=False)
    assert rel_err(m_p.cpu(), ref_mean.float().cpu()) < 1e-5 and rel_err(i_p.cpu(), (1 / torch.sqrt(ref_var + 1e-5)).float().cpu()) < 2e-5
    assert rel_err(m_p.cpu(), m_t.cpu()) < 2e-6 and rel_err(i_p.cpu(), i_t.cpu()) < 5e-6 and rel_err(aff_p.cpu(), aff_t.cpu()) < 5e-6


def test_split_products_with_non_finite_operands():
    """What include/infomax3d_hip.h states about the split form (the library default) on operands an fp32 product handles and a
    three-part bf16 split does not: +-Inf and |x| > 3.39e38 (bf16 RNE of `hi` overflows) leave a NaN remainder - the outputs that
    depend on such an operand are NON-FINITE in exactly the positions where v_mfma_f32_32x32x2_f32 gives a non-finite value (Inf
    there, NaN here), every other output is untouched; values up to 3e38 split exactly."""
    M, N, K = 512, 200, 200
    A, B = rnd(M, K, seed=5), rnd(N, K, seed=6) * K ** -0.5
    A[7, 13] = float('inf')
    A[100, 0] = -float('inf')
    A[300, 199] = 3.4e38            # finite in fp32, beyond the largest bf16
    A[400, 5] = 3.0e38              # splits exactly
    B[9, 5] = 0.0                   # 3e38 x 0 stays finite
    res = {}
    prev = ops.get_fp32_products()
    try:
        for mode in ('native', 'split'):
            ops.set_fp32_products(mode)
            res[mode] = ops.gemm(g(A), g(B), trans_b=True).cpu()
    finally:
        ops.set_fp32_products(prev)
    bad_n, bad_s = ~torch.isfinite(res['native']), ~torch.isfinite(res['split'])
    # rows 7 / 100 / 300 are non-finite in both forms (row 300: 3.4e38 x b overflows or not by b in fp32, always in the split form)
    assert bad_s[7].all() and bad_s[100].all() and bad_n[7].all() and bad_n[100].all()
    assert (bad_n <= bad_s).all()                               # non-finite in fp32 => non-finite in the split form
    rows = torch.ones(M, dtype=torch.bool)
    rows[[7, 100, 300]] = False
    assert not bad_s[rows].any()
    ref = A[rows].double() @ B.double().T
    keep = torch.isfinite(ref.float())
    assert rel_err(torch.where(keep, res['split'][rows].double(), ref), ref) < 1e-6
    assert torch.isfinite(res['split'][400, 9])


def test_split_products_in_the_fused_gemm():
    """the same statement for the fused forward GEMM (BatchNorm prologue + statistics epilogue).  (The one-launch weight gradients
    and the small-tile / unaligned kernels keep the fp32 matrix pipe: the six-product form of the panel kernel was measured slower,
    87.7 -> 95.1 us per layer, profiles/r05_split_products.txt)"""
    M, N, K = 6000, 200, 200
    A, W, bias, aff = rnd(M, K, seed=1), rnd(N, K, seed=2) * K ** -0.5, rnd(N, seed=3), rnd(3, K, seed=4)
    a = aff.double()
    ref = ((A.double() - a[0]) * a[1] + a[2]) @ W.double().T + bias.double()
    e, stats = {}, {}
    prev = ops.get_fp32_products()
    try:
        for mode in ('native', 'split'):
            ops.set_fp32_products(mode)
            out, st = ops.gemm_fused(g(A), g(W), g(bias), g(aff), None, want_stats=True)[:2]
            e[mode] = _rms_rel(out.cpu(), ref)
            stats[mode] = st.cpu()
    finally:
        ops.set_fp32_products(prev)
    assert e['split'] <= 1.05 * e['native'] and e['split'] < 1e-6, e
    assert torch.allclose(stats['native'], stats['split'], rtol=1e-4, atol=1e-4)


def test_gemm_weight_grad_split_k_and_strided_views():
    rows, Fo, Fi = 20000, 200, 2600
    dY, X = rnd(rows, Fo, seed=4, scale=0.1), rnd(rows, Fi, seed=5)
    ref = dY.double().T @ X.double()
    out = ops.gemm(g(dY), g(X), trans_a=True)                 # split-K with atomics
    assert rel_err(out.cpu(), ref) < 2e-5
    # column-slice views as operands and as output (leading dimension != width), accumulate on top
    W = rnd(Fo, Fi, seed=6)
    Wg = g(W)
    h, agg = rnd(1000, 200, seed=7), rnd(1000, 2400, seed=8)
    y = ops.gemm(g(h), Wg[:, :200], trans_b=True)
    ops.gemm(g(agg), Wg[:, 200:], trans_b=True, out=y, accumulate=True)
    ref = torch.cat([h, agg], 1).double() @ W.double().T
    assert rel_err(y.cpu(), ref) < 1e-5
    gW = torch.zeros(Fo, Fi, device=DEV)
    ops.gemm(g(dY[:1000]), g(h), trans_a=True, out=gW[:, :200])
    ops.gemm(g(dY[:1000]), g(agg), trans_a=True, out=gW[:, 200:])
    ref = dY[:1000].double().T @ torch.cat([h, agg], 1).double()
    assert rel_err(gW.cpu(), ref) < 2e-5


@pytest.mark.parametrize('rows', [2100, 4200, 530])
def test_gemm_split_k_into_strided_column_block(rows):
    """weight gradient of posttrans written into a column block of dW (leading dimension != width) with split-K:
    the zero-fill + atomics path must only touch - and fully initialise - its own block."""
    Fo, Fa, Fc = 200, 200, 2400
    dY, a, c = rnd(rows, Fo, seed=1, scale=0.3), rnd(rows, Fa, seed=2), rnd(rows, Fc, seed=3)
    gW = torch.full((Fo, Fa + Fc), 7.0, device=DEV)           # poisoned: stale values must not survive
    ops.gemm(g(dY), g(c), trans_a=True, out=gW[:, Fa:])
    ops.gemm(g(dY), g(a), trans_a=True, out=gW[:, :Fa])
    ref = dY.double().T @ torch.cat([a, c], 1).double()
    assert rel_err(gW.cpu(), ref) < 2e-5
    # the same through the tuning entry with forced split factors
    L = importlib.import_module('3dinfomax_amd._lib')
    from ctypes import c_void_p
    lib = L.load()
    ws = torch.empty(64 << 20, dtype=torch.uint8, device=DEV)
    for cfg, splits, scratch in ((4, 4, 0), (3, 16, 0), (2, 2, 0), (0, 7, 0), (3, 16, 64 << 20), (2, 5, 64 << 20), (8, 3, 64 << 20),
                                 (3, 16, 1 << 20)):    # the last one: scratch too small -> atomics
        out = torch.full((Fo, Fa + Fc), 7.0, device=DEV)
        A, B = g(dY), g(c)
        rc = lib.i3d_gemm_f32_ex(1, 0, Fo, Fc, rows, c_void_p(A.data_ptr()), Fo, c_void_p(B.data_ptr()), Fc,
                                 c_void_p(out[:, Fa:].data_ptr()), Fa + Fc, None, 0, cfg, splits,
                                 c_void_p(ws.data_ptr()) if scratch else None, scratch,
                                 c_void_p(torch.cuda.current_stream().cuda_stream))
        assert rc == 0
        assert rel_err(out[:, Fa:].cpu(), ref[:, Fa:]) < 2e-5, (cfg, splits, scratch)
        assert torch.all(out[:, :Fa] == 7.0)
        if scratch >= (64 << 20):        # two-stage reduction: bit-identical from run to run, accumulate adds on top
            out2 = torch.full((Fo, Fa + Fc), 7.0, device=DEV)
            lib.i3d_gemm_f32_ex(1, 0, Fo, Fc, rows, c_void_p(A.data_ptr()), Fo, c_void_p(B.data_ptr()), Fc,
                                c_void_p(out2[:, Fa:].data_ptr()), Fa + Fc, None, 0, cfg, splits, c_void_p(ws.data_ptr()), scratch,
                                c_void_p(torch.cuda.current_stream().cuda_stream))
            assert torch.equal(out, out2)
            lib.i3d_gemm_f32_ex(1, 0, Fo, Fc, rows, c_void_p(A.data_ptr()), Fo, c_void_p(B.data_ptr()), Fc,
                                c_void_p(out2[:, Fa:].data_ptr()), Fa + Fc, None, 1, cfg, splits, c_void_p(ws.data_ptr()), scratch,
                                c_void_p(torch.cuda.current_stream().cuda_stream))
            assert rel_err(out2[:, Fa:].cpu(), 2 * ref[:, Fa:]) < 2e-5


@pytest.mark.parametrize('scratch', [True, False])
def test_two_block_gemms_of_the_edge_mlp(scratch):
    """P = h [W_s | W_d]^T, dh = dP [W_s; W_d] and d[W_s | W_d] = dP^T h, each as ONE GEMM over the two column blocks of the
    pretrans weight (i3d_gemm_f32_blocks) against the two-GEMM formulation."""
    from ctypes import c_void_p
    L = importlib.import_module('3dinfomax_amd._lib')
    lib = L.load()
    st = c_void_p(torch.cuda.current_stream().cuda_stream)
    N, Fh, Fo, ldw = 3000, 200, 200, 600
    h, W, dP = rnd(N, Fh, seed=1), rnd(Fo, ldw, seed=2), rnd(N, 2 * Fo, seed=3)
    hg, Wg, dPg = g(h), g(W), g(dP)
    ws = torch.empty(32 << 20, dtype=torch.uint8, device=DEV)
    wsp, wsb = (c_void_p(ws.data_ptr()), 32 << 20) if scratch else (None, 0)
    delta, view = Fh - Fo * ldw, (Fo - 1) * ldw + 2 * Fh
    P = torch.empty(N, 2 * Fo, device=DEV)
    assert lib.i3d_gemm_f32_blocks(0, 1, N, 2 * Fo, Fh, c_void_p(hg.data_ptr()), Fh, c_void_p(Wg.data_ptr()), ldw, Fo, delta, view,
                                   c_void_p(P.data_ptr()), 2 * Fo, 0, 0, 0, None, 0, st) == 0
    ref = torch.cat([h.double() @ W[:, :Fh].double().T, h.double() @ W[:, Fh:2 * Fh].double().T], 1)
    assert rel_err(P.cpu(), ref) < 1e-5
    gW = torch.full((Fo, ldw), 7.0, device=DEV)
    assert lib.i3d_gemm_f32_blocks(1, 0, 2 * Fo, Fh, N, c_void_p(dPg.data_ptr()), 2 * Fo, c_void_p(hg.data_ptr()), Fh, 0, 0, 0,
                                   c_void_p(gW.data_ptr()), ldw, Fo, delta, 0, wsp, wsb, st) == 0
    ref = torch.cat([dP[:, :Fo].double().T @ h.double(), dP[:, Fo:].double().T @ h.double()], 1)
    assert rel_err(gW[:, :2 * Fh].cpu(), ref) < 2e-5
    assert torch.all(gW[:, 2 * Fh:] == 7.0)
    gh = torch.empty(N, Fh, device=DEV)
    assert lib.i3d_gemm_f32_blocks(0, 0, N, Fh, 2 * Fo, c_void_p(dPg.data_ptr()), 2 * Fo, c_void_p(Wg.data_ptr()), ldw, Fo, delta, view,
                                   c_void_p(gh.data_ptr()), Fh, 0, 0, 0, None, 0, st) == 0
    ref = dP[:, :Fo].double() @ W[:, :Fh].double() + dP[:, Fo:].double() @ W[:, Fh:2 * Fh].double()
    assert rel_err(gh.cpu(), ref) < 1e-5


def test_degree_grouped_posttrans_gemms():
    """gemm_grouped / gemm_rowsubset / combine_weights against the reference-shaped computation
    [a | amp(D) a | att(D) a] W_agg^T with per-node scalers."""
    graph = importlib.import_module('3dinfomax_amd.graph')
    n, F_out, A, S = 1500, 200, 800, 3
    rng = np.random.default_rng(0)
    indeg = rng.choice([0, 1, 2, 3, 4, 6], size=n, p=[0.03, 0.45, 0.1, 0.1, 0.3, 0.02])
    rows, tiles, groups = graph.group_nodes_by_degree(indeg)
    assert rows.shape[0] % 64 == 0 and sorted(rows[rows >= 0].tolist()) == np.nonzero(indeg > 0)[0].tolist()
    coef = [[1.0, float(np.float32(math.log(D + 1))), float(np.float32(1 / math.log(D + 1)))] for D, _, _ in groups]
    flat = [c for gco in coef for c in gco]
    a, W = rnd(n, A, seed=1), rnd(F_out, 200 + S * A, seed=2, scale=0.05)
    a[indeg == 0] = 0
    amp = torch.tensor([math.log(d + 1) if d > 0 else 0.0 for d in indeg], dtype=torch.float32)[:, None]
    att = torch.tensor([1 / math.log(d + 1) if d > 0 else 0.0 for d in indeg], dtype=torch.float32)[:, None]
    agg12 = torch.cat([a, a * amp, a * att], 1)
    ref = agg12.double() @ W[:, 200:].double().T
    WD = ops.combine_weights_fwd(g(W), 200, A, flat, len(groups), S)
    out = torch.zeros(n, F_out, device=DEV)
    ops.gemm_grouped(g(a), g(torch.from_numpy(rows)), g(torch.from_numpy(tiles)), WD, out, trans_b=True, accumulate=True)
    assert rel_err(out.cpu(), ref) < 1e-5
    # data gradient: d a = dY W_D  (only rows with D > 0 are written)
    dY = rnd(n, F_out, seed=3)
    ga = torch.zeros(n, A, device=DEV)
    ops.gemm_grouped(g(dY), g(torch.from_numpy(rows)), g(torch.from_numpy(tiles)), WD, ga, trans_b=False, accumulate=False)
    Wagg = W[:, 200:].double()
    ref_ga = dY.double() @ Wagg[:, :A] + amp.double() * (dY.double() @ Wagg[:, A:2 * A]) + att.double() * (dY.double() @ Wagg[:, 2 * A:])
    ref_ga[indeg == 0] = 0
    assert rel_err(ga.cpu(), ref_ga) < 1e-5
    # weight gradient: per-group dW_D folded back into the three scaler blocks
    gWD = torch.empty_like(WD)
    rows_d = g(torch.from_numpy(rows))
    for gi, (_, start, count) in enumerate(groups):
        ops.gemm_rowsubset(g(dY), g(a), rows_d[start:start + count], gWD[gi])
    # ... and all groups in one launch (several tile configurations / segment lengths)
    for cfg, seg in ((-1, 0), (3, 128), (4, 64), (2, 2048)):
        gWD2 = torch.full_like(WD, 3.0)
        for use_ws in (True, False):     # scratch + fixed-order reduction / fp32 atomics
            gWD2 = torch.full_like(WD, 3.0)
            ops.gemm_rowsubset_multi(g(dY), g(a), rows_d, [st for _, st, _ in groups], [ct for _, _, ct in groups], gWD2,
                                     tile_cfg=cfg, seg_rows=seg, use_workspace=use_ws)
            assert rel_err(gWD2.cpu(), gWD.cpu()) < 2e-6, (cfg, seg, use_ws)
    gW = torch.full((F_out, 200 + S * A), 7.0, device=DEV)
    ops.combine_weights_bwd(gWD, gW, 200, A, flat, len(groups), S)
    ref_gW = dY.double().T @ agg12.double()
    assert rel_err(gW[:, 200:].cpu(), ref_gW) < 2e-5
    assert torch.all(gW[:, :200] == 7.0)


# ---- K4 aggregation --------------------------------------------------------------------------------------
def _random_csr(n, max_deg, seed, zero_frac=0.1):
    rng = np.random.default_rng(seed)
    deg = rng.integers(1, max_deg + 1, size=n)
    deg[rng.random(n) < zero_frac] = 0
    dst = np.repeat(np.arange(n), deg)
    ptr = np.zeros(n + 1, dtype=np.int32)
    ptr[1:] = np.cumsum(deg)
    return torch.from_numpy(dst), torch.from_numpy(ptr)


@pytest.mark.parametrize('feat', [200, 20, 7])
@pytest.mark.parametrize('aggs,scalers', [(['mean', 'max', 'min', 'std'], ['identity', 'amplification', 'attenuation']),
                                         (['sum', 'var', 'max'], ['identity']),
                                         (['mean', 'std'], ['attenuation', 'identity'])])
def test_pna_aggregate_fwd_bwd_vs_oracle(feat, aggs, scalers):
    n = 300
    dst, ptr = _random_csr(n, 6, seed=feat)
    E = dst.shape[0]
    e = rnd(E, feat, seed=11)
    # exact ties: duplicate the first message of every node with degree >= 2 into its second slot
    for v in range(n):
        if ptr[v + 1] - ptr[v] >= 2:
            e[ptr[v] + 1] = e[ptr[v]]
    e_ref = e.clone().requires_grad_(True)
    n_out = len(aggs) * (len(scalers) if len(scalers) > 1 else 1) * feat
    ref = O.degree_bucketed_reduce(e_ref, dst, n, lambda mb, D: O.pna_reduce(mb, D, aggs, scalers), n_out)
    cot = rnd(n, n_out, seed=12)
    (ref * cot).sum().backward()
    ac, sc = ops.agg_codes(aggs), ops.scaler_codes(scalers)
    out = ops.pna_aggregate_fwd(g(e), g(ptr), n, ac, sc)
    assert rel_err(out.cpu(), ref.detach()) < 1e-6
    ge = ops.pna_aggregate_bwd(g(cot), g(e), g(ptr), n, ac, sc)
    # first-index tie routing as torch CPU; std gradient re-associated: 1e-5
    assert rel_err(ge.cpu(), e_ref.grad) < 1e-5


def test_readout_fwd_bwd():
    sizes = [5, 1, 18, 29, 3, 9]
    n, feat = sum(sizes), 200
    x = rnd(n, feat, seed=20)
    x[6 + 3] = x[6 + 1]      # tie inside graph 2
    ptr = torch.tensor(np.concatenate([[0], np.cumsum(sizes)]), dtype=torch.int32)
    names = ['min', 'max', 'mean', 'sum']
    xr = x.clone().requires_grad_(True)
    ref = torch.cat([O.segment_readout(xr, sizes, op) for op in names], -1)
    cot = rnd(len(sizes), 4 * feat, seed=21)
    (ref * cot).sum().backward()
    codes = ops.agg_codes(names)
    out = ops.segment_readout_fwd(g(x), g(ptr), len(sizes), codes)
    assert rel_err(out.cpu(), ref.detach()) < 1e-6
    gx = ops.segment_readout_bwd(g(cot), g(x), g(ptr), len(sizes), codes)
    assert rel_err(gx.cpu(), xr.grad) < 1e-6


# ---- K1 embedding ----------------------------------------------------------------------------------------
@pytest.mark.parametrize('rows,multihot', [(1000, True), (9000, True), (60, True), (1000, False)])
def test_embedding_sum_fwd_bwd_with_row_perm(rows, multihot, monkeypatch):
    monkeypatch.setattr(ops, 'MULTIHOT_EMB_BWD', multihot)     # multi-hot GEMM (deterministic) / LDS-atomics kernel
    dims, feat = [119, 5, 12, 12, 10, 6, 6, 2, 2], 200
    gen = torch.Generator().manual_seed(3)
    idx = torch.stack([torch.randint(0, d, (rows,), generator=gen) for d in dims], 1)
    tabs = [rnd(d, feat, seed=30 + i).requires_grad_(True) for i, d in enumerate(dims)]
    perm = torch.randperm(rows, generator=gen)
    ref = sum(F.embedding(idx[perm][:, k], tabs[k]) for k in range(len(dims)))
    cot = rnd(rows, feat, seed=40)
    (ref * cot).sum().backward()
    out = ops.embedding_sum_fwd(g(idx), [g(t.detach()) for t in tabs], g(perm.int()))
    assert rel_err(out.cpu(), ref.detach()) < 1e-6
    grads = ops.embedding_sum_bwd(g(idx), g(cot), dims, g(perm.int()))
    for gt, t in zip(grads, tabs):
        assert rel_err(gt.cpu(), t.grad) < 1e-5     # summation order differs
    if multihot:                                    # the GEMM path is deterministic
        again = ops.embedding_sum_bwd(g(idx), g(cot), dims, g(perm.int()))
        assert all(torch.equal(a, b) for a, b in zip(grads, again))


def test_padding_encoders_match_the_reference_module():
    """AtomEncoder / BondEncoder(padding=True), reference commons/mol_encoder.py:12-42, 47-73: one extra row per table with
    padding_idx 0, xavier over the whole table, features looked up at x + 1 (-1 -> row 0), no gradient for row 0"""
    amd = importlib.import_module('3dinfomax_amd')
    for cls, list_name in ((amd.AtomEncoder, 'atom_embedding_list'), (amd.BondEncoder, 'bond_embedding_list')):
        torch.manual_seed(4)
        enc = cls(emb_dim=40, padding=True).to(DEV)
        tabs = getattr(enc, list_name)
        dims = [e.num_embeddings - 1 for e in tabs]
        assert all(e.padding_idx == 0 for e in tabs) and enc.dims == [d + 1 for d in dims]
        gen = torch.Generator().manual_seed(8)
        x = torch.stack([torch.randint(-1, d, (700,), generator=gen) for d in dims], 1)      # -1: missing feature
        ref_tabs = [e.weight.detach().cpu().clone().requires_grad_(True) for e in tabs]
        ref = sum(F.embedding(x[:, k] + 1, ref_tabs[k], padding_idx=0) for k in range(len(dims)))
        cot = rnd(700, 40, seed=41)
        (ref * cot).sum().backward()
        out = enc(g(x))
        (out * g(cot)).sum().backward()
        assert rel_err(out.detach().cpu(), ref.detach()) < 1e-6
        for e, t in zip(tabs, ref_tabs):
            assert torch.all(e.weight.grad[0] == 0) and rel_err(e.weight.grad.cpu(), t.grad) < 1e-5


# ---- BatchNorm -------------------------------------------------------------------------------------------
@pytest.mark.parametrize('rows,feat', [(5000, 200), (300, 20), (17, 7), (70000, 20)])
@pytest.mark.parametrize('act,post', [('relu', None), (None, None), ('silu', 'silu'), ('leakyrelu', None), ('sigmoid', 'leakyrelu')])
def test_act_bn_fwd_bwd_vs_torch(rows, feat, act, post):
    """activation -> BatchNorm1d (train) -> activation (+ residual) of FCLayer, reference models/base_layers.py:100-111, through the
    fused kernels (the activations they carry: csrc/common.h) against the torch functions of the same names."""
    pre = rnd(rows, feat, seed=50) + 3.0            # large mean: exercises the shifted statistics
    gamma, beta = rnd(feat, seed=51) * 0.2 + 1, rnd(feat, seed=52) * 0.2
    res = rnd(rows, feat, seed=53)
    rm, rv = torch.zeros(feat), torch.ones(feat)
    acts = {'relu': F.relu, 'silu': F.silu, None: lambda t: t, 'leakyrelu': F.leaky_relu, 'sigmoid': torch.sigmoid}
    pr = pre.clone().requires_grad_(True)
    gr, br = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    x = acts[act](pr)
    y = acts[post](F.batch_norm(x, rm, rv, gr, br, True, 0.93, 1e-5)) + res
    cot = rnd(rows, feat, seed=54)
    (y * cot).sum().backward()
    rm_g, rv_g = torch.zeros(feat, device=DEV), torch.ones(feat, device=DEV)
    pre_g = g(pre)
    keep = act not in (None, 'relu', 'leakyrelu')
    xg, mean, invstd = ops.act_stats_fwd(pre_g.clone(), act, 1e-5, 0.93, rm_g, rv_g,
                                         out=torch.empty_like(pre_g) if keep else None)
    yg = ops.bn_apply_fwd(xg, mean, invstd, g(gamma), g(beta), post, g(res))
    assert rel_err(yg.cpu(), y.detach()) < 2e-5
    assert rel_err(rm_g.cpu(), rm) < 1e-5 and rel_err(rv_g.cpu(), rv) < 1e-5
    gp, gg, gb = ops.bn_bwd(g(cot), xg, pre_g if keep else None, act, post, mean, invstd, g(gamma), g(beta))
    assert rel_err(gp.cpu(), pr.grad) < 5e-5
    assert rel_err(gg.cpu(), gr.grad) < 5e-5 and rel_err(gb.cpu(), br.grad) < 5e-5
    # the fused variant: same grad_pre (bit for bit) plus its column sums (the bias gradient of the Linear in front) - against the
    # two-pass form of the plain call (the one-launch form takes its row sums in another order: test_bn_bwd_one_launch_...)
    gbias = torch.full((feat,), 7.0, device=DEV)
    was = ops.set_bn_bwd_one_launch(False)
    try:
        gp1, _, _ = ops.bn_bwd(g(cot), xg, pre_g if keep else None, act, post, mean, invstd, g(gamma), g(beta))
        gp2, _, _ = ops.bn_bwd(g(cot), xg, pre_g if keep else None, act, post, mean, invstd, g(gamma), g(beta), grad_bias=gbias)
    finally:
        ops.set_bn_bwd_one_launch(was)
    assert torch.equal(gp2, gp1) and rel_err(gp1.cpu(), gp.cpu()) < 2e-6
    ref_bias = gp.double().sum(0).cpu()
    assert float((gbias.cpu().double() - ref_bias).abs().max()) < 1e-5 * max(1.0, float(gp.abs().double().sum(0).max()))
    # eval mode
    with torch.no_grad():
        y_eval = acts[post](F.batch_norm(acts[act](pre), rm, rv, gamma, beta, False, 0.93, 1e-5)) + res
    x_eval = ops.act_fwd(pre_g, act) if act else pre_g
    ye = ops.bn_eval_fwd(x_eval, g(rm), g(rv), 1e-5, g(gamma), g(beta), post, g(res))
    assert rel_err(ye.cpu(), y_eval) < 2e-5


@pytest.mark.parametrize('rows,feat', [(16638, 200), (8409, 200), (512, 200), (20480, 200), (20481, 200), (33, 8), (40, 4), (5000, 256), (700, 1000),
                                       (3000, 64), (257, 92)])
@pytest.mark.parametrize('act,post', [(None, None), ('relu', None), ('leakyrelu', 'relu'), (None, 'silu')])
def test_bn_bwd_one_launch_against_the_two_pass_kernels(rows, feat, act, post):
    """The BatchNorm backward as ONE launch (csrc/bn.hip: bn_bwd_fused_kernel - row chunks kept in registers across an in-launch
    reduction, every workgroup waits for the finalised sums) against the two-pass kernels (reduction, then data gradient) and
    against an fp64 evaluation of the reference's autograd (models/base_layers.py:100-111): same expressions, the row sums in
    another fixed order - grad_pre / grad_gamma / grad_beta to rounding; bit-identical between repeated launches (the counters
    re-arm, shapes alternate on one workspace); grad_pre may alias grad_y; a strided output; the exact-zero bias gradient."""
    acts = {'relu': F.relu, None: lambda t: t, 'leakyrelu': F.leaky_relu, 'silu': F.silu}
    pre = rnd(rows, feat, seed=60) + 1.5
    x = acts[act](pre).contiguous()
    dy = rnd(rows, feat, seed=61)
    gamma, beta = rnd(feat, seed=62) * 0.2 + 1, rnd(feat, seed=63) * 0.2
    mean, var = x.mean(0), x.var(0, unbiased=False)
    invstd = 1 / torch.sqrt(var + 1e-5)
    # fp64 reference of autograd's expression with the statistics the kernels are given (fp32 mean / invstd) and the GATES of the
    # ReLU-class activations decided as the kernels decide them (from the fp32 values, same operation order): a gate within fp32
    # rounding of its kink is right either way, but it moves its whole column through the two sums
    def dact(name, z32):
        if name == 'relu':
            return (z32 > 0).double()
        if name == 'leakyrelu':
            return torch.where(z32 > 0, 1.0, 0.01).double()
        if name == 'silu':
            z = z32.double()
            sg = torch.sigmoid(z)
            return sg * (1 + z * (1 - sg))
        return torch.ones_like(z32, dtype=torch.float64)
    xh32 = (x - mean) * invstd
    xh = (x.double() - mean.double()) * invstd.double()
    d = dy.double() * dact(post, xh32 * gamma + beta)
    s1, s2 = d.sum(0), (d * xh).sum(0)
    ref_gp = gamma.double() * invstd.double() * (d - s1 / rows - xh * s2 / rows) * dact(act, x)
    xg, dyg, mg, ig, gg_, bg = g(x), g(dy), g(mean), g(invstd), g(gamma), g(beta)
    was = ops.set_bn_bwd_one_launch(False)
    try:
        gp2, gg2, gb2 = ops.bn_bwd(dyg, xg, None, act, post, mg, ig, gg_, bg)
        ops.set_bn_bwd_one_launch(True)
        gp1, gg1, gb1 = ops.bn_bwd(dyg, xg, None, act, post, mg, ig, gg_, bg)
        # interleave another shape on the same workspace, then repeat: the same bits
        other = ops.bn_bwd(g(rnd(777, feat, seed=64)), g(rnd(777, feat, seed=65)), None, None, None, mg, ig, gg_, bg)
        gp1b, gg1b, gb1b = ops.bn_bwd(dyg, xg, None, act, post, mg, ig, gg_, bg)
        assert torch.equal(gp1, gp1b) and torch.equal(gg1, gg1b) and torch.equal(gb1, gb1b)
        # in place (grad_pre aliases grad_y) and into a column block of a wider matrix, with the exact-zero bias gradient
        dy_io = dyg.clone()
        ops.bn_bwd(dy_io, xg, None, act, post, mg, ig, gg_, bg, out=dy_io)
        assert torch.equal(dy_io, gp1)
        if act is None and feat % 4 == 0:
            import ctypes
            L = importlib.import_module('3dinfomax_amd._lib')
            wide = torch.full((rows, feat + 12), 3.0, device=DEV)
            gbias = torch.full((feat,), 7.0, device=DEV)
            gg3, gb3 = torch.empty(feat, device=DEV), torch.empty(feat, device=DEV)
            p = lambda t: ctypes.c_void_p(t.data_ptr())      # noqa: E731
            ops.check(L.load().i3d_bn_bwd_strided(p(dyg), p(xg), None, rows, feat, L.ACT[act], L.ACT[post], p(mg), p(ig), p(gg_), p(bg), p(gg3), p(gb3),
                                                  ctypes.c_void_p(wide.data_ptr() + 4 * 8), feat + 12, p(gbias), p(ops._workspace(feat, xg.device)),
                                                  None, ops._stream()), 'i3d_bn_bwd_strided')
            assert torch.equal(wide[:, 8:8 + feat], gp1) and torch.all(wide[:, :8] == 3.0) and torch.all(wide[:, 8 + feat:] == 3.0)
            assert torch.all(gbias == 0) and torch.equal(gg3, gg1) and torch.equal(gb3, gb1)
    finally:
        ops.set_bn_bwd_one_launch(was)
    scale = float(ref_gp.abs().max())
    assert float((gp1.cpu().double() - ref_gp).abs().max()) < 3e-5 * scale
    assert float((gp1 - gp2).abs().max()) < 2e-6 * scale
    assert rel_err(gg1.cpu(), s2.float()) < 5e-5 and rel_err(gb1.cpu(), s1.float()) < 5e-5
    assert rel_err(gg1.cpu(), gg2.cpu()) < 2e-6 and rel_err(gb1.cpu(), gb2.cpu()) < 2e-6


def test_in_launch_finalisation_matches_separate_launch():
    """The default (stage 2 of the column reductions inside the stage-1 launch, last arrivers reduce) gives the same bits as
    separate finalise launches (I3D_FUSED_FINAL=0); the switch is read once per process -> subprocess."""
    import os
    import subprocess
    import sys
    code = (
        "import importlib, torch, sys\n"
        "sys.path.insert(0, %r)\n"
        "ops = importlib.import_module('3dinfomax_amd.ops')\n"
        "torch.manual_seed(0)\n"
        "out = []\n"
        "for rows, feat in ((70000, 20), (16000, 200), (300, 7), (5000, 2400)):\n"
        "    pre = torch.randn(rows, feat, device='cuda:0') + 2\n"
        "    rm, rv = torch.zeros(feat, device='cuda:0'), torch.ones(feat, device='cuda:0')\n"
        "    for rep in range(3):\n"
        "        x, mean, invstd = ops.act_stats_fwd(pre.clone(), 'relu', 1e-5, 0.9, rm, rv)\n"
        "        gb = torch.empty(feat, device='cuda:0')\n"
        "        gp, gg, gbeta = ops.bn_bwd(pre, x, None, 'relu', None, mean, invstd, torch.ones(feat, device='cuda:0'), torch.zeros(feat, device='cuda:0'), grad_bias=gb)\n"
        "        cs = ops.colsum(pre)\n"
        "    out += [mean, invstd, rm, rv, gg, gbeta, gb, cs, gp.sum(0)]\n"
        "torch.save([t.cpu() for t in out], sys.argv[1])\n") % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = {}
    for mode in ('0', '1'):
        path = f'/tmp/i3d_fused_final_{mode}.pt'
        env = dict(os.environ, I3D_FUSED_FINAL=mode)
        subprocess.run([sys.executable, '-c', code, path], check=True, env=env, timeout=300)
        res[mode] = torch.load(path)
    for a, b in zip(res['0'], res['1']):
        assert torch.equal(a, b)


def test_colsum_and_elementwise():
    x, w = rnd(3000, 200, seed=60), rnd(3000, seed=61)
    assert rel_err(ops.colsum(g(x)).cpu(), x.double().sum(0)) < 1e-5
    assert rel_err(ops.colsum(g(x), g(w)).cpu(), (x.double() * w.double()[:, None]).sum(0)) < 1e-5
    for act, fn in (('relu', F.relu), ('silu', F.silu), ('sigmoid', torch.sigmoid)):
        xr = x.clone().requires_grad_(True)
        fn(xr).backward(torch.ones_like(x))
        assert rel_err(ops.act_fwd(g(x), act).cpu(), fn(x)) < 1e-6
        assert rel_err(ops.act_bwd(g(torch.ones_like(x)), g(x), act).cpu(), xr.grad) < 1e-5
    a = g(x.clone())
    assert rel_err(ops.add_inplace(a, g(x)).cpu(), 2 * x) < 1e-7


# ---- edge kernels ----------------------------------------------------------------------------------------
@pytest.mark.parametrize('feat', [200, 20])
def test_edge_combine_and_segment_ops(feat):
    n = 200
    dst, ptr = _random_csr(n, 5, seed=70, zero_frac=0.05)
    E = dst.shape[0]
    gen = torch.Generator().manual_seed(7)
    src = torch.randint(0, n, (E,), generator=gen)
    P, Q, b = rnd(n, 2 * feat, seed=71), rnd(E, feat, seed=72), rnd(feat, seed=73)
    ref = P[src, :feat] + P[dst, feat:] + Q + b
    out = ops.edge_combine_fwd(g(P), g(Q), g(b), g(src.int()), g(dst.int()))
    assert rel_err(out.cpu(), ref) < 1e-6
    out = ops.edge_combine_fwd(g(P), None, None, g(src.int()), g(dst.int()))
    assert rel_err(out.cpu(), P[src, :feat] + P[dst, feat:]) < 1e-6
    # segment sums: by destination (contiguous) and by source (through an index list)
    x = rnd(E, feat, seed=74)
    ref_in = torch.zeros(n, feat).index_add(0, dst, x)
    assert rel_err(ops.segment_sum(g(x), g(ptr), None, n).cpu(), ref_in) < 1e-6
    deg = torch.bincount(dst, minlength=n).clamp(min=1).float()
    assert rel_err(ops.segment_sum(g(x), g(ptr), None, n, mean=True).cpu(), ref_in / deg[:, None]) < 1e-6
    order = torch.sort(src, stable=True)[1]
    optr = torch.zeros(n + 1, dtype=torch.int32)
    optr[1:] = torch.cumsum(torch.bincount(src, minlength=n), 0)
    ref_out = torch.zeros(n, feat).index_add(0, src, x)
    buf = torch.zeros(n, 2 * feat, device=DEV)
    ops.segment_sum(g(x), g(optr), g(order.int()), n, out=buf[:, feat:])          # strided output view
    assert rel_err(buf[:, feat:].cpu(), ref_out) < 1e-6
    gn = rnd(n, feat, seed=75)
    assert rel_err(ops.segment_bcast(g(gn), g(ptr), g(dst.int()), E, mean=True).cpu(), (gn / deg[:, None])[dst]) < 1e-6
    perm = torch.randperm(E, generator=gen).int()
    assert torch.equal(ops.gather_rows(g(x), g(perm)).cpu(), x[perm.long()])


@pytest.mark.parametrize('E,H', [(4000, 20), (4001, 32), (333, 8), (500, 13), (700, 40)])   # 8-lanes-per-edge / scalar kernels
def test_fourier_and_soft_edge(E, H):
    d = torch.rand(5000, generator=torch.Generator().manual_seed(1)) * 8 + 0.5
    ref = O.fourier_encode_dist(d[:, None], 4)
    assert rel_err(ops.fourier_encode(g(d), 4).cpu(), ref) < 1e-6
    m = rnd(E, H, seed=80).requires_grad_(True)
    ws, bs = rnd(1, H, seed=81).requires_grad_(True), rnd(1, seed=82).requires_grad_(True)
    w = torch.sigmoid(F.linear(m, ws, bs))
    msg = m * w
    cot = rnd(E, H, seed=83)
    (msg * cot).sum().backward()
    mg, wg = ops.soft_edge_fwd(g(m.detach()), g(ws.detach()), g(bs.detach()))
    assert rel_err(mg.cpu(), msg.detach()) < 1e-6
    gm, gg = ops.soft_edge_bwd(g(cot), g(m.detach()), wg, g(ws.detach()))
    assert rel_err(gm.cpu(), m.grad) < 1e-5
    assert rel_err(ops.colsum(g(m.detach()), gg).cpu(), ws.grad.view(-1)) < 1e-5
    assert rel_err(ops.colsum(gg.view(-1, 1)).cpu(), bs.grad) < 1e-5


# ---- NT-Xent ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize('B,dim,conf', [(8, 256, 1), (64, 64, 1), (512, 256, 1), (8, 256, 3), (500, 256, 3)])
def test_ntxent_fwd_bwd_vs_oracle(B, dim, conf):
    losses = importlib.import_module('3dinfomax_amd.losses')
    z1 = rnd(B, dim, seed=90).requires_grad_(True)
    z2 = rnd(B * conf, dim, seed=91).requires_grad_(True)
    ref = O.ntxent(z1, z2, 0.1) if conf == 1 else O.ntxent_multiple_positives(z1, z2, 0.1)
    ref.backward()
    a, b = g(z1.detach()).requires_grad_(True), g(z2.detach()).requires_grad_(True)
    loss_mod = losses.NTXent(tau=0.1) if conf == 1 else losses.NTXentMultiplePositives(tau=0.1)
    loss = loss_mod(a, b)
    loss.backward()
    assert abs(loss.item() - ref.item()) < 1e-5 * max(1.0, abs(ref.item()))
    assert rel_err(a.grad.cpu(), z1.grad) < 2e-5
    assert rel_err(b.grad.cpu(), z2.grad) < 2e-5


@pytest.mark.parametrize('B,dim,conf', [(64, 64, 1), (48, 32, 3)])
def test_ntxent_without_normalisation_vs_oracle(B, dim, conf):
    """norm=False (reference commons/losses.py:147, :236 skipped): raw dot products over tau; small embeddings so that
    exp(sim / tau) stays finite, as it has to in the reference too"""
    losses = importlib.import_module('3dinfomax_amd.losses')
    z1 = (rnd(B, dim, seed=94) * 0.15).requires_grad_(True)
    z2 = (rnd(B * conf, dim, seed=95) * 0.15).requires_grad_(True)
    ref = O.ntxent(z1, z2, 0.5, norm=False) if conf == 1 else O.ntxent_multiple_positives(z1, z2, 0.5, norm=False)
    ref.backward()
    a, b = g(z1.detach()).requires_grad_(True), g(z2.detach()).requires_grad_(True)
    loss_mod = losses.NTXent(norm=False, tau=0.5) if conf == 1 else losses.NTXentMultiplePositives(norm=False, tau=0.5)
    loss = loss_mod(a, b)
    loss.backward()
    assert abs(loss.item() - ref.item()) < 1e-5 * max(1.0, abs(ref.item()))
    assert rel_err(a.grad.cpu(), z1.grad) < 2e-5 and rel_err(b.grad.cpu(), z2.grad) < 2e-5


def test_ntxent_row_sharded_equals_full():
    """Data-parallel form on one device: shares over row shards with pos_offset sum to the full loss/grads."""
    losses = importlib.import_module('3dinfomax_amd.losses')
    B, dim, world = 64, 128, 4
    z1, z2 = g(rnd(B, dim, seed=92)), g(rnd(B, dim, seed=93))
    full1, full2 = z1.clone().requires_grad_(True), z2.clone().requires_grad_(True)
    full = losses.NTXentFn.apply(full1, full2, 0.1, 1e-8, 1, 0, B)
    full.backward()
    s1, s2 = z1.clone().requires_grad_(True), z2.clone().requires_grad_(True)
    per = B // world
    total = sum(losses.NTXentFn.apply(s1[r * per:(r + 1) * per], s2, 0.1, 1e-8, 1, r * per, B) for r in range(world))
    total.backward()
    assert abs(total.item() - full.item()) < 1e-6 * abs(full.item())
    assert rel_err(s1.grad.cpu(), full1.grad.cpu()) < 1e-5 and rel_err(s2.grad.cpu(), full2.grad.cpu()) < 1e-5


@pytest.mark.parametrize('cfg', list(range(13)))
def test_every_tile_configuration_on_awkward_shapes(cfg):
    """i3d_gemm_f32_ex with a forced tile configuration (workgroups of 1, 2 and 4 waves, 16- and 32-wide MFMA tiles, both
    LDS images) on shapes that are not multiples of anything, all four operand layouts, with bias and accumulate."""
    from ctypes import c_void_p
    L = importlib.import_module('3dinfomax_amd._lib')
    lib = L.load()
    st = c_void_p(torch.cuda.current_stream().cuda_stream)
    ws = torch.empty(16 << 20, dtype=torch.uint8, device=DEV)
    for (M, N, K) in ((131, 77, 53), (64, 200, 800), (1000, 33, 20), (37, 37, 1300)):
        for ta in (0, 1):
            for tb in (0, 1):
                A = rnd(K, M, seed=1) if ta else rnd(M, K, seed=1)
                B = rnd(N, K, seed=2) if tb else rnd(K, N, seed=2)
                bias = rnd(1, N, seed=3).reshape(N)
                C0 = rnd(M, N, seed=4)
                Ag, Bg, bg, C = g(A), g(B), g(bias), g(C0)
                splits = 3 if (ta and K > 1000) else 1
                rc = lib.i3d_gemm_f32_ex(ta, tb, M, N, K, c_void_p(Ag.data_ptr()), Ag.shape[1], c_void_p(Bg.data_ptr()), Bg.shape[1],
                                         c_void_p(C.data_ptr()), N, c_void_p(bg.data_ptr()), 1, cfg, splits,
                                         c_void_p(ws.data_ptr()), 16 << 20, st)
                assert rc == 0, (cfg, M, N, K, ta, tb, lib.i3d_last_error())
                ref = C0.double() + (A.double().T if ta else A.double()) @ (B.double().T if tb else B.double()) + bias.double()
                assert rel_err(C.cpu(), ref) < 2e-5, (cfg, M, N, K, ta, tb)


# ---- all weight gradients of a layer in one launch (csrc/wgrad.hip) ------------------------------------------
def _padded_groups(counts, n_rows, seed):
    """node ids grouped like graph.group_nodes_by_degree: every group padded to 64 with -1"""
    perm = torch.randperm(n_rows, generator=torch.Generator().manual_seed(seed))
    rows, starts, o = [], [], 0
    for c in counts:
        pad = (c + 63) // 64 * 64
        starts.append(len(rows))
        rows += perm[o:o + c].tolist() + [-1] * (pad - c)
        o += c
    return torch.tensor(rows, dtype=torch.int32), starts


@pytest.mark.parametrize('F,N,E', [(200, 3001, 6007), (40, 500, 1100)])
@pytest.mark.parametrize('mode', ['split_bf16', 'bf16'])
def test_wgrad_multi_on_the_bf16_pipe(F, N, E, mode, monkeypatch):
    """the one-launch weight gradients with split bf16 operands (a_h b_h + a_h b_l + a_l b_h, opt-in in fp32 mode: <= ~2^-15 per
    product, measured 2e-6 of the largest entry on these sums) and in the bf16 matmul mode (operands rounded once: against
    the fp64 product of the ROUNDED operands to 1e-4, like the other GEMMs of that mode, tests/test_gpu_bf16.py)"""
    dpost, h, agg = rnd(N, F, seed=1, scale=0.1), rnd(N, F, seed=2), rnd(N, 4 * F, seed=3)
    dpre2, x1 = rnd(E, F, seed=4, scale=0.1), rnd(E, F, seed=5)
    aff, row = rnd(3 * F, seed=10), rnd(F, seed=11)
    gW_h, gW2 = torch.empty(F, F), torch.empty(F, F)
    d = dict(dpost=g(dpost), h=g(h), dpre2=g(dpre2), x1=g(x1), aff=g(aff), row=g(row), gW_h=g(gW_h), gW2=g(gW2))
    problems = [dict(A=d['dpost'], B=d['h']), dict(A=d['dpre2'], B=d['x1'])]
    outputs = [dict(first_problem=0, C=d['gW_h']), dict(kind=ops.WGRAD_BN, first_problem=1, C=d['gW2'], aff=d['aff'], row=d['row'])]
    prev = ops.get_matmul_precision()
    try:
        if mode == 'bf16':
            ops.set_matmul_precision('bf16')
        else:
            monkeypatch.setenv('I3D_WGRAD_SPLIT_BF16', '1')
        ops.wgrad_multi(problems, outputs)
        torch.cuda.synchronize()
        first = [d['gW_h'].clone(), d['gW2'].clone()]
        ops.wgrad_multi(problems, outputs)
        torch.cuda.synchronize()
        assert torch.equal(first[0], d['gW_h']) and torch.equal(first[1], d['gW2'])       # fixed summation order
    finally:
        ops.set_matmul_precision(prev)
    R = (lambda t: t.bfloat16().double()) if mode == 'bf16' else (lambda t: t.double())
    tol = 1e-4 if mode == 'bf16' else 1e-5
    D = lambda t: t.double()
    assert rel_err(d['gW_h'].cpu(), R(dpost).T @ R(h)) < tol
    mean, scale, shift = D(aff[:F]), D(aff[F:2 * F]), D(aff[2 * F:])
    ref2 = (R(dpre2).T @ R(x1) - D(row)[:, None] * mean[None]) * scale[None] + D(row)[:, None] * shift[None]
    assert rel_err(d['gW2'].cpu(), ref2) < tol * (10 if mode == 'bf16' else 1)      # (bf16: `row` is not the column sum of the rounded dpre2)
    if mode == 'split_bf16':      # and it is NOT the plain bf16 product: that one is off by ~2e-3
        assert rel_err(d['gW_h'].cpu(), dpost.bfloat16().double().T @ h.bfloat16().double()) > 1e-4


@pytest.mark.parametrize('F,N,E', [(200, 3001, 6007), (40, 500, 1100), (16, 130, 70), (208, 2100, 900)])
def test_wgrad_multi_matches_fp64(F, N, E):
    """the five products of a PNA layer's backward (posttrans h-block, per-degree blocks folded into the scaler blocks,
    pretrans block against a never-materialised BatchNorm output, the two-block [W_s | W_d] gradient, the bond table's
    one-hot product) from ONE call, against fp64 torch; reference op: autograd of nn.Linear, models/base_layers.py:101"""
    A4, S = 4 * F, 3
    dpost, h, agg = rnd(N, F, seed=1, scale=0.1), rnd(N, F, seed=2), rnd(N, A4, seed=3)
    dpre2, x1 = rnd(E, F, seed=4, scale=0.1), rnd(E, F, seed=5)
    dP = rnd(N, 2 * F, seed=6, scale=0.1)
    dpre1 = rnd(E, F, seed=7, scale=0.1)
    V = 64
    onehot = torch.zeros(E, V)
    onehot[torch.arange(E), torch.randint(0, 60, (E,), generator=torch.Generator().manual_seed(8))] = 1.0
    counts = [N // 2, N // 7, N - N // 2 - N // 7 - 5, 5]
    deg_rows, starts = _padded_groups(counts, N, seed=9)
    coef = [1.0, 0.7, 1.4, 1.0, 1.1, 0.9, 1.0, 1.6, 0.6, 1.0, 1.8, 0.55]
    aff, row = rnd(3 * F, seed=10), rnd(F, seed=11)
    ldw = F + S * A4
    gW_post = torch.full((F, ldw), float('nan'))
    gW2 = torch.full((F, F), float('nan'))
    gW1 = torch.full((F, 3 * F), float('nan'))
    gQ = torch.full((V, F), float('nan'))
    d = dict(dpost=g(dpost), h=g(h), agg=g(agg), dpre2=g(dpre2), x1=g(x1), dP=g(dP), dpre1=g(dpre1), onehot=g(onehot),
             rows=g(deg_rows), aff=g(aff), row=g(row), gW_post=g(gW_post), gW2=g(gW2), gW1=g(gW1), gQ=g(gQ))
    problems = [dict(A=d['dpost'], B=d['h'])]
    for s0, c in zip(starts, counts):
        problems.append(dict(A=d['dpost'], B=d['agg'], rows=d['rows'], k_begin=s0, k_count=c))
    problems += [dict(A=d['dpre2'], B=d['x1']), dict(A=d['dP'], B=d['h']), dict(A=d['onehot'], B=d['dpre1'])]
    G = len(counts)
    outputs = [dict(first_problem=0, C=d['gW_post'], ldc=ldw),
               dict(kind=ops.WGRAD_COMBINE, first_problem=1, n_groups=G, C=d['gW_post'], c_offset=F, ldc=ldw, coef=coef,
                    n_scalers=S, scaler_stride=A4),
               dict(kind=ops.WGRAD_BN, first_problem=1 + G, C=d['gW2'], aff=d['aff'], row=d['row']),
               dict(first_problem=2 + G, C=d['gW1'], ldc=3 * F, c_split=F, c_delta=F - F * 3 * F),
               dict(first_problem=3 + G, C=d['gQ'])]
    ops.wgrad_multi(problems, outputs)
    torch.cuda.synchronize()
    D = lambda t: t.double()
    ref_h = D(dpost).T @ D(h)
    ref_blocks = torch.zeros(S, F, A4, dtype=torch.float64)
    for gi, (s0, c) in enumerate(zip(starts, counts)):
        idx = deg_rows[s0:s0 + c].long()
        wd = D(dpost[idx]).T @ D(agg[idx])
        for s in range(S):
            ref_blocks[s] += coef[gi * S + s] * wd
    got = d['gW_post'].cpu()
    assert rel_err(got[:, :F], ref_h) < 1e-5
    for s in range(S):
        assert rel_err(got[:, F + s * A4:F + (s + 1) * A4], ref_blocks[s]) < 1e-5
    mean, scale, shift = D(aff[:F]), D(aff[F:2 * F]), D(aff[2 * F:])
    ref2 = D(dpre2).T @ ((D(x1) - mean) * scale + shift)
    ref2_fix = (D(dpre2).T @ D(x1) - D(row)[:, None] * mean[None]) * scale[None] + D(row)[:, None] * shift[None]
    assert rel_err(d['gW2'].cpu(), ref2_fix) < 1e-5
    del ref2
    refP = D(dP).T @ D(h)                                  # [2F, F]: rows >= F land in the second column block of dW1
    g1 = d['gW1'].cpu()
    assert rel_err(g1[:, :F], refP[:F]) < 1e-5 and rel_err(g1[:, F:2 * F], refP[F:]) < 1e-5
    assert torch.isnan(g1[:, 2 * F:]).all()              # the W_q block is someone else's: untouched
    assert rel_err(d['gQ'].cpu(), D(onehot).T @ D(dpre1)) < 1e-5
    # bit-deterministic: the slices are summed in a fixed order
    first = [d[k].clone() for k in ('gW_post', 'gW2', 'gW1', 'gQ')]
    ops.wgrad_multi(problems, outputs)
    torch.cuda.synchronize()
    for a, k in zip(first, ('gW_post', 'gW2', 'gW1', 'gQ')):
        assert torch.equal(torch.nan_to_num(a), torch.nan_to_num(d[k]))


# ---- dropout (reference models/base_layers.py:84-85, 104-105; models/pna_original.py:260, 428) -------------------------------
@pytest.mark.parametrize('act,bn', [('relu', True), ('none', True), ('silu', True), ('relu', False)])
def test_fclayer_with_dropout_matches_torch_on_the_same_random_stream(act, bn):
    """FCLayer(dropout=p): Linear -> activation -> nn.Dropout -> BatchNorm1d, forward and backward against the same sequence of
    torch modules on the GPU with the same seed (the mask is torch's own: ops.dropout_mask draws it with torch's dropout kernel
    on a tensor of the same shape), and the eval-mode forward (no dropout)."""
    layers = importlib.import_module('3dinfomax_amd.layers')
    torch.manual_seed(0)
    fc = layers.FCLayer(24, 40, activation=act, dropout=0.3, batch_norm=bn, batch_norm_momentum=0.1).cuda().train()
    with torch.no_grad():
        fc.linear.weight.mul_(24 * 0.7)
        fc.linear.bias.normal_()
    x = torch.randn(300, 24, device='cuda:0')
    w = torch.randn(300, 40, device='cuda:0')
    ref_lin = torch.nn.Linear(24, 40).cuda()
    ref_lin.load_state_dict(fc.linear.state_dict())
    ref_bn = torch.nn.BatchNorm1d(40, momentum=0.1).cuda().train() if bn else None
    ref_act = {'relu': torch.relu, 'silu': torch.nn.functional.silu, 'none': lambda t: t}[act]

    def ref(xx, training):
        h = ref_act(ref_lin(xx))
        h = torch.nn.functional.dropout(h, 0.3, training)
        return ref_bn(h) if ref_bn is not None else h
    xa, xb = x.clone().requires_grad_(), x.clone().requires_grad_()
    torch.manual_seed(7)
    ya = fc(xa)
    torch.manual_seed(7)
    yb = ref(xb, True)
    assert rel_err(ya.cpu(), yb.detach().cpu()) < 2e-5
    (ya * w).sum().backward()
    (yb * w).sum().backward()
    assert rel_err(xa.grad.cpu(), xb.grad.cpu()) < 2e-4
    assert rel_err(fc.linear.weight.grad.cpu(), ref_lin.weight.grad.cpu()) < 2e-4
    if bn:
        assert rel_err(fc.batch_norm.weight.grad.cpu(), ref_bn.weight.grad.cpu()) < 2e-4
        assert rel_err(fc.batch_norm.running_var.cpu(), ref_bn.running_var.cpu()) < 1e-5
    fc.eval()
    if ref_bn is not None:
        ref_bn.eval()
    with torch.no_grad():
        assert rel_err(fc(x).cpu(), ref(x, False).cpu()) < 2e-5


def test_dropout_function_is_torch_dropout():
    layers = importlib.import_module('3dinfomax_amd.layers')
    x = torch.randn(513, 70, device='cuda:0').requires_grad_()
    torch.manual_seed(3)
    y = layers.dropout(x, 0.3, True)
    torch.manual_seed(3)
    y_ref = torch.nn.functional.dropout(x.detach(), 0.3, True)
    assert torch.equal(y.detach(), y_ref)
    y.sum().backward()
    assert torch.equal(x.grad, (y_ref != 0).float() / 0.7) or rel_err(x.grad.cpu(), ((y_ref != 0).float() / 0.7).cpu()) < 1e-6
    assert layers.dropout(x, 0.3, False) is x and layers.dropout(x, 0.0, True) is x


@pytest.mark.parametrize('bn', [True, False])
@pytest.mark.parametrize('act', ['Tanh', 'ELU', 'SELU', 'Softplus', 'LeakyReLU', 'Sigmoid'])
def test_fclayer_with_every_elementwise_activation_of_the_reference_map(act, bn):
    """FCLayer(activation=<name>) for the entries of the reference's SUPPORTED_ACTIVATION_MAP (models/base_layers.py:5, resolved by
    get_activation :9-20 to torch.nn.modules.activation.<name>() with default parameters) against Linear -> that torch module ->
    BatchNorm1d, forward, backward and eval mode.  (GLU halves the feature dimension: not an elementwise activation, not offered.)"""
    layers = importlib.import_module('3dinfomax_amd.layers')
    torch.manual_seed(1)
    fc = layers.FCLayer(24, 40, activation=act, batch_norm=bn, batch_norm_momentum=0.1).cuda().train()
    with torch.no_grad():
        fc.linear.weight.mul_(24 * 0.3)
        fc.linear.bias.normal_()
    ref_lin = torch.nn.Linear(24, 40).cuda()
    ref_lin.load_state_dict(fc.linear.state_dict())
    ref_bn = torch.nn.BatchNorm1d(40, momentum=0.1).cuda().train() if bn else None
    ref_act = vars(torch.nn.modules.activation)[act]()
    x = torch.randn(300, 24, device='cuda:0')
    w = torch.randn(300, 40, device='cuda:0')

    def ref(xx):
        h = ref_act(ref_lin(xx))
        return ref_bn(h) if ref_bn is not None else h
    xa, xb = x.clone().requires_grad_(), x.clone().requires_grad_()
    ya, yb = fc(xa), ref(xb)
    assert rel_err(ya.cpu(), yb.detach().cpu()) < 2e-5
    (ya * w).sum().backward()
    (yb * w).sum().backward()
    assert rel_err(xa.grad.cpu(), xb.grad.cpu()) < 2e-4
    assert rel_err(fc.linear.weight.grad.cpu(), ref_lin.weight.grad.cpu()) < 2e-4
    if bn:
        assert rel_err(fc.batch_norm.weight.grad.cpu(), ref_bn.weight.grad.cpu()) < 2e-4
        assert rel_err(fc.batch_norm.bias.grad.cpu(), ref_bn.bias.grad.cpu()) < 2e-4
        assert rel_err(fc.batch_norm.running_var.cpu(), ref_bn.running_var.cpu()) < 1e-5
    else:
        assert rel_err(fc.linear.bias.grad.cpu(), ref_lin.bias.grad.cpu()) < 2e-4
    fc.eval()
    if ref_bn is not None:
        ref_bn.eval()
    with torch.no_grad():
        assert rel_err(fc(x).cpu(), ref(x).cpu()) < 2e-5


@pytest.mark.parametrize('bn', [True, False])
@pytest.mark.parametrize('act', ['relu', 'SiLU', 'none'])
def test_fclayer_without_bias(act, bn):
    """FCLayer(bias=False), reference models/base_layers.py:86 (nn.Linear(in_dim, out_dim, bias=False)): no bias parameter, the
    reference's state_dict keys, Linear -> activation -> BatchNorm1d forward / backward / eval against torch."""
    layers = importlib.import_module('3dinfomax_amd.layers')
    torch.manual_seed(2)
    fc = layers.FCLayer(24, 40, activation=act, batch_norm=bn, batch_norm_momentum=0.1, bias=False).cuda().train()
    assert fc.linear.bias is None
    assert set(fc.state_dict()) == {'linear.weight'} | ({'batch_norm.weight', 'batch_norm.bias', 'batch_norm.running_mean',
                                                         'batch_norm.running_var', 'batch_norm.num_batches_tracked'} if bn else set())
    with torch.no_grad():
        fc.linear.weight.mul_(24 * 0.3)
    ref_lin = torch.nn.Linear(24, 40, bias=False).cuda()
    ref_lin.load_state_dict(fc.linear.state_dict())
    ref_bn = torch.nn.BatchNorm1d(40, momentum=0.1).cuda().train() if bn else None
    ref_act = {'relu': torch.nn.ReLU(), 'SiLU': torch.nn.SiLU(), 'none': torch.nn.Identity()}[act]
    x = torch.randn(300, 24, device='cuda:0')
    w = torch.randn(300, 40, device='cuda:0')

    def ref(xx):
        h = ref_act(ref_lin(xx))
        return ref_bn(h) if ref_bn is not None else h
    xa, xb = x.clone().requires_grad_(), x.clone().requires_grad_()
    ya, yb = fc(xa), ref(xb)
    assert rel_err(ya.cpu(), yb.detach().cpu()) < 2e-5
    (ya * w).sum().backward()
    (yb * w).sum().backward()
    assert rel_err(xa.grad.cpu(), xb.grad.cpu()) < 2e-4
    assert rel_err(fc.linear.weight.grad.cpu(), ref_lin.weight.grad.cpu()) < 2e-4
    if bn:
        assert rel_err(fc.batch_norm.weight.grad.cpu(), ref_bn.weight.grad.cpu()) < 2e-4
        assert rel_err(fc.batch_norm.running_var.cpu(), ref_bn.running_var.cpu()) < 1e-5
    assert [n for n, _ in fc.named_parameters()] == ['linear.weight'] + (['batch_norm.weight', 'batch_norm.bias'] if bn else [])
    fc.eval()
    if ref_bn is not None:
        ref_bn.eval()
    with torch.no_grad():
        assert rel_err(fc(x).cpu(), ref(x).cpu()) < 2e-5


def test_glu_is_refused_by_name():
    layers = importlib.import_module('3dinfomax_amd.layers')
    with pytest.raises(NotImplementedError):
        layers.FCLayer(8, 8, activation='GLU')


@pytest.mark.parametrize('training', [True, False])
@pytest.mark.parametrize('bn', [True, False])
@pytest.mark.parametrize('act,post', [('tanh', 'elu'), ('selu', 'softplus'), ('softplus', 'tanh'), ('elu', None), (None, 'selu'),
                                      ('relu', 'tanh'), ('tanh', 'silu')])
def test_tail_with_elementwise_only_activations_vs_torch(act, post, bn, training):
    """The activations the fused kernels do not carry (Tanh, ELU, SELU, Softplus: layers.ELEMENTWISE_ONLY) in front of and behind
    the BatchNorm: layers._Tail runs them as passes of their own around the fused BatchNorm kernels - forward, backward, running
    statistics and the residual against the torch functions; the fused entry points refuse their codes."""
    layers = importlib.import_module('3dinfomax_amd.layers')
    rows, feat = 700, 36
    acts = {'relu': F.relu, 'silu': F.silu, None: lambda t: t, 'tanh': torch.tanh, 'elu': F.elu, 'selu': F.selu, 'softplus': F.softplus}
    pre = rnd(rows, feat, seed=60) + 0.3
    gamma, beta, res, cot = rnd(feat, seed=61) * 0.2 + 1, rnd(feat, seed=62) * 0.2, rnd(rows, feat, seed=63), rnd(rows, feat, seed=64)
    rm, rv = rnd(feat, seed=65) * 0.1, rnd(feat, seed=66).abs() * 0.5 + 0.5
    pr, gr, br = pre.clone().requires_grad_(True), gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    rm_ref, rv_ref = rm.clone(), rv.clone()
    h = acts[act](pr)
    if bn:
        h = F.batch_norm(h, rm_ref, rv_ref, gr, br, training, 0.2, 1e-5)
    y = acts[post](h) + res
    (y * cot).sum().backward()
    spec_bn = layers.BNSpec(g(rm), g(rv), None, 0.2, 1e-5, training) if bn else None
    spec = layers.FCSpec(act, spec_bn, post)
    pre_g = g(pre)
    yg, saved = layers._Tail.forward(pre_g, g(gamma) if bn else None, g(beta) if bn else None, spec, g(res))
    assert rel_err(yg.cpu(), y.detach()) < 2e-5
    assert torch.equal(pre_g.cpu(), pre)                      # the pre-activation is kept (its derivative needs it)
    gp, gg, gb = layers._Tail.backward(saved, g(cot), g(gamma) if bn else None, g(beta) if bn else None, spec)
    assert rel_err(gp.cpu(), pr.grad) < 5e-5
    if bn:
        assert rel_err(gg.cpu(), gr.grad) < 5e-5 and rel_err(gb.cpu(), br.grad) < 5e-5
        assert rel_err(spec_bn.running_mean.cpu(), rm_ref) < 1e-5 and rel_err(spec_bn.running_var.cpu(), rv_ref) < 1e-5
    ops_ = importlib.import_module('3dinfomax_amd.ops')
    lib = importlib.import_module('3dinfomax_amd._lib')
    with pytest.raises(lib.HipLibraryError):
        ops_.act_stats_fwd(pre_g.clone(), 'tanh', 1e-5, 0.1, torch.zeros(feat, device=DEV), torch.ones(feat, device=DEV))


@pytest.mark.parametrize('feat,act', [(200, 'relu'), (20, 'relu'), (64, None), (36, 'leakyrelu')])
def test_edge_block_bn_backward_fused_with_its_segmented_sums(feat, act):
    """i3d_bn_bwd_edge_sums (the BatchNorm backward of a PNA layer's edge block with the data gradient formed inside the two segmented
    sums that consume it) against the launches it replaces - i3d_bn_bwd, then the sums over the in-edges (contiguous rows) and over
    the out-edges (through the index list) - bit for bit: grad_pre, dP[src], dP[dst] (written into column blocks of a wider matrix),
    grad_gamma / grad_beta; and the bias gradient as the column sum of dP[dst] (i3d_colsum_strided) against the fp64 sum."""
    import ctypes
    L = importlib.import_module('3dinfomax_amd._lib')
    lib = L.load()
    n = 900
    dst, ptr = _random_csr(n, 5, seed=170, zero_frac=0.05)
    E = dst.shape[0]
    gen = torch.Generator().manual_seed(17)
    src = torch.randint(0, n, (E,), generator=gen)
    order = torch.sort(src, stable=True)[1].int()
    optr = torch.zeros(n + 1, dtype=torch.int32)
    optr[1:] = torch.cumsum(torch.bincount(src, minlength=n), 0)
    pre = rnd(E, feat, seed=171) + 0.5
    acts = {'relu': F.relu, None: lambda t: t, 'leakyrelu': F.leaky_relu}
    x = acts[act](pre).contiguous()
    dy = rnd(E, feat, seed=172)
    gamma, beta = rnd(feat, seed=173) * 0.2 + 1, rnd(feat, seed=174) * 0.2
    mean, var = x.mean(0), x.var(0, unbiased=False)
    invstd = 1 / torch.sqrt(var + 1e-5)
    xg, dyg, mg, ig, gg_, bg = g(x), g(dy), g(mean), g(invstd), g(gamma), g(beta)
    # the launches it replaces (two-pass form: the reduction whose sums the fused entry takes as well)
    was = ops.set_bn_bwd_one_launch(False)
    try:
        gp_ref, ggam_ref, gbet_ref = ops.bn_bwd(dyg, xg, None, act, None, mg, ig, gg_, bg)
    finally:
        ops.set_bn_bwd_one_launch(was)
    W = 2 * feat + 8
    DL_ref = torch.zeros(n, W, device=DEV)
    ops.segment_sum(gp_ref, g(optr), g(order), n, out=DL_ref[:, :feat])
    ops.segment_sum(gp_ref, g(ptr), None, n, out=DL_ref[:, feat:2 * feat])
    # the fused entry
    gp = torch.empty_like(xg)
    ggam, gbet = torch.empty(feat, device=DEV), torch.empty(feat, device=DEV)
    DL = torch.zeros(n, W, device=DEV)
    ws = ops._workspace(feat, xg.device)
    st = torch.cuda.current_stream().cuda_stream
    p = lambda t: ctypes.c_void_p(t.data_ptr())      # noqa: E731
    ptr_g, optr_g, order_g = g(ptr), g(optr), g(order)          # (kept alive: only their addresses go through the C ABI)
    ops.check(lib.i3d_bn_bwd_edge_sums(p(dyg), p(xg), E, feat, L.ACT[act], p(mg), p(ig), p(gg_), p(bg), p(ggam), p(gbet), p(gp), p(ptr_g),
                                       p(optr_g), p(order_g), n, p(DL), ctypes.c_void_p(DL.data_ptr() + 4 * feat), W, p(ws), st),
              'i3d_bn_bwd_edge_sums')
    assert torch.equal(gp, gp_ref) and torch.equal(ggam, ggam_ref) and torch.equal(gbet, gbet_ref)
    assert torch.equal(DL, DL_ref)
    bias = torch.empty(feat, device=DEV)
    part = torch.empty(lib.i3d_bn_bias_partial_floats(feat), device=DEV)
    ops.check(lib.i3d_colsum_strided(ctypes.c_void_p(DL.data_ptr() + 4 * feat), W, n, feat, p(bias), p(part), st), 'i3d_colsum_strided')
    ref_bias = gp_ref.double().sum(0).cpu()
    assert float((bias.cpu().double() - ref_bias).abs().max()) < 2e-6 * max(1.0, float(gp_ref.abs().double().sum(0).max()))


# ---- the tower variant's block-diagonal posttrans products and tower-major aggregation (round 5) ---------------------------
@pytest.mark.parametrize('rows,T,Fo,Kt', [(8500, 5, 20, 1104), (300, 5, 20, 240), (1000, 3, 8, 48), (64, 2, 4, 16)])
def test_batched_gemm_is_the_block_diagonal_product(rows, T, Fo, Kt):
    """i3d_gemm_f32_batched (csrc/gemm.hip): the T diagonal blocks of [rows, T Kt] x [T Fo, T Kt]^T in one launch - forward layout,
    data-gradient layout and weight-gradient layout (K-slices through the scratch) - against fp64 products of the blocks."""
    import ctypes
    _lib = importlib.import_module('3dinfomax_amd._lib')
    L = _lib.load()
    st = ops._stream()
    D = 12                                     # the blocks sit behind D columns of a wider weight, as in csrc/tower.hip
    ldq = D + T * Kt
    A = g(rnd(rows, T * Kt, seed=1))
    W = g(rnd(T * Fo, ldq, seed=2))
    G = g(rnd(rows, T * Fo, seed=3))
    ws = ops._gemm_workspace(DEV)
    Ad, Wd, Gd = A.double().cpu(), W.double().cpu(), G.double().cpu()
    # forward: C[:, t Fo ..] += A[:, t Kt ..] W[t Fo .., D + t Kt ..]^T
    C0 = g(rnd(rows, T * Fo, seed=4))
    C = C0.clone()
    _lib.check(L.i3d_gemm_f32_batched(0, 1, rows, Fo, Kt, A.data_ptr(), T * Kt, Kt, W.data_ptr() + 4 * D, ldq, Fo * ldq + Kt, C.data_ptr(),
                                      T * Fo, Fo, T, 1, None, 0, st), 'i3d_gemm_f32_batched')
    ref = C0.double().cpu().clone()
    for t in range(T):
        ref[:, t * Fo:(t + 1) * Fo] += Ad[:, t * Kt:(t + 1) * Kt] @ Wd[t * Fo:(t + 1) * Fo, D + t * Kt:D + (t + 1) * Kt].T
    assert rel_err(C.cpu(), ref) < 2e-5
    # data gradient: dA[:, t Kt ..] = G[:, t Fo ..] W[t Fo .., D + t Kt ..]
    dA = torch.full_like(A, float('nan'))
    _lib.check(L.i3d_gemm_f32_batched(0, 0, rows, Kt, Fo, G.data_ptr(), T * Fo, Fo, W.data_ptr() + 4 * D, ldq, Fo * ldq + Kt, dA.data_ptr(),
                                      T * Kt, Kt, T, 0, None, 0, st), 'i3d_gemm_f32_batched')
    ref = torch.cat([Gd[:, t * Fo:(t + 1) * Fo] @ Wd[t * Fo:(t + 1) * Fo, D + t * Kt:D + (t + 1) * Kt] for t in range(T)], dim=1)
    assert rel_err(dA.cpu(), ref) < 2e-5
    # weight gradient: dW[t Fo .., D + t Kt ..] = G[:, t Fo ..]^T A[:, t Kt ..]; nothing else of dW is written
    dW = torch.full_like(W, 7.0)
    _lib.check(L.i3d_gemm_f32_batched(1, 0, Fo, Kt, rows, G.data_ptr(), T * Fo, Fo, A.data_ptr(), T * Kt, Kt, dW.data_ptr() + 4 * D, ldq,
                                      Fo * ldq + Kt, T, 0, ws.data_ptr(), ops.GEMM_WORKSPACE_BYTES, st), 'i3d_gemm_f32_batched')
    ref = torch.full((T * Fo, ldq), 7.0, dtype=torch.float64)
    for t in range(T):
        ref[t * Fo:(t + 1) * Fo, D + t * Kt:D + (t + 1) * Kt] = Gd[:, t * Fo:(t + 1) * Fo].T @ Ad[:, t * Kt:(t + 1) * Kt]
    assert rel_err(dW.cpu(), ref) < 2e-5


@pytest.mark.parametrize('T,Ft,std', [(5, 20, True), (5, 92, True), (3, 8, False)])
def test_tower_major_aggregation_is_a_column_permutation(T, Ft, std):
    """i3d_pna_aggregate_fwd_towers / _bwd_towers: the node rows as [tower][block][feature] - the same values as the
    [block][tower][feature] rows of i3d_pna_aggregate_fwd, bit for bit, and the same message gradient from the permuted gradient."""
    synth = importlib.import_module('3dinfomax_amd.synth')
    amd = importlib.import_module('3dinfomax_amd')
    mols = synth.make_dataset(24, seed=3)
    idx = amd.batch([amd.bond_graph(m) for m in mols]).to(DEV).index()
    E, N, F_ = idx.num_edges, idx.num_nodes, T * Ft
    aggs = ops.agg_codes(['mean', 'max', 'min', 'std'] if std else ['sum', 'max', 'var'])
    scal = ops.scaler_codes(['identity', 'amplification', 'attenuation'] if std else ['identity', 'attenuation'])
    B = len(aggs) * len(scal)
    e = g(rnd(E, F_, seed=5))
    ref = ops.pna_aggregate_fwd(e, idx.in_ptr, N, aggs, scal, 1.3, True)
    out = ops.pna_aggregate_fwd(e, idx.in_ptr, N, aggs, scal, 1.3, True, tower_feat=Ft)
    perm = ref.view(N, B, T, Ft).permute(0, 2, 1, 3).reshape(N, B * F_)
    assert torch.equal(out, perm)
    go = g(rnd(N, B * F_, seed=6))
    go_tm = go.view(N, B, T, Ft).permute(0, 2, 1, 3).reshape(N, B * F_).contiguous()
    ge_ref = ops.pna_aggregate_bwd(go, e, idx.in_ptr, N, aggs, scal, 1.3, True)
    ge = ops.pna_aggregate_bwd(go_tm, e, idx.in_ptr, N, aggs, scal, 1.3, True, tower_feat=Ft)
    assert torch.equal(ge, ge_ref)
