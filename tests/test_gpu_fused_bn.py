"""-m gpu: the fused-BatchNorm building blocks (csrc/fused_bn.hip, the FUSE variants of csrc/gemm.hip, the aff loads of
csrc/aggregate.hip) against fp64 torch on the CPU.  The reference op is FCLayer: Linear -> activation -> BatchNorm1d in
training mode (reference models/base_layers.py:100-111); tolerances next to each check (bar: 1e-4, kernels held to ~1e-5)."""
import importlib

import numpy as np
import pytest
import torch

from helpers import rel_err

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


def act_ref(x, act):
    return {None: lambda v: v, 'relu': torch.relu, 'leakyrelu': lambda v: torch.nn.functional.leaky_relu(v, 0.01)}[act](x)


def bn_stats_ref(x):
    x = x.double()
    mean, var = x.mean(0), x.var(0, unbiased=False)
    return mean, 1.0 / torch.sqrt(var + 1e-5), x.var(0, unbiased=True)


def check_stats(partial, tiles, x_ref, gamma, beta, momentum=0.93, col_shift=None):
    feat = x_ref.shape[1]
    rm, rv = rnd(feat, seed=90), rnd(feat, seed=91).abs() + 0.5
    rm_d, rv_d = g(rm.clone()), g(rv.clone())
    nbt = torch.tensor(4, dtype=torch.int64, device=DEV)
    mean, invstd, aff = ops.bn_finalize_partials(partial, tiles, feat, 1e-5, momentum, g(gamma), g(beta), rm_d, rv_d, nbt)
    m_ref, is_ref, unb = bn_stats_ref(x_ref)
    # statistics are centred per tile and merged in fp64: relative to the column's own scale even when |mean| >> std
    assert ((mean.cpu().double() - m_ref).abs() / (1 / is_ref + m_ref.abs() * 1e-6)).max().item() < 2e-6
    assert rel_err(invstd.cpu(), is_ref) < 1e-5
    assert rel_err(rm_d.cpu(), (1 - momentum) * rm.double() + momentum * m_ref) < 1e-5
    assert rel_err(rv_d.cpu(), (1 - momentum) * rv.double() + momentum * unb) < 1e-5
    assert int(nbt.item()) == 5
    assert torch.equal(aff[0], mean) and torch.equal(aff[2].cpu(), beta)
    assert rel_err(aff[1].cpu(), gamma.double() * is_ref) < 1e-5
    return mean, invstd, aff


@pytest.mark.parametrize('E,N,F,act,table', [(16907, 8511, 200, 'relu', True), (333, 90, 200, None, False),
                                             (2000, 500, 20, 'leakyrelu', True), (41, 7, 36, 'relu', False),
                                             (5000, 900, 7, 'relu', False)])
def test_edge_combine_act_stats(E, N, F, act, table):
    rng = np.random.default_rng(E)
    P = rnd(N, 2 * F, seed=1)
    P[:, :F] += 3.0                                       # columns with |mean| >> std after the activation
    src = torch.from_numpy(rng.integers(0, N, E).astype(np.int32))
    dst = torch.from_numpy(np.sort(rng.integers(0, N, E)).astype(np.int32))
    V = 60
    Q = rnd(V if table else E, F, seed=2)
    code = torch.from_numpy(rng.integers(0, V, E).astype(np.int32)) if table else None
    bias = rnd(F, seed=3)
    x, partial, tiles = ops.edge_combine_act_stats(g(P), g(Q), g(bias), g(src), g(dst), act, g(code) if table else None)
    qrow = code.long() if table else torch.arange(E)
    ref = act_ref(P[src.long(), :F] + P[dst.long(), F:] + Q[qrow] + bias, act)
    assert torch.equal(x.cpu(), ref)                      # same fp32 additions in the same order as edge_combine_fwd
    assert float(partial[:, 2].sum().item()) == E * F
    check_stats(partial, tiles, ref, 1 + 0.2 * rnd(F, seed=4), 0.2 * rnd(F, seed=5))


@pytest.mark.parametrize('M,N,K', [(16907, 200, 200), (8511, 200, 800), (130, 200, 200), (1000, 36, 64), (64, 256, 1024)])
@pytest.mark.parametrize('act', [None, 'relu'])
@pytest.mark.parametrize('prologue', [True, False])
def test_gemm_fused_prologue_and_statistics(M, N, K, act, prologue):
    A, W, bias = rnd(M, K, seed=1) + 2.0, rnd(N, K, seed=2, scale=K ** -0.5), rnd(N, seed=3)
    aff = None
    A_eff = A.double()
    if prologue:
        aff = torch.stack([rnd(K, seed=4) + 2.0, 1 + 0.3 * rnd(K, seed=5), 0.3 * rnd(K, seed=6)])
        A_eff = (A.double() - aff[0].double()) * aff[1].double() + aff[2].double()
    ref = act_ref(A_eff @ W.double().T + bias.double(), act)
    out, partial, tiles = ops.gemm_fused(g(A), g(W), g(bias), g(aff) if prologue else None, act)
    assert tiles == (M + 63) // 64 and rel_err(out.cpu(), ref) < 1e-5
    # the statistics describe the values that were STORED (fp32), to fp32 accuracy of their own
    check_stats(partial, tiles, out.cpu(), 1 + 0.2 * rnd(N, seed=7), 0.2 * rnd(N, seed=8))
    if prologue:       # without the statistics epilogue
        out2, p2, _ = ops.gemm_fused(g(A), g(W), g(bias), g(aff), None, want_stats=False)
        assert p2 is None and rel_err(out2.cpu(), A_eff @ W.double().T + bias.double()) < 1e-5


def test_gemm_fused_grouped_accumulate_statistics():
    """posttrans: lin = h W_h^T + b (plain GEMM), += agg[deg group] W_D^T with the statistics of the SUM in the epilogue;
    groups padded to 64 rows with -1, an in-degree-0 group with zero weights included"""
    graph = importlib.import_module('3dinfomax_amd.graph')
    n, Fo, A = 1500, 200, 800
    rng = np.random.default_rng(0)
    indeg = rng.choice([0, 1, 2, 3, 4, 6], size=n, p=[0.03, 0.45, 0.1, 0.1, 0.3, 0.02])
    rows, tiles_g, groups = graph.group_nodes_by_degree(indeg, include_zero=True)
    assert groups[0][0] == 0 and sorted(rows[rows >= 0].tolist()) == list(range(n))
    agg, lin0 = rnd(n, A, seed=1), rnd(n, Fo, seed=2)
    WD = rnd(len(groups), Fo, A, seed=3, scale=A ** -0.5)
    WD[0] = 0
    lin = g(lin0.clone())
    _, partial, tiles = ops.gemm_fused(g(agg), g(WD), None, None, None, out=lin, accumulate=True,
                                       m_rows=g(torch.from_numpy(rows)), tile_group=g(torch.from_numpy(tiles_g)))
    ref = lin0.double().clone()
    for gi, (D, start, count) in enumerate(groups):
        r = torch.from_numpy(rows[start:start + count]).long()
        ref[r] += agg[r].double() @ WD[gi].double().T
    assert rel_err(lin.cpu(), ref) < 1e-5
    assert float(partial[:, 2, 0].sum().item()) == n           # padding rows are not counted
    check_stats(partial, tiles, lin.cpu(), 1 + 0.2 * rnd(Fo, seed=7), 0.2 * rnd(Fo, seed=8))


@pytest.mark.parametrize('rows,Fo,Fi', [(16907, 200, 200), (700, 36, 64)])
def test_wgrad_against_unmaterialised_batchnorm_output(rows, Fo, Fi):
    dY, x = rnd(rows, Fo, seed=1, scale=0.1), rnd(rows, Fi, seed=2).abs()
    aff = torch.stack([x.mean(0), 1 + 0.3 * rnd(Fi, seed=5), 0.3 * rnd(Fi, seed=6)])
    y = (x.double() - aff[0].double()) * aff[1].double() + aff[2].double()
    ref = dY.double().T @ y
    gb = dY.sum(0)
    dW = ops.gemm_wgrad_bn(g(dY), g(x), g(gb), g(aff))
    assert rel_err(dW.cpu(), ref) < 2e-5


@pytest.mark.parametrize('F,scalers', [(200, ['identity']), (200, ['identity', 'amplification', 'attenuation']), (36, ['identity'])])
def test_aggregation_with_batchnorm_applied_on_load(F, scalers):
    n = 900
    rng = np.random.default_rng(3)
    deg = rng.integers(0, 7, size=n)
    ptr = np.zeros(n + 1, dtype=np.int32)
    ptr[1:] = np.cumsum(deg)
    E = int(ptr[-1])
    e = rnd(E, F, seed=1) * 3 + 1
    aff = torch.stack([rnd(F, seed=4) + 1.0, 1 + 0.3 * rnd(F, seed=5), 0.3 * rnd(F, seed=6)])
    aff[1, ::7] *= -1                                   # negative scale: max and min swap roles
    m = ((e - aff[0]) * aff[1] + aff[2]).contiguous()   # the kernel's arithmetic: sub, mul, add in fp32
    aggs = ops.agg_codes(['mean', 'max', 'min', 'std'])
    sc = ops.scaler_codes(scalers)
    ptr_d = g(torch.from_numpy(ptr))
    out = ops.pna_aggregate_fwd_aff(g(e), g(aff), ptr_d, n, aggs, sc, 1.0)
    ref = ops.pna_aggregate_fwd_aff(g(m), None, ptr_d, n, aggs, sc, 1.0)
    assert torch.equal(out, ref)
    gout = g(rnd(*out.shape, seed=9))
    ge = ops.pna_aggregate_bwd_aff(gout, g(e), g(aff), ptr_d, n, aggs, sc, 1.0)
    ge_ref = ops.pna_aggregate_bwd_aff(gout, g(m), None, ptr_d, n, aggs, sc, 1.0)
    assert torch.equal(ge, ge_ref)                      # gradient w.r.t. the normalised message, as documented


@pytest.mark.parametrize('scalers', [['identity'], ['identity', 'amplification', 'attenuation']])
def test_aggregation_reads_messages_stored_as_bf16(scalers):
    """the bf16 matmul mode's storage form of the messages (i3d_pna_aggregate_fwd_ex / _bwd_ex, e_bf16 = 1): the same bits as the
    fp32 kernels on the same (bf16-representable) values, with and without the BatchNorm applied on load"""
    n, F = 700, 200
    rng = np.random.default_rng(5)
    deg = rng.integers(0, 7, size=n)
    ptr = np.zeros(n + 1, dtype=np.int32)
    ptr[1:] = np.cumsum(deg)
    E = int(ptr[-1])
    e16 = (rnd(E, F, seed=2) * 3 + 1).bfloat16()
    aff = torch.stack([rnd(F, seed=4) + 1.0, 1 + 0.3 * rnd(F, seed=5), 0.3 * rnd(F, seed=6)])
    aggs = ops.agg_codes(['mean', 'max', 'min', 'std'])
    sc = ops.scaler_codes(scalers)
    ptr_d = g(torch.from_numpy(ptr))
    for a in (None, g(aff)):
        out = ops.pna_aggregate_fwd_aff(e16.cuda(), a, ptr_d, n, aggs, sc, 1.0)
        ref = ops.pna_aggregate_fwd_aff(g(e16.float()), a, ptr_d, n, aggs, sc, 1.0)
        assert torch.equal(out, ref)
        gout = g(rnd(*out.shape, seed=9))
        ge = ops.pna_aggregate_bwd_aff(gout, e16.cuda(), a, ptr_d, n, aggs, sc, 1.0)
        ge_ref = ops.pna_aggregate_bwd_aff(gout, g(e16.float()), a, ptr_d, n, aggs, sc, 1.0)
        assert torch.equal(ge, ge_ref)

