"""bf16 matmul precision (i3d_set_matmul_precision(1), configs[3] of BASELINE.json): every GEMM of the library multiplies
bf16-rounded operands on the bf16 matrix pipe and accumulates in fp32; tensors in memory, BatchNorm statistics and master
weights stay fp32.

Kernel level: the result must equal the fp64 product of the ROUNDED operands to fp32-accumulation accuracy (this pins the
rounding point, the lane -> k mapping of v_mfma_f32_32x32x16_bf16 / 16x16x32_bf16 and every layout / tiling the step uses).
Model level: the documented bf16 tolerance against the fp32 path.
"""
import importlib
import math

import numpy as np
import pytest
import torch

from helpers import NET3D_YML, PNA_YML

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


@pytest.fixture(scope='module')
def amd():
    return importlib.import_module('3dinfomax_amd')


@pytest.fixture()
def ops(amd):
    o = importlib.import_module('3dinfomax_amd.ops')
    prev = o.set_matmul_precision('bf16')
    assert prev == 'fp32' and o.get_matmul_precision() == 'bf16'
    yield o
    o.set_matmul_precision('fp32')


def g(t):
    return t.to(DEV)


def rnd(*shape, seed=0, scale=1.0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale


def r16(t):
    """round-to-nearest-even to bf16, as fp64"""
    return t.float().bfloat16().double()


def rel_err(a, b):
    return ((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-30)).item()


@pytest.mark.parametrize('M,N,K', [(64, 200, 200), (1000, 200, 600), (777, 200, 2600), (513, 400, 200), (5000, 20, 60),
                                   (129, 208, 20), (16907, 200, 200)])
@pytest.mark.parametrize('ta,tb', [(False, True), (False, False), (True, False)])
def test_gemm_is_the_product_of_the_rounded_operands(ops, M, N, K, ta, tb):
    A = rnd(*((K, M) if ta else (M, K)), seed=1)
    B = rnd(*((N, K) if tb else (K, N)), seed=2)
    bias = rnd(N, seed=3)
    ref = (r16(A).T if ta else r16(A)) @ (r16(B).T if tb else r16(B)) + bias.double()
    out = ops.gemm(g(A), g(B), trans_a=ta, trans_b=tb, bias=g(bias))
    exact = (A.double().T if ta else A.double()) @ (B.double().T if tb else B.double()) + bias.double()
    vec = ((M if ta else K) % 4 == 0) and ((K if tb else N) % 4 == 0)     # 16-byte loads: the variants the step launches
    if N <= 32 or not vec:
        # the tilings of the narrow (hidden 20) products of the 3D network and the unaligned variants (not on the training
        # path) have no bf16 form: exact fp32 products, or the rounded ones where a bf16-shaped tiling is picked
        assert min(rel_err(out.cpu(), exact), rel_err(out.cpu(), ref)) < 2e-5
        return
    assert rel_err(out.cpu(), ref) < 2e-5
    # ... and differs from the exact fp32 product by what bf16 rounding costs (2^-9 per operand, random signs)
    e = rel_err(out.cpu(), exact)
    assert 1e-5 < e < 2e-2, e


def test_weight_gradient_split_k_through_the_scratch(ops):
    rows, Fo, Fi = 20000, 200, 200
    dY, X = rnd(rows, Fo, seed=4, scale=0.1), rnd(rows, Fi, seed=5)
    out = ops.gemm(g(dY), g(X), trans_a=True)
    assert rel_err(out.cpu(), r16(dY).T @ r16(X)) < 2e-5


@pytest.mark.parametrize('M,N,K', [(16907, 200, 200), (8511, 200, 800), (130, 200, 200)])
@pytest.mark.parametrize('prologue', [True, False])
def test_fused_gemm_rounds_after_the_batchnorm_prologue(ops, M, N, K, prologue):
    A, W, bias = rnd(M, K, seed=1) + 2.0, rnd(N, K, seed=2, scale=K ** -0.5), rnd(N, seed=3)
    aff, A_eff = None, A
    if prologue:
        aff = torch.stack([rnd(K, seed=4) + 2.0, 1 + 0.3 * rnd(K, seed=5), 0.3 * rnd(K, seed=6)])
        A_eff = (A - aff[0]) * aff[1] + aff[2]                      # fp32, as the kernel stages it
    ref = torch.relu(r16(A_eff) @ r16(W).T + bias.double())
    out, partial, tiles = ops.gemm_fused(g(A), g(W), g(bias), g(aff) if prologue else None, 'relu')
    # (a product that lands within an ulp of a bf16 rounding boundary of A_eff may round the other way: fmaf vs two
    # roundings in the reference expression) -> 1e-4
    assert rel_err(out.cpu(), ref) < 1e-4
    # the statistics describe the stored values
    s = partial.cpu().double()
    assert abs(s[:, 0, :].sum(0) - out.cpu().double().sum(0)).max().item() < 1e-3 * out.abs().sum(0).max().item()


def test_grouped_and_row_subset_gemms(ops):
    graph = importlib.import_module('3dinfomax_amd.graph')
    n, Fo, A = 1500, 200, 800
    rng = np.random.default_rng(0)
    indeg = rng.choice([1, 2, 3, 4, 6], size=n, p=[0.45, 0.1, 0.13, 0.3, 0.02])
    rows, tiles_g, groups = graph.group_nodes_by_degree(indeg)
    agg, dY = rnd(n, A, seed=1), rnd(n, Fo, seed=3)
    WD = rnd(len(groups), Fo, A, seed=2, scale=A ** -0.5)
    rows_d, tiles_d = g(torch.from_numpy(rows)), g(torch.from_numpy(tiles_g))
    out = torch.zeros(n, Fo, device=DEV)
    ops.gemm_grouped(g(agg), rows_d, tiles_d, g(WD), out, trans_b=True, accumulate=True)
    ga = torch.zeros(n, A, device=DEV)
    ops.gemm_grouped(g(dY), rows_d, tiles_d, g(WD), ga, trans_b=False, accumulate=False)
    gWD = torch.zeros_like(g(WD))
    ops.gemm_rowsubset_multi(g(dY), g(agg), rows_d, [st for _, st, _ in groups], [ct for _, _, ct in groups], gWD)
    for gi, (D, start, count) in enumerate(groups):
        r = torch.from_numpy(rows[start:start + count]).long()
        assert rel_err(out.cpu()[r], r16(agg[r]) @ r16(WD[gi]).T) < 2e-5
        assert rel_err(ga.cpu()[r], r16(dY[r]) @ r16(WD[gi])) < 2e-5
        assert rel_err(gWD.cpu()[gi], r16(dY[r]).T @ r16(agg[r])) < 2e-5


def _pretrain_step(amd, mols, depth):
    torch.manual_seed(3)
    pna = amd.PNA(avg_d=1.0, device=DEV, **dict(PNA_YML, propagation_depth=depth)).to(DEV).train()
    net = amd.Net3D(node_dim=0, edge_dim=1, avg_d=1.0, **NET3D_YML).to(DEV).train()
    crit = amd.NTXent(tau=0.1)
    g2 = amd.batch([amd.bond_graph(m) for m in mols]).to(DEV)
    g3 = amd.batch([amd.complete_graph(m) for m in mols]).to(DEV)
    z2, z3 = pna(g2), net(g3)
    loss = crit(z2, z3)
    loss.backward()
    grads = torch.cat([p.grad.flatten() for p in pna.parameters()])
    return loss.item(), z2.detach().clone(), g2.ndata['feat'].detach().clone(), grads


def test_pretraining_step_bf16_against_fp32(amd):
    """The documented tolerance of the bf16 matmul mode on the pre-training step (hidden 200, depth 4, 256 molecules,
    init-like weights): loss within 2e-3 relative, projection-head and node embeddings within 2e-2 of their scale, the
    parameter gradient within 1e-1 in relative L2 - measured 0.068 (bf16 operands: 2^-9 relative rounding per element,
    amplified by the BatchNorm of every block; statistics and accumulation in fp32)."""
    o = importlib.import_module('3dinfomax_amd.ops')
    mols = amd.synth.make_dataset(256, seed=5)
    ref = _pretrain_step(amd, mols, 4)
    prev = o.set_matmul_precision('bf16')
    try:
        out = _pretrain_step(amd, mols, 4)
    finally:
        o.set_matmul_precision(prev)
    assert out[0] != ref[0]                                       # the mode is what ran
    assert abs(out[0] - ref[0]) <= 2e-3 * abs(ref[0]), (out[0], ref[0])
    for a, b in ((out[1], ref[1]), (out[2], ref[2])):
        assert (a - b).abs().max().item() <= 2e-2 * b.abs().max().item()
    assert ((out[3] - ref[3]).norm() / ref[3].norm()).item() <= 1e-1
    assert math.isfinite(out[0])
