"""Per-family roofline fractions of the pre-training step, measured live with HIP events (20 launches per event pair, on
the shapes of a resident batch): the MFMA GEMMs of a PNA layer (forward, data gradient, weight gradient), the BatchNorm
passes, the aggregation kernel K4 (also on a batch whose working set is beyond the 256 MiB Infinity Cache).
bench.py imports `measure` and puts the result into its JSON line (`roofline.families`, `roofline.step`); stand-alone:

    python tools/family_bench.py [--batch 512]
"""
import argparse
import importlib
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

MFMA_F32_PEAK_TF = 157.3      # /opt/skills/guides/MI355X_MICROARCH.md: dense fp32 MFMA peak of MI355X
MFMA_BF16_PEAK_TF = 2500.0    # dense bf16 MFMA peak (same guide); used when ops.get_matmul_precision() == 'bf16'
HBM_PEAK_GBS = 8000.0


def _time(fn, reps=20):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3          # us


def _two_pass_bn(ops, dy, x, mean, invstd, gamma, beta, gb):
    was = ops.set_bn_bwd_one_launch(False)
    try:
        return _time(lambda: ops.bn_bwd(dy, x, None, None, None, mean, invstd, gamma, beta, grad_bias=gb))
    finally:
        ops.set_bn_bwd_one_launch(was)


def layer_flops(N, E, m_pad, F, n_comb=60):
    """MFMA flops of one PNA layer (hidden F, pretrans 2 blocks, grouped posttrans): forward, data gradients, weight gradients"""
    fwd = 2.0 * N * 2 * F * F + 2.0 * n_comb * F * F + 2.0 * E * F * F + 2.0 * N * F * F + 2.0 * m_pad * 4 * F * F
    dgrad = 2.0 * E * F * F + 2.0 * N * 2 * F * F + 2.0 * N * F * F + 2.0 * m_pad * 4 * F * F
    wgrad = 2.0 * E * F * F + 2.0 * N * 2 * F * F + 2.0 * N * F * F + 2.0 * N * 4 * F * F + 2.0 * E * 64 * F
    return fwd, dgrad, wgrad


def measure(amd, ops, g2, dev, F=200, big_batch=8192):
    """-> dict(families=..., shapes=...) for the bond graph `g2` (a resident batch)"""
    idx = g2.index()
    N, E = idx.num_nodes, idx.num_edges
    rows_d, tiles_d, groups = idx.degree_groups()
    m_pad, nG = rows_d.shape[0], len(groups)
    rnd = lambda *s: torch.randn(*s, device=dev)
    out = {}

    # ---- GEMM family (fp32 MFMA): the shapes of one layer, forward / dgrad / wgrad
    h, x1, agg = rnd(N, F), rnd(E, F).abs(), rnd(N, 4 * F)
    W1, W2, W3 = rnd(F, 3 * F) * 0.05, rnd(F, F) * 0.07, rnd(F, 13 * F) * 0.02
    WD = rnd(nG, F, 4 * F) * 0.03
    aff = torch.stack([x1.mean(0), torch.ones(F, device=dev), torch.zeros(F, device=dev)])
    bias = rnd(F)
    dY_e, dY_n, dP = rnd(E, F) * 0.1, rnd(N, F) * 0.1, rnd(N, 2 * F) * 0.1
    Wsd = torch.cat([W1[:, :F], W1[:, F:2 * F]], 0).contiguous()                # [2F, F]
    lin3 = rnd(N, F)
    Wcat = torch.cat([Wsd, W3[:, :F]], 0).contiguous()                          # [3F, F]
    bias3 = torch.cat([torch.zeros(2 * F, device=dev), bias])
    dPL = rnd(N, 3 * F) * 0.1
    # round 6: the products whose reduction is short (and the merged dL/dh product) run in row-panel form (csrc/panel.hip) on weights
    # packed once per step - what the step launches; the tiled forms stay listed next to them
    pk_cat, _, _ = ops.panel_pack(Wcat, True)
    pk_w2, _, _ = ops.panel_pack(W2, True)
    pk_w2d, _, _ = ops.panel_pack(W2, False)
    pk_catd, _, _ = ops.panel_pack(Wcat, False)
    tiled = {
        'tiled fwd PL [N,F]x[3F,F]^T': (lambda: ops.gemm(h, Wcat, trans_b=True, bias=bias3), 2.0 * N * 3 * F * F),
        'tiled fwd FC2 [E,F]x[F,F]^T + BN prologue + statistics': (lambda: ops.gemm_fused(x1, W2, bias, aff, None), 2.0 * E * F * F),
        'tiled dgrad FC2 [E,F]x[F,F]': (lambda: ops.gemm(dY_e, W2), 2.0 * E * F * F),
        'tiled dgrad DL [N,3F]x[3F,F]': (lambda: ops.gemm(dPL, Wcat), 2.0 * N * 3 * F * F),
    }
    gemms = {
        # the three products that read the node features are ONE GEMM per direction since round 3 (I3dPnaLayerArgs.merge_h):
        # PL = h [W_s ; W_d ; W_h]^T forward, dL/dh = [dP | dlin] [W_s ; W_d ; W_h] backward
        'fwd PL [N,F]x[3F,F]^T (P | post h), row-panel': (lambda: ops.panel_gemm(h, pk_cat, 3 * F, bias=bias3), 2.0 * N * 3 * F * F),
        'fwd FC2 [E,F]x[F,F]^T + BN prologue + statistics, row-panel': (lambda: ops.panel_gemm_fused(x1, pk_w2, F, bias, aff, None), 2.0 * E * F * F),
        'fwd post agg grouped [N,4F]->F + statistics': (
            lambda: ops.gemm_fused(agg, WD, None, None, None, out=lin3, accumulate=True, m_rows=rows_d, tile_group=tiles_d),
            2.0 * m_pad * 4 * F * F),
        'dgrad FC2 [E,F]x[F,F], row-panel': (lambda: ops.panel_gemm(dY_e, pk_w2d, F), 2.0 * E * F * F),
        'dgrad DL [N,3F]x[3F,F] (dP | dlin), row-panel': (lambda: ops.panel_gemm(dPL, pk_catd, F), 2.0 * N * 3 * F * F),
        'dgrad post agg grouped [N,F]->4F': (lambda: ops.gemm_grouped(dY_n, rows_d, tiles_d, WD, agg, trans_b=False, accumulate=False),
                                              2.0 * m_pad * 4 * F * F),
    }
    if not (ops.get_matmul_precision() == 'fp32' and ops.get_fp32_products() == 'split'):
        # bf16 matmul mode / native fp32 products: the step launches the tiled forms (csrc/composite.hip: panel_ok)
        gemms = {k: v for k, v in gemms.items() if 'row-panel' not in k}
        gemms.update({k.replace('tiled ', ''): v for k, v in tiled.items()})
        tiled = {}
    # all weight gradients of the layer: ONE launch + one reduction (csrc/wgrad.hip) in fp32; the per-product launches of
    # round 2 in the bf16 matmul mode (the panel kernel has no bf16 form yet)
    onehot = torch.zeros(E, 64, device=dev)
    onehot[torch.arange(E, device=dev), torch.randint(0, 60, (E,), device=dev)] = 1.0
    gW3, gW2, gW1, gQ = torch.empty(F, 13 * F, device=dev), torch.empty(F, F, device=dev), torch.empty(F, 3 * F, device=dev), torch.empty(64, F, device=dev)
    if ops.get_matmul_precision() == 'fp32' or os.environ.get('I3D_WGRAD_MULTI_BF16', '1') != '0':
        coef = [c for D, _, _ in groups for c in ((1.0, float(np.log(D + 1)), 1.0 / float(np.log(D + 1))) if D > 0 else (0.0, 0.0, 0.0))]
        live = [(s0, cnt, coef[3 * k:3 * k + 3]) for k, (D, s0, cnt) in enumerate(groups) if D > 0 and cnt > 0]
        problems = [dict(A=dY_n, B=h)] + [dict(A=dY_n, B=agg, rows=rows_d, k_begin=s0, k_count=cnt) for s0, cnt, _ in live]
        problems += [dict(A=dY_e, B=x1), dict(A=dP, B=h), dict(A=onehot, B=dY_e)]
        G = len(live)
        outputs = [dict(first_problem=0, C=gW3, ldc=13 * F),
                   dict(kind=ops.WGRAD_COMBINE, first_problem=1, n_groups=G, C=gW3, c_offset=F, ldc=13 * F,
                        coef=[c for _, _, cs in live for c in cs], n_scalers=3, scaler_stride=4 * F),
                   dict(kind=ops.WGRAD_BN, first_problem=1 + G, C=gW2, aff=aff.reshape(-1), row=bias),
                   dict(first_problem=2 + G, C=gW1, ldc=3 * F, c_split=F, c_delta=F - F * 3 * F),
                   dict(first_problem=3 + G, C=gQ)]
        gemms['wgrad: all five products of the layer, one launch + one reduction (K = N / E rows)'] = (
            lambda: ops.wgrad_multi(problems, outputs),
            2.0 * E * F * F + 2.0 * N * 2 * F * F + 2.0 * N * F * F + 2.0 * N * 4 * F * F + 2.0 * E * 64 * F)
    else:
        gemms['wgrad FC2 [F,F] K=E (BN fix-up)'] = (lambda: ops.gemm_wgrad_bn(dY_e, x1, bias, aff), 2.0 * E * F * F)
        gemms['wgrad P [2F,F] K=N'] = (lambda: ops.gemm(dP, h, trans_a=True), 2.0 * N * 2 * F * F)
        gemms['wgrad post h [F,F] K=N'] = (lambda: ops.gemm(dY_n, h, trans_a=True), 2.0 * N * F * F)
    rows, tot_us, tot_fl = [], 0.0, 0.0
    peak_tf = MFMA_BF16_PEAK_TF if ops.get_matmul_precision() == 'bf16' else MFMA_F32_PEAK_TF
    for name, (fn, fl) in gemms.items():
        us = _time(fn)
        rows.append(dict(kernel=name, us=round(us, 2), tflops=round(fl / us / 1e6, 1), frac=round(fl / us / 1e6 / peak_tf, 3)))
        tot_us += us
        tot_fl += fl
    tiled_rows = []
    if ops.get_matmul_precision() == 'fp32':
        for name, (fn, fl) in tiled.items():
            us = _time(fn)
            tiled_rows.append(dict(kernel=name, us=round(us, 2), tflops=round(fl / us / 1e6, 1), frac=round(fl / us / 1e6 / peak_tf, 3)))
    out['gemm'] = dict(bound='mfma', peak=peak_tf, unit='TFLOP/s', achieved=round(tot_fl / tot_us / 1e6, 1),
                       frac=round(tot_fl / tot_us / 1e6 / peak_tf, 3), kernels=rows, tiled_forms_of_the_row_panel_products=tiled_rows,
                       note='flop-weighted over the GEMM launches of one PNA layer (forward, data gradient, all weight gradients), '
                            'each launched 20 times back to back between one event pair')

    # ---- BatchNorm family (HBM): producer-fused statistics, backward reduction + apply, apply + residual
    P, Q = rnd(N, 2 * F), rnd(60, F)
    code = torch.randint(0, 60, (E,), device=dev, dtype=torch.int32)
    gamma, beta = torch.ones(F, device=dev), torch.zeros(F, device=dev)
    xact, partial, tiles = ops.edge_combine_act_stats(P, Q, bias, idx.src_s, idx.dst_s, 'relu', code)
    mean, invstd, _ = ops.bn_finalize_partials(partial, tiles, F, 1e-5, 0.1, gamma, beta)
    gb = torch.empty(F, device=dev)
    bn = {
        'edge gather-combine + ReLU + statistics [E,F]': (lambda: ops.edge_combine_act_stats(P, Q, bias, idx.src_s, idx.dst_s, 'relu', code),
                                                          4.0 * E * F * 3),
        'statistics finalisation': (lambda: ops.bn_finalize_partials(partial, tiles, F, 1e-5, 0.1, gamma, beta), 4.0 * tiles * 3 * F),
        # the chain's form since round 6: ONE launch (csrc/bn.hip: bn_bwd_fused_kernel; dy and x read once, the data gradient written:
        # 3 tensors) - the pretrans FC2 / posttrans blocks (no activation in front of the BatchNorm: exact-zero bias gradient)
        'backward, one launch: reduction + data gradient [E,F]': (
            lambda: ops.bn_bwd(dY_e, xact, None, None, None, mean, invstd, gamma, beta, grad_bias=gb), 4.0 * E * F * 3),
        'backward, one launch: reduction + data gradient [N,F]': (
            lambda: ops.bn_bwd(dY_n, lin3, None, None, None, mean, invstd, gamma, beta, grad_bias=gb), 4.0 * N * F * 3),
        'apply + residual [N,F]': (lambda: ops.bn_apply_fwd(lin3, mean, invstd, gamma, beta, None, h), 4.0 * N * F * 3),
    }
    rows, tot_us, tot_b = [], 0.0, 0.0
    for name, (fn, byts) in bn.items():
        us = _time(fn)
        rows.append(dict(kernel=name, us=round(us, 2), gbs=round(byts / us / 1e3, 1), frac=round(byts / us / 1e3 / HBM_PEAK_GBS, 3)))
        tot_us += us
        tot_b += byts
    out['batchnorm'] = dict(bound='hbm', peak=HBM_PEAK_GBS, unit='GB/s', achieved=round(tot_b / tot_us / 1e3, 1),
                            frac=round(tot_b / tot_us / 1e3 / HBM_PEAK_GBS, 3), kernels=rows,
                            two_pass_backward_us=round(_two_pass_bn(ops, dY_e, xact, mean, invstd, gamma, beta, gb), 2),
                            note='algorithmic bytes (one read per input tensor, one write per output tensor) / event time; two_pass_backward_us: '
                                 'the [E,F] backward as rounds 1-5 ran it (reduction launch + data-gradient launch, 5 tensor passes); at batch 512 the '
                                 'tensors (13.5 MB each) sit in the 256 MiB Infinity Cache, the kernels are bound by launch + finalisation latency')

    # ---- K4 (HBM): forward with the BatchNorm applied on load, backward; batch 512 and a batch beyond the Infinity Cache
    aggs, ident = ops.agg_codes(['mean', 'max', 'min', 'std']), ops.scaler_codes(['identity'])

    def k4(index, tag):
        n, e = index.num_nodes, index.num_edges
        msg, gout = torch.randn(e, F, device=dev), torch.randn(n, 4 * F, device=dev)
        us_f = _time(lambda: ops.pna_aggregate_fwd_aff(msg, aff, index.in_ptr, n, aggs, ident, 1.0))
        us_b = _time(lambda: ops.pna_aggregate_bwd_aff(gout, msg, aff, index.in_ptr, n, aggs, ident, 1.0))
        bf = 4.0 * e * F + 4.0 * n * 4 * F + 4.0 * (n + 1)
        bb = 4.0 * n * 4 * F + 2 * 4.0 * e * F + 4.0 * (n + 1)
        return {f'fwd {tag}': dict(us=round(us_f, 2), gbs=round(bf / us_f / 1e3, 1), frac=round(bf / us_f / 1e3 / HBM_PEAK_GBS, 3),
                                   algorithmic_bytes=int(bf)),
                f'bwd {tag}': dict(us=round(us_b, 2), gbs=round(bb / us_b / 1e3, 1), frac=round(bb / us_b / 1e3 / HBM_PEAK_GBS, 3),
                                   algorithmic_bytes=int(bb))}
    k = k4(idx, f'B={idx.num_graphs}')
    if big_batch:
        mols = amd.synth.make_dataset(big_batch, seed=77)
        big = amd.batch([amd.bond_graph(m) for m in mols]).to(dev).index()
        k.update(k4(big, f'B={big_batch} (working set {(4.0 * big.num_edges * F + 16.0 * big.num_nodes * F) / 2 ** 20:.0f} MiB > 256 MiB Infinity Cache)'))
    out['k4_aggregation'] = dict(bound='hbm', peak=HBM_PEAK_GBS, unit='GB/s', kernels=k,
                                 note='[N,4F] form as launched by the step, messages normalised on load (aff); back to back')
    return dict(families=out, shapes=dict(N=N, E=E, m_padded=m_pad, degree_groups=nG, F=F))


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=512)
    a = ap.parse_args()
    amd = importlib.import_module('3dinfomax_amd')
    ops = importlib.import_module('3dinfomax_amd.ops')
    dev = torch.device('cuda:0')
    g2 = amd.batch([amd.bond_graph(m) for m in amd.synth.make_dataset(a.batch, seed=1000)]).to(dev)
    print(json.dumps(measure(amd, ops, g2, dev), indent=1))
