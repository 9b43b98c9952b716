"""FCLayer / MLP towers and the autograd.Functions that chain the HIP kernels.

Mirror of reference models/base_layers.py (FCLayer :23-111, MLP :114-147): same constructor kwargs, same
sub-module names (`linear`, `batch_norm`, `fully_connected`) hence the same state_dict keys, same init
(xavier_uniform with gain 1/in_dim, zero bias), same op order Linear -> activation -> dropout -> BatchNorm.
nn.Linear / nn.BatchNorm1d are used as PARAMETER CONTAINERS only - their forward is never called; all
arithmetic runs in the gfx950 kernels through ops.py, forward and hand-written backward.
"""
import ctypes
import os
import threading
from dataclasses import dataclass
from typing import Optional

import torch
import torch.nn as nn

from . import _lib, ops, tape
from .streams import fork

SUPPORTED_ACTIVATIONS = {'relu', 'silu', 'sigmoid', 'leakyrelu', 'tanh', 'elu', 'selu', 'softplus', 'none'}
# activations that exist as an elementwise pass only (csrc/common.h: apply_act_any): the fused kernels - GEMM epilogues, statistics /
# BatchNorm passes - keep the four the reference's configurations use (their switch is inlined into every kernel of the step)
ELEMENTWISE_ONLY = {'tanh', 'elu', 'selu', 'softplus'}


def act_name(activation):
    """reference models/base_layers.py:9-20 (get_activation): case-insensitive name or None."""
    if activation is None:
        return None
    if callable(activation) and not isinstance(activation, str):
        activation = type(activation).__name__
    a = activation.lower()
    if a not in SUPPORTED_ACTIVATIONS:
        raise NotImplementedError(f'activation {activation!r} has no HIP kernel yet (supported: ReLU, SiLU, Sigmoid, LeakyReLU, Tanh, ELU, SELU, Softplus, None)')
    return None if a == 'none' else a


@dataclass
class BNSpec:
    """Everything the fused (activation +) BatchNorm needs besides the affine parameters."""
    running_mean: Optional[torch.Tensor]
    running_var: Optional[torch.Tensor]
    num_batches_tracked: Optional[torch.Tensor]
    momentum: float
    eps: float
    training: bool
    sync_group: object = None        # torch.distributed process group for synchronised statistics (or None)


@dataclass
class FCSpec:
    act: Optional[str]               # activation between Linear and BatchNorm
    bn: Optional[BNSpec]
    post_act: Optional[str] = None   # activation after the BatchNorm (Net3D edge_input: silu(BN(silu(.))))
    dropout: float = 0.0             # nn.Dropout between the activation and the BatchNorm (reference models/base_layers.py:104-105);
    #                                  0.0 in eval mode.  > 0: the per-kernel path (the composites and native sequencers stand aside)


# num_batches_tracked of every BatchNorm touched in a model forward is bumped by ONE multi-tensor op at the end of that
# forward (a torch op costs ~35 us of host time on this stack; 16-25 separate add_ calls were 0.6 ms per step).
_counters = threading.local()


class bn_counter_scope:
    """with bn_counter_scope(): ... model forward ...   -> one torch._foreach_add_ over the pending counters at exit."""

    def __enter__(self):
        _counters.depth = getattr(_counters, 'depth', 0) + 1
        return self

    def __exit__(self, *exc):
        _counters.depth -= 1
        if _counters.depth == 0:
            flush_bn_counters()
        return False


def flush_bn_counters():
    pending = getattr(_counters, 'pending', None)
    if pending:
        torch._foreach_add_(pending, 1)
        _counters.pending = []


def _bump(counter):
    if counter is None:
        return
    if getattr(_counters, 'depth', 0) > 0:
        if getattr(_counters, 'pending', None) is None:
            _counters.pending = []
        _counters.pending.append(counter)
    else:
        counter.add_(1)


class _Tail:
    """activation + BatchNorm part of an FCLayer, shared by the three FC autograd.Functions.

    forward(pre)  -> y, saved tuple         backward(saved, grad_y) -> grad_pre, grad_gamma, grad_beta"""

    @staticmethod
    def forward(pre, gamma, beta, spec: FCSpec, residual=None):
        bn = spec.bn
        if spec.dropout > 0.0 or spec.act in ELEMENTWISE_ONLY or spec.post_act in ELEMENTWISE_ONLY:
            return _Tail._forward_split(pre, gamma, beta, spec, residual)
        if bn is None:
            acts = [a for a in (spec.act, spec.post_act) if a is not None]
            inputs, y = [], pre
            for a in acts:
                inputs.append(y)
                y = ops.act_fwd(y, a)
            if residual is not None:
                y = ops.add_inplace(y.clone() if y is pre else y, residual)
            return y, (inputs, acts)
        keep_pre = spec.act not in (None, 'relu', 'leakyrelu')   # silu'/sigmoid' need the pre-activation
        if bn.training:
            if bn.sync_group is not None:
                from . import dist as adist
                sums = torch.empty(2 * pre.shape[1] + 1, dtype=torch.float64, device=pre.device)
                x, _, _ = ops.act_stats_fwd(pre, spec.act, bn.eps, bn.momentum, sums_out=sums,
                                            out=torch.empty_like(pre) if keep_pre else None)
                adist.all_reduce_sum(sums, bn.sync_group)
                mean, invstd = ops.bn_finalize_stats(sums, pre.shape[1], bn.eps, bn.momentum, bn.running_mean,
                                                     bn.running_var)
            else:
                x, mean, invstd = ops.act_stats_fwd(pre, spec.act, bn.eps, bn.momentum, bn.running_mean, bn.running_var,
                                                    out=torch.empty_like(pre) if keep_pre else None)
            _bump(bn.num_batches_tracked)
            y = ops.bn_apply_fwd(x, mean, invstd, gamma, beta, spec.post_act, residual)
            return y, (x, pre if keep_pre else None, mean, invstd)
        x = ops.act_fwd(pre, spec.act) if spec.act is not None else pre
        y = ops.bn_eval_fwd(x, bn.running_mean, bn.running_var, bn.eps, gamma, beta, spec.post_act, residual)
        return y, (x, pre if keep_pre else None, bn.running_mean.clone(), bn.running_var.clone())

    @staticmethod
    def _forward_split(pre, gamma, beta, spec: FCSpec, residual):
        """Linear -> activation -> DROPOUT -> BatchNorm -> post-activation (reference models/base_layers.py:100-111) with the
        activation as a pass of its own: the form for dropout > 0 (the mask is the one torch's dropout kernel draws for this shape
        - ops.dropout_mask: the reference module's mask for the same seed on this device - and the BatchNorm statistics are those
        of the dropped values; training mode only, spec.dropout is 0 in eval) and for the activations the fused kernels do not
        carry (ELEMENTWISE_ONLY)."""
        bn = spec.bn
        a = ops.act_fwd(pre, spec.act) if spec.act is not None else pre
        m = ops.dropout_mask(a, spec.dropout) if spec.dropout > 0.0 else None
        x = ops.mul(a, m) if m is not None else a
        post = spec.post_act
        fused_post = post if post not in ELEMENTWISE_ONLY else None       # (applied by the BatchNorm apply kernel, with the residual)
        if bn is None:
            mean = stat2 = None
            y0 = x
            y = ops.act_fwd(x, post) if post is not None else x
            if residual is not None:
                y = ops.add_inplace(y.clone() if (y is x and (x is pre or x is a)) else y, residual)
            return y, ('split', pre, a, x, m, None, None, y0)
        if bn.training:
            if bn.sync_group is not None:
                from . import dist as adist
                sums = torch.empty(2 * x.shape[1] + 1, dtype=torch.float64, device=x.device)
                ops.act_stats_fwd(x, None, bn.eps, bn.momentum, sums_out=sums)
                adist.all_reduce_sum(sums, bn.sync_group)
                mean, stat2 = ops.bn_finalize_stats(sums, x.shape[1], bn.eps, bn.momentum, bn.running_mean, bn.running_var)
            else:
                _, mean, stat2 = ops.act_stats_fwd(x, None, bn.eps, bn.momentum, bn.running_mean, bn.running_var)
            _bump(bn.num_batches_tracked)
            y0 = ops.bn_apply_fwd(x, mean, stat2, gamma, beta, fused_post, residual if fused_post is post else None)
        else:
            mean, stat2 = bn.running_mean.clone(), bn.running_var.clone()
            y0 = ops.bn_eval_fwd(x, bn.running_mean, bn.running_var, bn.eps, gamma, beta, fused_post,
                                 residual if fused_post is post else None)
        if fused_post is post:
            return y0, ('split', pre, a, x, m, mean, stat2, None)
        y = ops.act_fwd(y0, post)                       # elementwise-only post-activation: y0 = the BatchNorm's output, kept
        if residual is not None:
            y = ops.add_inplace(y, residual)
        return y, ('split', pre, a, x, m, mean, stat2, y0)

    @staticmethod
    def _backward_split(saved, grad_y, gamma, beta, spec: FCSpec):
        _, pre, a, x, m, mean, stat2, y0 = saved
        bn = spec.bn
        post = spec.post_act
        fused_post = post if post not in ELEMENTWISE_ONLY else None
        gg = gb = None
        g = grad_y
        if bn is None:
            if post is not None:
                g = ops.act_bwd(g, x, post)
        else:
            if fused_post is not post:
                g = ops.act_bwd(g, y0, post)            # d / d(BatchNorm output)
            if not bn.training:
                g, gg, gb = ops.bn_eval_bwd(g, x, None, None, fused_post, mean, stat2, bn.eps, gamma, beta)
            elif bn.sync_group is not None:
                from . import dist as adist
                feat = x.shape[1]
                sums = torch.empty(2 * feat + 1, dtype=torch.float64, device=x.device)
                gg = torch.empty(feat, dtype=torch.float32, device=x.device)
                gb = torch.empty(feat, dtype=torch.float32, device=x.device)
                ops.bn_bwd(g, x, None, None, fused_post, mean, stat2, gamma, beta, sums_out=sums, grad_gamma=gg, grad_beta=gb, out=g)
                sums[2 * feat:].fill_(x.shape[0])
                adist.all_reduce_sum(sums, bn.sync_group)
                g, _, _ = ops.bn_bwd(g, x, None, None, fused_post, mean, stat2, gamma, beta, sums_in=sums, grad_gamma=gg, grad_beta=gb)
            else:
                g, gg, gb = ops.bn_bwd(g, x, None, None, fused_post, mean, stat2, gamma, beta)      # d / d(dropped value)
        if m is not None:
            g = ops.mul(g, m)                                                                   # d / d(activation)
        if spec.act is not None:
            g = ops.act_bwd(g, pre, spec.act)                                                   # d / d(pre-activation)
        return g, gg, gb

    @staticmethod
    def backward(saved, grad_y, gamma, beta, spec: FCSpec):
        bn = spec.bn
        if len(saved) == 8 and isinstance(saved[0], str) and saved[0] == 'split':
            return _Tail._backward_split(saved, grad_y, gamma, beta, spec)
        if bn is None:
            inputs, acts = saved
            grad_pre = grad_y
            for x_in, a in zip(reversed(inputs), reversed(acts)):
                grad_pre = ops.act_bwd(grad_pre, x_in, a)
            return grad_pre, None, None
        x, pre, mean, stat2 = saved
        if bn.training:
            if bn.sync_group is not None:
                from . import dist as adist
                feat = x.shape[1]
                sums = torch.empty(2 * feat + 1, dtype=torch.float64, device=x.device)
                gg = torch.empty(feat, dtype=torch.float32, device=x.device)
                gb = torch.empty(feat, dtype=torch.float32, device=x.device)
                ops.bn_bwd(grad_y, x, pre, spec.act, spec.post_act, mean, stat2, gamma, beta, sums_out=sums,
                           grad_gamma=gg, grad_beta=gb, out=grad_y)
                sums[2 * feat:].fill_(x.shape[0])          # local row count rides along in the all-reduce
                adist.all_reduce_sum(sums, bn.sync_group)
                grad_pre, _, _ = ops.bn_bwd(grad_y, x, pre, spec.act, spec.post_act, mean, stat2, gamma, beta,
                                            sums_in=sums, grad_gamma=gg, grad_beta=gb)
                return grad_pre, gg, gb
            return ops.bn_bwd(grad_y, x, pre, spec.act, spec.post_act, mean, stat2, gamma, beta)
        return ops.bn_eval_bwd(grad_y, x, pre, spec.act, spec.post_act, mean, stat2, bn.eps, gamma, beta)


# ------------------------------------------------------------------------------------------------------------------
# Composite fast path (csrc/composite.hip): one C call enqueues the whole block.  Eligible: BatchNorm in training mode
# with local statistics.  I3D_COMPOSITE=0 forces the per-kernel path (identical arithmetic, used by the tests to
# cross-check).
COMPOSITE = True


def _composite_ok(spec: FCSpec, *tensors):
    return (COMPOSITE and spec.bn is not None and spec.bn.training and spec.bn.sync_group is None and spec.dropout == 0.0
            and spec.act not in ELEMENTWISE_ONLY and spec.post_act not in ELEMENTWISE_ONLY
            and all(t is None or t.is_cuda for t in tensors))


def _ptr(t):
    return t.data_ptr() if t is not None else None


def _f32(shape, device):
    return torch.empty(shape, dtype=torch.float32, device=device)


def _fill_tail(tail, spec: FCSpec, gamma, beta, mean, invstd, feat, device):
    bn = spec.bn
    tail.act, tail.post_act = ops.ACT[spec.act], ops.ACT[spec.post_act]
    tail.eps, tail.momentum = bn.eps, bn.momentum
    tail.gamma, tail.beta = gamma.data_ptr(), beta.data_ptr()
    tail.running_mean, tail.running_var = bn.running_mean.data_ptr(), bn.running_var.data_ptr()
    tail.mean, tail.invstd = mean.data_ptr(), invstd.data_ptr()
    nbt = bn.num_batches_tracked           # bumped by the statistics kernel (no _foreach_add_ per model forward)
    tail.num_batches_tracked = nbt.data_ptr() if (nbt is not None and nbt.is_cuda) else None
    _set_workspaces(tail, feat, device)


def _set_workspaces(tail, feat, device):
    # per thread and stream (the backward runs on autograd's thread)
    tail.workspace = ops._workspace(feat, device).data_ptr()
    tail.gemm_workspace, tail.gemm_workspace_bytes = ops._gemm_workspace(device).data_ptr(), ops.GEMM_WORKSPACE_BYTES


def _keeps_pre(spec):
    return spec.act not in (None, 'relu', 'leakyrelu')


def _call(fn_name, args):
    L = _lib.load()
    _lib.check(getattr(L, fn_name)(ctypes.byref(args), ops._stream()), fn_name)


class FCFn(torch.autograd.Function):
    """y = post_act(BN(act(x W^T + b))) (+ residual).  reference models/base_layers.py:100-111."""

    @staticmethod
    def forward(ctx, x, W, b, gamma, beta, residual, spec: FCSpec):
        x = x.contiguous()
        ctx.spec, ctx.has_res, ctx.cargs = spec, residual is not None, None
        if _composite_ok(spec, x, W) and W.is_contiguous():
            rows, f_out, dev = x.shape[0], W.shape[0], x.device
            xact, y = _f32((rows, f_out), dev), _f32((rows, f_out), dev)
            pre_keep = _f32((rows, f_out), dev) if _keeps_pre(spec) else None
            mean, invstd = _f32((f_out,), dev), _f32((f_out,), dev)
            a = _lib.FcArgs()
            _fill_tail(a.tail, spec, gamma, beta, mean, invstd, f_out, dev)
            a.rows, a.f_in, a.f_out, a.ldw = rows, x.shape[1], f_out, W.stride(0)
            a.x, a.W, a.bias, a.residual = x.data_ptr(), W.data_ptr(), _ptr(b), _ptr(residual)
            a.xact, a.pre_keep, a.y = xact.data_ptr(), _ptr(pre_keep), y.data_ptr()
            _call('i3d_fc_bn_fwd', a)
            ctx.cargs, ctx.saved = a, (xact, pre_keep, mean, invstd)
            ctx.save_for_backward(x, W, gamma, beta)
            return y
        pre = ops.gemm(x, W, trans_b=True, bias=b)
        y, saved = _Tail.forward(pre, gamma, beta, spec, residual)
        ctx.saved = saved
        ctx.save_for_backward(x, W, gamma, beta)
        return y

    @staticmethod
    def backward(ctx, grad_y):
        x, W, gamma, beta = ctx.saved_tensors
        grad_y = grad_y.contiguous()
        grad_res = grad_y if ctx.has_res else None
        if ctx.cargs is not None:
            a, dev, f_out = ctx.cargs, x.device, W.shape[0]
            grad_pre = _f32((x.shape[0], f_out), dev)
            gg, gb, gbias = tape.grad_like(gamma), tape.grad_like(beta), tape.grad_for_bias_of(W, f_out)
            gW = tape.grad_like(W)
            gx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
            _set_workspaces(a.tail, f_out, dev)
            a.grad_y, a.grad_pre, a.grad_gamma, a.grad_beta = grad_y.data_ptr(), grad_pre.data_ptr(), gg.data_ptr(), gb.data_ptr()
            a.grad_W, a.grad_bias, a.grad_x = gW.data_ptr(), gbias.data_ptr(), _ptr(gx)
            _call('i3d_fc_bn_bwd', a)
            return gx, gW, gbias, gg, gb, grad_res, None
        grad_pre, gg, gb = _Tail.backward(ctx.saved, grad_y, gamma, beta, ctx.spec)
        gW = tape.grad_like(W) if ctx.needs_input_grad[1] else None
        gbias = tape.grad_for_bias_of(W, W.shape[0]) if ctx.needs_input_grad[2] else None
        with fork(grad_pre, x) as f:            # weight/bias gradients next to the data gradient (streams.py)
            if gW is not None:
                ops.gemm(grad_pre, x, trans_a=True, out=gW)
            if gbias is not None:
                ops.colsum(grad_pre, out=gbias)
        gx = ops.gemm(grad_pre, W) if ctx.needs_input_grad[0] else None
        f.join()
        return gx, gW, gbias, gg, gb, grad_res, None


class Concat2FCFn(torch.autograd.Function):
    """FC layer on the column concatenation [a | c] without materialising it:
    pre = a W[:, :Fa]^T + c W[:, Fa:]^T + b   (reference models/pna.py:207-209: cat([h, agg]) -> posttrans)."""

    @staticmethod
    def forward(ctx, a, c, W, b, gamma, beta, residual, spec: FCSpec):
        Fa = a.shape[1]
        pre = ops.gemm(a, W[:, :Fa], trans_b=True, bias=b)
        ops.gemm(c, W[:, Fa:], trans_b=True, out=pre, accumulate=True)
        y, saved = _Tail.forward(pre, gamma, beta, spec, residual)
        ctx.spec, ctx.saved, ctx.has_res = spec, saved, residual is not None
        ctx.save_for_backward(a, c, W, gamma, beta)
        return y

    @staticmethod
    def backward(ctx, grad_y):
        a, c, W, gamma, beta = ctx.saved_tensors
        Fa = a.shape[1]
        grad_y = grad_y.contiguous()
        grad_res = grad_y if ctx.has_res else None
        grad_pre, gg, gb = _Tail.backward(ctx.saved, grad_y, gamma, beta, ctx.spec)
        gW = tape.grad_like(W)
        gbias = tape.grad_for_bias_of(W, W.shape[0])
        with fork(grad_pre, a, c) as f:
            ops.gemm(grad_pre, c, trans_a=True, out=gW[:, Fa:])
            ops.gemm(grad_pre, a, trans_a=True, out=gW[:, :Fa])
            ops.colsum(grad_pre, out=gbias)
        gc = ops.gemm(grad_pre, W[:, Fa:])
        ga = ops.gemm(grad_pre, W[:, :Fa])
        f.join()
        return ga, gc, gW, gbias, gg, gb, grad_res, None


class GroupedConcat2FCFn(torch.autograd.Function):
    """FC layer on [h | a | s_2(D) a | s_3(D) a ...] where only the aggregator block `a` [N, A] exists: the scaler
    blocks are per-node multiples that depend on the in-degree D alone, so nodes of one degree share the combined weight
    W_D = sum_s c_s(D) W_s  (csrc/grouped.hip).  pre = h W_h^T + b + a[deg group] W_D^T  - K of the dominant GEMM drops from
    S*A to A.  reference models/pna.py:207-209 + 229-233, re-associated.  `coef` = [[c_s(D) for s] for D in groups]."""

    @staticmethod
    def forward(ctx, h, a, W, b, gamma, beta, residual, index, coef, spec: FCSpec):
        h, a = h.contiguous(), a.contiguous()
        Fh, A = h.shape[1], a.shape[1]
        rows, tiles, groups = index.degree_groups()
        nG, nS = len(groups), len(coef[0])
        flat = [c for g in coef for c in g]
        ctx.cargs = None
        if _composite_ok(spec, h, a, W) and W.is_contiguous() and nG <= 32 and nG * nS <= 128:
            N, f_out, dev = h.shape[0], W.shape[0], h.device
            xact, y = _f32((N, f_out), dev), _f32((N, f_out), dev)
            pre_keep = _f32((N, f_out), dev) if _keeps_pre(spec) else None
            mean, invstd = _f32((f_out,), dev), _f32((f_out,), dev)
            WD = _f32((nG, f_out, A), dev)
            c = _lib.GroupedFcArgs()
            _fill_tail(c.tail, spec, gamma, beta, mean, invstd, f_out, dev)
            c.num_nodes, c.f_h, c.f_out, c.agg_width, c.ldw = N, Fh, f_out, A, W.stride(0)
            c.n_groups, c.n_scalers, c.m_padded = nG, nS, rows.shape[0]
            for gi, (_, start, count) in enumerate(groups):
                c.group_start[gi], c.group_count[gi] = start, count
            for i, v in enumerate(flat):
                c.coef[i] = v
            c.h, c.agg, c.W, c.bias, c.residual = h.data_ptr(), a.data_ptr(), W.data_ptr(), _ptr(b), _ptr(residual)
            c.deg_rows, c.deg_tile_group = rows.data_ptr(), tiles.data_ptr()
            c.WD, c.xact, c.pre_keep, c.y = WD.data_ptr(), xact.data_ptr(), _ptr(pre_keep), y.data_ptr()
            _call('i3d_grouped_fc_bn_fwd', c)
            ctx.cargs, ctx.saved = c, (xact, pre_keep, mean, invstd)
            ctx.spec, ctx.has_res, ctx.index, ctx.coef = spec, residual is not None, index, (flat, nG, nS)
            ctx.save_for_backward(h, a, W, WD, gamma, beta)
            return y
        pre = ops.gemm(h, W[:, :Fh], trans_b=True, bias=b)
        WD = ops.combine_weights_fwd(W, Fh, A, flat, nG, nS)
        ops.gemm_grouped(a, rows, tiles, WD, pre, trans_b=True, accumulate=True)
        y, saved = _Tail.forward(pre, gamma, beta, spec, residual)
        ctx.spec, ctx.saved, ctx.has_res, ctx.index, ctx.coef = spec, saved, residual is not None, index, (flat, nG, nS)
        ctx.save_for_backward(h, a, W, WD, gamma, beta)
        return y

    @staticmethod
    def backward(ctx, grad_y):
        h, a, W, WD, gamma, beta = ctx.saved_tensors
        Fh, A = h.shape[1], a.shape[1]
        flat, nG, nS = ctx.coef
        rows, tiles, groups = ctx.index.degree_groups()
        grad_y = grad_y.contiguous()
        grad_res = grad_y if ctx.has_res else None
        if ctx.cargs is not None:
            c, dev, f_out, N = ctx.cargs, h.device, W.shape[0], h.shape[0]
            grad_pre, gWD = _f32((N, f_out), dev), torch.empty_like(WD)
            gg, gb, gbias = _f32((f_out,), dev), _f32((f_out,), dev), _f32((f_out,), dev)
            gW, gh, ga = torch.empty_like(W), torch.empty_like(h), torch.empty_like(a)
            _set_workspaces(c.tail, f_out, dev)
            c.grad_y, c.grad_pre, c.grad_WD = grad_y.data_ptr(), grad_pre.data_ptr(), gWD.data_ptr()
            c.grad_gamma, c.grad_beta, c.grad_W, c.grad_bias = gg.data_ptr(), gb.data_ptr(), gW.data_ptr(), gbias.data_ptr()
            c.grad_h, c.grad_agg = gh.data_ptr(), ga.data_ptr()
            _call('i3d_grouped_fc_bn_bwd', c)
            return gh, ga, gW, gbias, gg, gb, grad_res, None, None, None
        grad_pre, gg, gb = _Tail.backward(ctx.saved, grad_y, gamma, beta, ctx.spec)
        gW = torch.empty_like(W)
        ops.gemm(grad_pre, h, trans_a=True, out=gW[:, :Fh])
        gWD = torch.empty_like(WD)
        # dW_D = dY_D^T a_D over the rows of each in-degree group, one launch for all groups
        ops.gemm_rowsubset_multi(grad_pre, a, rows, [st for _, st, _ in groups], [ct for _, _, ct in groups], gWD)
        ops.combine_weights_bwd(gWD, gW, Fh, A, flat, nG, nS)
        gbias = ops.colsum(grad_pre)
        gh = ops.gemm(grad_pre, W[:, :Fh])
        ga = torch.empty_like(a)        # rows of in-degree 0 stay unwritten: the aggregation backward never reads them
        ops.gemm_grouped(grad_pre, rows, tiles, WD, ga, trans_b=False, accumulate=False)
        return gh, ga, gW, gbias, gg, gb, grad_res, None, None, None


class EdgeTable:
    """Categorical edge features with a small joint vocabulary (ops.edge_codes): the edge operand q of EdgeFCFn is then
    the [V, F] table of all combinations, `codes` [E] (destination-sorted order) names the row of every edge and
    `onehot` [E, v_pad] drives the gradient reductions."""
    __slots__ = ('codes', 'onehot', 'rows', 'v_pad')

    def __init__(self, codes, onehot, rows, v_pad):
        self.codes, self.onehot, self.rows, self.v_pad = codes, onehot, rows, v_pad


class EdgeFCFn(torch.autograd.Function):
    """First FC layer of an edge MLP on [h_src | h_dst | q] (q optional) without the gather/concat:
        P = h [W_s | W_d]^T  (node level),  Q = q W_q^T,  pre[j] = P[src_j, :F] + P[dst_j, F:] + Q[j] + b
    reference models/pna.py:237-252 (pretrans_edges), models/net3d.py:113-115 (message_function).
    Edge tensors are in destination-sorted order; backward of the gathers = segmented sums (no atomics)."""

    @staticmethod
    def forward(ctx, h, q, W, b, gamma, beta, index, spec: FCSpec, qmap: EdgeTable = None):
        h = h.contiguous()
        Fh = h.shape[1]
        Fo = W.shape[0]
        N = h.shape[0]
        ctx.cargs = None
        ctx.qmap = qmap
        if q is not None:
            q = q.contiguous()
            assert qmap is None or q.shape[0] == qmap.rows
        if _composite_ok(spec, h, q, W) and W.is_contiguous() and index.num_edges > 0:
            E, dev = index.num_edges, h.device
            P = _f32((N, 2 * Fo), dev)
            Q = _f32((qmap.rows if qmap is not None else E, Fo), dev) if q is not None else None
            xact, y = _f32((E, Fo), dev), _f32((E, Fo), dev)
            pre_keep = _f32((E, Fo), dev) if _keeps_pre(spec) else None
            mean, invstd = _f32((Fo,), dev), _f32((Fo,), dev)
            a = _lib.EdgeFcArgs()
            _fill_tail(a.tail, spec, gamma, beta, mean, invstd, Fo, dev)
            a.num_nodes, a.num_edges, a.f_h, a.f_q, a.f_out, a.ldw = N, E, Fh, (q.shape[1] if q is not None else 0), Fo, W.stride(0)
            a.h, a.q, a.W, a.bias = h.data_ptr(), _ptr(q), W.data_ptr(), _ptr(b)
            if qmap is not None and q is not None:
                a.q_rows, a.v_pad, a.q_code, a.onehot = qmap.rows, qmap.v_pad, qmap.codes.data_ptr(), qmap.onehot.data_ptr()
            a.src_s, a.dst_s, a.in_ptr = index.src_s.data_ptr(), index.dst_s.data_ptr(), index.in_ptr.data_ptr()
            a.out_ptr, a.out_epos = index.out_ptr.data_ptr(), index.out_epos.data_ptr()
            a.P, a.Q, a.xact, a.pre_keep, a.y = P.data_ptr(), _ptr(Q), xact.data_ptr(), _ptr(pre_keep), y.data_ptr()
            _call('i3d_edge_fc_bn_fwd', a)
            ctx.cargs, ctx.saved = a, (xact, pre_keep, mean, invstd)
            ctx.spec, ctx.index, ctx.has_q = spec, index, q is not None
            ctx.save_for_backward(h, q if q is not None else h, W, gamma, beta)
            return y
        P = torch.empty(N, 2 * Fo, dtype=torch.float32, device=h.device)
        ops.gemm(h, W[:, :Fh], trans_b=True, out=P[:, :Fo])
        ops.gemm(h, W[:, Fh:2 * Fh], trans_b=True, out=P[:, Fo:])
        Q = None
        if q is not None:
            Q = ops.gemm(q, W[:, 2 * Fh:], trans_b=True)
        pre = ops.edge_combine_fwd(P, Q, b, index.src_s, index.dst_s,
                                   q_code=qmap.codes if (qmap is not None and q is not None) else None)
        y, saved = _Tail.forward(pre, gamma, beta, spec)
        ctx.spec, ctx.saved, ctx.index, ctx.has_q = spec, saved, index, q is not None
        ctx.save_for_backward(h, q if q is not None else h, W, gamma, beta)
        return y

    @staticmethod
    def backward(ctx, grad_y):
        h, q, W, gamma, beta = ctx.saved_tensors
        idx = ctx.index
        Fh, Fo, N = h.shape[1], W.shape[0], h.shape[0]
        if ctx.cargs is not None:
            a, dev, E = ctx.cargs, h.device, idx.num_edges
            grad_y = grad_y.contiguous()
            grad_pre, gP = _f32((E, Fo), dev), _f32((N, 2 * Fo), dev)
            gg, gb, gbias = tape.grad_like(gamma), tape.grad_like(beta), tape.grad_for_bias_of(W, Fo)
            gW, gh = tape.grad_like(W), torch.empty_like(h)
            gq = torch.empty_like(q) if (ctx.has_q and ctx.needs_input_grad[1]) else None
            if ctx.has_q and ctx.qmap is not None:
                gQ = _f32((ctx.qmap.v_pad, Fo), dev)
                a.grad_Q = gQ.data_ptr()
            _set_workspaces(a.tail, Fo, dev)
            a.grad_y, a.grad_pre, a.grad_P = grad_y.data_ptr(), grad_pre.data_ptr(), gP.data_ptr()
            a.grad_gamma, a.grad_beta, a.grad_W, a.grad_bias = gg.data_ptr(), gb.data_ptr(), gW.data_ptr(), gbias.data_ptr()
            a.grad_h, a.grad_q = gh.data_ptr(), _ptr(gq)
            _call('i3d_edge_fc_bn_bwd', a)
            return gh, gq, gW, gbias, gg, gb, None, None, None
        grad_pre, gg, gb = _Tail.backward(ctx.saved, grad_y.contiguous(), gamma, beta, ctx.spec)
        gW = tape.grad_like(W)
        gP = torch.empty(N, 2 * Fo, dtype=torch.float32, device=h.device)
        ops.segment_sum(grad_pre, idx.out_ptr, idx.out_epos, N, out=gP[:, :Fo])     # d P[src]
        ops.segment_sum(grad_pre, idx.in_ptr, None, N, out=gP[:, Fo:])              # d P[dst]
        gbias = tape.grad_for_bias_of(W, Fo)
        qmap = ctx.qmap if ctx.has_q else None
        gQ = None
        if qmap is not None:      # dQ[v] = sum of dpre over the edges of category v
            gQ = ops.gemm(qmap.onehot, grad_pre, trans_a=True)[:qmap.rows]
        with fork(gP, h, grad_pre, q if ctx.has_q else None) as f:
            if ctx.has_q:
                ops.gemm(gQ if qmap is not None else grad_pre, q, trans_a=True, out=gW[:, 2 * Fh:])
            ops.gemm(gP[:, :Fo], h, trans_a=True, out=gW[:, :Fh])
            ops.gemm(gP[:, Fo:], h, trans_a=True, out=gW[:, Fh:2 * Fh])
            ops.colsum(grad_pre, out=gbias)
        gq = None
        if ctx.has_q and ctx.needs_input_grad[1]:
            gq = ops.gemm(gQ if qmap is not None else grad_pre, W[:, 2 * Fh:])
        gh = ops.gemm(gP[:, :Fo], W[:, :Fh])
        ops.gemm(gP[:, Fo:], W[:, Fh:2 * Fh], out=gh, accumulate=True)
        f.join()
        return gh, gq, gW, gbias, gg, gb, None, None, None


# ------------------------------------------------------------------------------------------------------------
class FCLayer(nn.Module):
    """Drop-in for reference models/base_layers.py:23-111."""

    def __init__(self, in_dim, out_dim, activation='relu', dropout=0., batch_norm=False, batch_norm_momentum=0.1,
                 bias=True, init_fn=None, device='cpu'):
        super().__init__()
        self.in_dim, self.out_dim, self.bias = in_dim, out_dim, bias
        self.linear = nn.Linear(in_dim, out_dim, bias=bias).to(device)
        if not bias:
            # reference models/base_layers.py:86: nn.Linear(..., bias=False) - no bias parameter, no state_dict key.  The kernels take
            # a bias vector: a zero buffer outside the state_dict (it follows .to(); its gradient is computed and dropped).  The MLPs of
            # the models never build such a layer (base_layers.py:114-147 does not pass `bias`), so the whole-model sequencers never
            # meet one (pna_native / net3d_native check).
            self.register_buffer('_zero_bias', torch.zeros(out_dim, device=device), persistent=False)
        self.dropout = nn.Dropout(p=dropout) if dropout else None      # (the reference's attribute; applied in _Tail, training mode only)
        self.batch_norm = nn.BatchNorm1d(out_dim, momentum=batch_norm_momentum).to(device) if batch_norm else None
        self.activation = act_name(activation)
        self.init_fn = nn.init.xavier_uniform_
        self.sync_group = None
        self.reset_parameters()

    def reset_parameters(self, init_fn=None):
        init_fn = init_fn or self.init_fn
        if init_fn is not None:
            init_fn(self.linear.weight, 1 / self.in_dim)      # reference :93-98: gain = 1/in_dim
        if self.bias:
            self.linear.bias.data.zero_()

    # Every attribute below goes through nn.Module.__getattr__ (parameters, buffers and sub-modules live in dicts): ~14
    # slow lookups and two dataclass constructions per call, ~18 calls per training step.  `hot(post_act)` caches
    # (weight, bias, gamma, beta, spec); anything that can change what it holds drops the cache: train()/eval(),
    # .to()/.cuda()/.float() (`_apply` replaces the buffer objects), a new sync group.  load_state_dict copies in place.
    def spec(self, post_act=None) -> FCSpec:
        return self.hot(post_act)[4]

    def bn_affine(self):
        h = self.hot(None)
        return h[2], h[3]

    def hot(self, post_act=None):
        cache = self.__dict__.get('_i3d_hot')
        if cache is None:
            cache = self.__dict__['_i3d_hot'] = {}
        h = cache.get(post_act)
        if h is not None:
            lp, bp = h[5], h[6]      # the sub-modules' parameter dicts: a re-assigned Parameter object invalidates the entry
            bias_now = lp['bias'] if self.bias else self._buffers['_zero_bias']
            if lp['weight'] is h[0] and bias_now is h[1] and (bp is None or (bp['weight'] is h[2] and bp['bias'] is h[3])):
                dm = h[7]            # the nn.Dropout module (or None): `layer.dropout.p = ...` in place must take effect too
                if dm is None or (float(dm.p) if self.training else 0.0) == h[4].dropout:
                    return h
        bn, gamma, beta, bp = None, None, None, None
        if self.batch_norm is not None:
            m = self.batch_norm
            bn = BNSpec(m.running_mean, m.running_var, m.num_batches_tracked, m.momentum, m.eps, self.training,
                        self.sync_group if self.training else None)
            gamma, beta, bp = m.weight, m.bias, m._parameters
        lin = self.linear
        drop = float(self.dropout.p) if (self.dropout is not None and self.training) else 0.0
        h = cache[post_act] = (lin.weight, lin.bias if self.bias else self._buffers['_zero_bias'], gamma, beta,
                               FCSpec(self.activation, bn, post_act, drop), lin._parameters, bp, self.dropout)
        return h

    def _drop_hot(self):
        self.__dict__.pop('_i3d_hot', None)

    def train(self, mode: bool = True):
        self._drop_hot()
        return super().train(mode)

    def _apply(self, fn, *args, **kwargs):
        self._drop_hot()
        return super()._apply(fn, *args, **kwargs)

    def __setattr__(self, name, value):
        if name in ('sync_group', 'batch_norm', 'linear', 'activation', 'dropout'):
            self.__dict__.pop('_i3d_hot', None)
        super().__setattr__(name, value)

    def forward(self, x, residual=None, post_act=None):
        W, b, gamma, beta, spec = self.hot(post_act)[:5]
        return tape.apply(FCFn, x, W, b, gamma, beta, residual, spec)


class MLP(nn.Module):
    """Drop-in for reference models/base_layers.py:114-147."""

    def __init__(self, in_dim, out_dim, layers, hidden_size=None, mid_activation='relu', last_activation='none',
                 dropout=0., mid_batch_norm=False, last_batch_norm=False, batch_norm_momentum=0.1, device='cpu'):
        super().__init__()
        self.in_dim, self.hidden_size, self.out_dim = in_dim, hidden_size, out_dim
        self.fully_connected = nn.ModuleList()
        if layers <= 1:
            self.fully_connected.append(FCLayer(in_dim, out_dim, activation=last_activation, batch_norm=last_batch_norm,
                                                device=device, dropout=dropout,
                                                batch_norm_momentum=batch_norm_momentum))
        else:
            self.fully_connected.append(FCLayer(in_dim, hidden_size, activation=mid_activation,
                                                batch_norm=mid_batch_norm, device=device, dropout=dropout,
                                                batch_norm_momentum=batch_norm_momentum))
            for _ in range(layers - 2):
                self.fully_connected.append(FCLayer(hidden_size, hidden_size, activation=mid_activation,
                                                    batch_norm=mid_batch_norm, device=device, dropout=dropout,
                                                    batch_norm_momentum=batch_norm_momentum))
            self.fully_connected.append(FCLayer(hidden_size, out_dim, activation=last_activation,
                                                batch_norm=last_batch_norm, device=device, dropout=dropout,
                                                batch_norm_momentum=batch_norm_momentum))

    def forward(self, x, residual=None, post_act=None):
        n = len(self.fully_connected)
        for i, fc in enumerate(self.fully_connected):
            last = i == n - 1
            x = fc(x, residual if last else None, post_act if last else None)
        return x

    # first layer fed by a fused input operator (edge gather / two-segment concat), rest plain
    def forward_edge(self, h, q, index, residual=None, qmap=None):
        fc0 = self.fully_connected[0]
        W, b, gamma, beta, spec = fc0.hot()[:5]
        x = tape.apply(EdgeFCFn, h, q, W, b, gamma, beta, index, spec, qmap)
        for fc in list(self.fully_connected)[1:]:
            x = fc(x)
        return x

    def forward_concat2_grouped(self, h, a, index, coef, residual=None):
        """[h | scaler blocks of a] -> MLP with the degree-combined weights (GroupedConcat2FCFn) for the first layer."""
        fcs = list(self.fully_connected)
        fc0 = fcs[0]
        gamma, beta = fc0.bn_affine()
        x = tape.apply(GroupedConcat2FCFn, h, a, fc0.linear.weight, fc0.hot()[1], gamma, beta,      # hot()[1]: the bias, or the zero buffer of bias=False
                                     residual if len(fcs) == 1 else None, index, coef, fc0.spec())
        for i, fc in enumerate(fcs[1:]):
            x = fc(x, residual if i == len(fcs) - 2 else None)
        return x

    def forward_concat2(self, a, c, residual=None):
        fcs = list(self.fully_connected)
        fc0 = fcs[0]
        gamma, beta = fc0.bn_affine()
        x = tape.apply(Concat2FCFn, a, c, fc0.linear.weight, fc0.hot()[1], gamma, beta,
                              residual if len(fcs) == 1 else None, fc0.spec())
        for i, fc in enumerate(fcs[1:]):
            x = fc(x, residual if i == len(fcs) - 2 else None)
        return x


# ------------------------------------------------------------------------------------------------------------
class EmbeddingSumFn(torch.autograd.Function):
    """sum_k Emb_k[idx[:,k]]  (reference commons/mol_encoder.py:34-42, 65-73)."""

    @staticmethod
    def forward(ctx, idx, row_perm, *tables):
        ctx.idx, ctx.row_perm = idx, row_perm
        ctx.dims = [t.shape[0] for t in tables]
        return ops.embedding_sum_fwd(idx, [t.contiguous() for t in tables], row_perm)

    @staticmethod
    def backward(ctx, grad_out):
        grads = ops.embedding_sum_bwd(ctx.idx, grad_out.contiguous(), ctx.dims, ctx.row_perm)
        return (None, None, *grads)


class EmbeddingSumPadFn(torch.autograd.Function):
    """the `padding=True` form of the encoders (reference commons/mol_encoder.py:22-23, 36-37): tables with one extra row 0,
    looked up at idx + 1 (-1 -> row 0); row 0 takes part in the sum but, being `padding_idx`, receives no gradient"""

    @staticmethod
    def forward(ctx, idx, row_perm, *tables):
        ctx.idx, ctx.row_perm = idx + 1, row_perm
        ctx.dims = [t.shape[0] for t in tables]
        return ops.embedding_sum_fwd(ctx.idx, [t.contiguous() for t in tables], row_perm)

    @staticmethod
    def backward(ctx, grad_out):
        grads = ops.embedding_sum_bwd(ctx.idx, grad_out.contiguous(), ctx.dims, ctx.row_perm)
        for g in grads:
            g[0].zero_()
        return (None, None, *grads)


class AggregateFn(torch.autograd.Function):
    """K4: segmented mean/max/min/std x degree scalers (reference models/pna.py:206, 221-235)."""

    @staticmethod
    def forward(ctx, e, index, aggregators, scalers, avg_d_log, force_scalers=False, tower_feat=0):
        e = e.contiguous()
        ctx.cfg = (index, aggregators, scalers, avg_d_log, force_scalers, tower_feat)
        ctx.save_for_backward(e)
        return ops.pna_aggregate_fwd(e, index.in_ptr, index.num_nodes, aggregators, scalers, avg_d_log, force_scalers, tower_feat)

    @staticmethod
    def backward(ctx, grad_out):
        (e,) = ctx.saved_tensors
        index, aggregators, scalers, avg, force, tower_feat = ctx.cfg
        return ops.pna_aggregate_bwd(grad_out.contiguous(), e, index.in_ptr, index.num_nodes, aggregators, scalers,
                                     avg, force, tower_feat), None, None, None, None, None, None


class ReadoutFn(torch.autograd.Function):
    """K6: per-graph min/max/mean/sum (reference models/pna.py:133-134)."""

    @staticmethod
    def forward(ctx, h, index, op_codes):
        h = h.contiguous()
        ctx.cfg = (index, op_codes)
        ctx.save_for_backward(h)
        return ops.segment_readout_fwd(h, index.graph_ptr, index.num_graphs, op_codes)

    @staticmethod
    def backward(ctx, grad_out):
        (h,) = ctx.saved_tensors
        index, op_codes = ctx.cfg
        return ops.segment_readout_bwd(grad_out.contiguous(), h, index.graph_ptr, index.num_graphs, op_codes), None, None


class DropoutFn(torch.autograd.Function):
    """y = nn.Dropout(p)(x) in training mode, as a step of the tape: the mask is torch's own (ops.dropout_mask), the products are
    HIP kernels.  reference models/pna_original.py:260 (PNATower), :428 (PNASimpleLayer), :188, :375 (in_feat_dropout)."""

    @staticmethod
    def forward(ctx, x, p):
        x = x.contiguous()
        m = ops.dropout_mask(x, p)
        ctx.save_for_backward(m)
        return ops.mul(x, m)

    @staticmethod
    def backward(ctx, g):
        (m,) = ctx.saved_tensors
        return ops.mul(g.contiguous(), m), None


def dropout(x, p, training):
    """nn.Dropout(p) on the tape; identity when p == 0 or in eval mode"""
    if not training or not p:
        return x
    return tape.apply(DropoutFn, x, float(p))
