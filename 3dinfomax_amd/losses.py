"""NT-Xent losses on the MI355X kernels - drop-in for `NTXent` / `NTXentMultiplePositives` of reference
commons/losses.py:126-163, 206-258 (same constructor kwargs, `forward(z1, z2, **kwargs) -> 0-dim tensor`).

Data parallel (absent in the reference, required by BASELINE.json:north_star): when a process group is
attached (`loss.attach_group(group)` or 3dinfomax_amd.dist.setup), z2 - the 3D-view embeddings - is
all-gathered over RCCL so each local 2D row sees the FULL negative set; the backward of the gather is a
reduce-scatter of dz2.  The returned value is this rank's share  sum_local(l_i) / B_global; the shares sum to
the reference's loss (dist.global_loss all-reduces it for logging).
"""
import torch
from torch import Tensor
from torch.nn.modules.loss import _Loss

import os

from . import _lib, ops

# I3D_LOSS_COMPOSITE=0: the loss as ~5 C calls per direction sequenced from Python (same kernels, same bits)
LOSS_COMPOSITE = os.environ.get('I3D_LOSS_COMPOSITE', '1') != '0'


class _AllGatherRowsFn(torch.autograd.Function):
    """all_gather along dim 0; backward = reduce_scatter(sum) of the gathered gradient.  `counts` (rows per rank, known on
    every rank) allows shards of different sizes (dist.shard_plan does not drop the remainder of a batch)."""

    @staticmethod
    def forward(ctx, x, group, counts=None):
        from . import dist as adist
        ctx.group, ctx.counts = group, counts
        if counts is not None:
            return adist.all_gather_rows_var(x, counts, group)
        return adist.all_gather_rows(x, group)

    @staticmethod
    def backward(ctx, g):
        from . import dist as adist
        if ctx.counts is not None:
            return adist.reduce_scatter_rows_var(g, ctx.counts, ctx.group), None, None
        return adist.reduce_scatter_rows(g, ctx.group), None, None


class NTXentFn(torch.autograd.Function):
    """loss_share = sum_i -log(pos_i / (rowsum_i - pos_i)) / global_batch  over the local rows of z1."""

    @staticmethod
    def forward(ctx, z1, z2, tau, eps, conf, pos_offset, global_batch, norm=True):
        z1, z2 = z1.contiguous(), z2.contiguous()
        b1, b2 = z1.shape[0], z2.shape[0] // conf
        ctx.norm = norm
        if not norm:
            # reference commons/losses.py:147-150 / :236-239 skipped: the same kernels with unit norms and no epsilon;
            # the backward then has no norm term (the row_axpy corrections)
            n1 = torch.ones(z1.shape[0], dtype=torch.float32, device=z1.device)
            n2 = torch.ones(z2.shape[0], dtype=torch.float32, device=z2.device)
            sim = ops.gemm(z1, z2, trans_b=True)
            row_sum, row_pos, loss = ops.ntxent_fwd(sim, n1, n2, b1, b2, conf, pos_offset, tau, 0.0, 1.0 / global_batch)
            ctx.cfg = (tau, 0.0, conf, pos_offset, global_batch, b1, b2)
            ctx.composite = False
            ctx.save_for_backward(z1, z2, n1, n2, sim, row_sum, row_pos)
            return loss.reshape(())
        if LOSS_COMPOSITE and z1.is_cuda and z1.dtype == torch.float32 and z2.dtype == torch.float32:
            # row norms, similarity GEMM (MFMA) and the fused exp / row-sum / log kernel from ONE C call (csrc/ntxent.hip)
            L = _lib.load()
            scratch = torch.empty(L.i3d_ntxent_loss_scratch_floats(b1, b2 * conf), dtype=torch.float32, device=z1.device)
            loss = torch.empty(1, dtype=torch.float32, device=z1.device)
            _lib.check(L.i3d_ntxent_loss_fwd(z1.data_ptr(), z2.data_ptr(), b1, b2, conf, z1.shape[1], pos_offset, float(tau),
                                             float(eps), 1.0 / global_batch, scratch.data_ptr(), loss.data_ptr(), ops._stream()),
                       'i3d_ntxent_loss_fwd')
            ctx.cfg = (tau, eps, conf, pos_offset, global_batch, b1, b2)
            ctx.composite = True
            ctx.save_for_backward(z1, z2, scratch)
            return loss.reshape(())
        n1, n2 = ops.row_norms(z1), ops.row_norms(z2)
        sim = ops.gemm(z1, z2, trans_b=True)                      # [b1, b2*conf] on the MFMA GEMM
        row_sum, row_pos, loss = ops.ntxent_fwd(sim, n1, n2, b1, b2, conf, pos_offset, tau, eps, 1.0 / global_batch)
        ctx.cfg = (tau, eps, conf, pos_offset, global_batch, b1, b2)
        ctx.composite = False
        ctx.save_for_backward(z1, z2, n1, n2, sim, row_sum, row_pos)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, grad_out):
        tau, eps, conf, pos_offset, global_batch, b1, b2 = ctx.cfg
        if ctx.composite:
            z1, z2, scratch = ctx.saved_tensors
            L = _lib.load()
            b2c = b2 * conf
            work = torch.empty(b1 * b2c + b1 + b2c + 12, dtype=torch.float32, device=z1.device)
            dz1, dz2 = torch.empty_like(z1), torch.empty_like(z2)
            gs = grad_out.contiguous().float()
            _lib.check(L.i3d_ntxent_loss_bwd(z1.data_ptr(), z2.data_ptr(), b1, b2, conf, z1.shape[1], pos_offset, float(tau),
                                             float(eps), 1.0 / global_batch, scratch.data_ptr(), gs.data_ptr(), work.data_ptr(),
                                             dz1.data_ptr(), dz2.data_ptr(), ops._stream()), 'i3d_ntxent_loss_bwd')
            return dz1, dz2, None, None, None, None, None, None
        z1, z2, n1, n2, sim, row_sum, row_pos = ctx.saved_tensors
        # the upstream scalar gradient is multiplied in on the device (no host read-back, no extra elementwise op)
        dsim, ca, cb = ops.ntxent_bwd(sim, n1, n2, row_sum, row_pos, b1, b2, conf, pos_offset, tau, eps,
                                      1.0 / global_batch, grad_out.contiguous().float())
        dz1 = ops.gemm(dsim, z2)                                  # dS z2
        dz2 = ops.gemm(dsim, z1, trans_a=True)                    # dS^T z1
        if ctx.norm:
            ops.row_axpy(z1, ca, dz1)
            ops.row_axpy(z2, cb, dz2)
        return dz1, dz2, None, None, None, None, None, None


class _GramFn(torch.autograd.Function):
    """x x^T (rows=True: [n, n]) or x^T x ([d, d]) through the library's GEMM (csrc/gemm.hip) - the matrix products of the
    regularisers below stay on the hand-written kernels; d(G)/dx = (dG + dG^T) x  resp.  x (dG + dG^T)."""

    @staticmethod
    def forward(ctx, x, rows):
        x = x.contiguous()
        ctx.rows = rows
        ctx.save_for_backward(x)
        return ops.gemm(x, x, trans_b=True) if rows else ops.gemm(x, x, trans_a=True)

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        gs = (g + g.transpose(0, 1)).contiguous()
        return (ops.gemm(gs, x) if ctx.rows else ops.gemm(x, gs)), None


def _gram(x, rows):
    if x.is_cuda and x.dtype == torch.float32 and x.dim() == 2:
        return _GramFn.apply(x, rows)
    return x @ x.T if rows else x.T @ x          # (CPU tensors: the host-logic tests)


# Optional regularisers (weights 0 in every BASELINE config): differentiable torch expressions around the library's GEMM for
# their matrix products.  Semantics of reference commons/losses.py:946-964.
def _log_mean_gaussian_potential(x, t):
    """log of the mean of exp(-t |x_i - x_j|^2) over all unordered pairs i < j"""
    n = x.shape[0]
    sq = (x * x).sum(dim=1)
    d2 = (sq[:, None] + sq[None, :] - 2.0 * _gram(x, True)).clamp_min(0.0)
    iu = torch.triu_indices(n, n, offset=1, device=x.device)
    return torch.exp(-t * d2[iu[0], iu[1]]).mean().log()


def uniformity_loss(x1: Tensor, x2: Tensor, t=2) -> Tensor:
    """mean of the two views' uniformity terms (Wang & Isola), reference :946-951"""
    return 0.5 * (_log_mean_gaussian_potential(x1, t) + _log_mean_gaussian_potential(x2, t))


def cov_loss(x):
    """sum of squared off-diagonal entries of the feature covariance, divided by the feature count (reference :954-959)"""
    n, dim = x.shape
    centred = x - x.mean(dim=0, keepdim=True)
    cov = _gram(centred, False) / (n - 1)
    off = cov - torch.diag_embed(torch.diagonal(cov))
    return (off * off).sum() / dim


def std_loss(x):
    """hinge on the per-feature standard deviation: mean(relu(1 - sqrt(var + 1e-4))) (reference :962-964)"""
    return torch.relu(1.0 - (x.var(dim=0) + 1e-4).sqrt()).mean()


class _NTXentBase(_Loss):
    _eps = 1e-8

    def __init__(self, norm: bool = True, tau: float = 0.5, uniformity_reg=0, variance_reg=0, covariance_reg=0):
        super().__init__()
        self.norm, self.tau = norm, tau
        self.uniformity_reg, self.variance_reg, self.covariance_reg = uniformity_reg, variance_reg, covariance_reg
        self.group = None
        self.shard_counts = None
        self._equal_checked = {}       # local row count -> True once this rank has READ a check of that count (see _check_equal_shards)
        self._pending = None           # (result of the last equal-shard all-reduce, event of its pinned copy, local rows)
        self._pinned = None

    def attach_group(self, group):
        """Enable the data-parallel form (all-gathered negatives) on a torch.distributed process group."""
        self.group = group
        return self

    def set_shard_counts(self, counts):
        """molecules per rank of the current global batch when they differ (dist.shard_counts / shard_plan); None: equal"""
        self.shard_counts = list(counts) if counts is not None else None
        return self

    def _check_equal_shards(self, z1, dist):
        """Equal shards are ASSUMED when no counts were given (all_gather_into_tensor / reduce_scatter_tensor with different row
        counts per rank hang or corrupt silently) - e.g. the last partial batch of an epoch sharded with the remainder kept.
        EVERY call issues the same 2-element MAX all-reduce on every rank (whether a collective runs must never depend on
        rank-local state: a rank that has seen this row count before and one that has not would otherwise issue different
        collectives).  What IS rank-local is only when the result is read: at once for a row count this rank has not verified
        yet and on host-side backends (gloo: the result is already there), otherwise - the steady state, no host
        synchronisation in the step - through a pinned copy examined at the start of the next call.  A rank whose count is new
        raises before its gather; a rank whose count it knew raises one call later."""
        self._raise_if_unequal(block=True)          # the previous call's result: long finished, costs no wait
        rows = z1.shape[0]
        host_side = dist.get_backend(self.group) == 'gloo'
        n = torch.tensor([rows, -rows], dtype=torch.float64, device='cpu' if host_side else z1.device)
        dist.all_reduce(n, op=dist.ReduceOp.MAX, group=self.group)
        if host_side:
            self._pending = (n, None, rows)
        else:
            if self._pinned is None:
                self._pinned = torch.empty(2, dtype=torch.float64).pin_memory()
            self._pinned.copy_(n, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            self._pending = (self._pinned, ev, rows)
        if host_side or rows not in self._equal_checked:
            self._raise_if_unequal(block=True)
            self._equal_checked[rows] = True

    def _raise_if_unequal(self, block):
        if self._pending is None:
            return
        n, ev, rows = self._pending
        if ev is not None:
            if not block and not ev.query():
                return
            ev.synchronize()
        self._pending = None
        if float(n[0]) != -float(n[1]):
            raise ValueError(f'ranks hold different numbers of molecules ({rows} here, {int(-float(n[1]))}..'
                             f'{int(float(n[0]))} over the group): pass them with loss.set_shard_counts(dist.shard_counts(...))')

    def _contrastive(self, z1, z2, conf):
        pos_offset, global_batch = 0, z1.shape[0]
        if self.group is not None:
            import torch.distributed as dist
            world, rank = dist.get_world_size(self.group), dist.get_rank(self.group)
            if world > 1:
                counts = self.shard_counts
                if counts is not None and (len(counts) != world or counts[rank] != z1.shape[0]):
                    raise ValueError(f'shard counts {counts} do not describe this batch ({z1.shape[0]} rows on rank {rank})')
                if counts is not None and max(counts) != min(counts):
                    z2 = _AllGatherRowsFn.apply(z2, self.group, [c * conf for c in counts])
                    pos_offset, global_batch = sum(counts[:rank]), sum(counts)
                else:
                    if counts is None:
                        self._check_equal_shards(z1, dist)
                    z2 = _AllGatherRowsFn.apply(z2, self.group)
                    pos_offset, global_batch = rank * z1.shape[0], world * z1.shape[0]
        return NTXentFn.apply(z1, z2, float(self.tau), float(self._eps), conf, pos_offset, global_batch, bool(self.norm))

    def _regularisers(self, loss, z1, z2):
        if self.variance_reg > 0:
            loss = loss + self.variance_reg * (std_loss(z1) + std_loss(z2))
        if self.covariance_reg > 0:
            loss = loss + self.covariance_reg * (cov_loss(z1) + cov_loss(z2))
        if self.uniformity_reg > 0:
            loss = loss + self.uniformity_reg * uniformity_loss(z1, z2)
        return loss


class NTXent(_NTXentBase):
    """Normalized Temperature-scaled Cross Entropy Loss (reference commons/losses.py:126-163)."""
    _eps = 1e-8        # reference :150

    def forward(self, z1, z2, **kwargs) -> Tensor:
        return self._regularisers(self._contrastive(z1, z2, 1), z1, z2)


class NTXentMultiplePositives(_NTXentBase):
    """reference commons/losses.py:206-258; z2 is [batch*num_conformers, dim], conformer-minor; no epsilon (:239)."""
    _eps = 0.0

    def __init__(self, norm: bool = True, tau: float = 0.5, uniformity_reg=0, variance_reg=0, covariance_reg=0,
                 conformer_variance_reg=0) -> None:
        super().__init__(norm, tau, uniformity_reg, variance_reg, covariance_reg)
        self.conformer_variance_reg = conformer_variance_reg

    def forward(self, z1, z2, **kwargs) -> Tensor:
        batch_size, metric_dim = z1.size()
        conf = z2.shape[0] // batch_size
        loss = self._contrastive(z1, z2, conf)
        z2v = z2.view(batch_size, -1, metric_dim)
        if self.variance_reg > 0:
            loss = loss + self.variance_reg * (std_loss(z1) + std_loss(z2v))
        if self.conformer_variance_reg > 0:       # the same hinge over the conformer axis
            loss = loss + self.conformer_variance_reg * torch.relu(1.0 - (z2v.var(dim=1) + 1e-4).sqrt()).mean()
        if self.covariance_reg > 0:
            loss = loss + self.covariance_reg * (cov_loss(z1) + cov_loss(z2v))
        if self.uniformity_reg > 0:
            loss = loss + self.uniformity_reg * uniformity_loss(z1, z2v)
        return loss
