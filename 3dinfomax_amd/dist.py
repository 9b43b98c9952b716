"""Data-parallel wiring over RCCL (torch.distributed backend "nccl" on ROCm; "gloo" in the CPU tests).

The reference has no distributed code at all; BASELINE.json:north_star asks for molecule
sharding across the 8 GPUs of a node.  The path shards by molecule with three exchange points:
  C1  all-gather of the 3D-view embeddings for NT-Xent's negatives (losses._AllGatherRowsFn, backward =
      reduce-scatter),
  C2  gradient all-reduce (SUM: each rank's loss share already carries 1/B_global), one flat bucket per
      ~32 MB so a ring over xGMI moves few, large messages,
  C3  synchronised BatchNorm statistics (fp64 [sum, sumsq, count] all-reduce per BN, forward and backward -
      layers._Tail) because the reference normalises over the WHOLE batch of edges / nodes / graphs.
"""
import torch
import torch.distributed as dist

from .layers import FCLayer


def setup(modules, loss=None, group=None, sync_bn=True, broadcast=True):
    """Attach `group` to every FCLayer (sync-BN) of `modules` and to the loss; broadcast rank-0 weights."""
    group = group if group is not None else dist.group.WORLD
    for m in modules:
        for sub in m.modules():
            if isinstance(sub, FCLayer):
                sub.sync_group = group if sync_bn else None
        if broadcast:
            for t in list(m.parameters()) + list(m.buffers()):
                dist.broadcast(t.data, src=dist.get_global_rank(group, 0) if hasattr(dist, 'get_global_rank') else 0,
                               group=group)
    if loss is not None and hasattr(loss, 'attach_group'):
        loss.attach_group(group)
    return group


def allreduce_grads(params, group=None, bucket_bytes=32 << 20):
    """Sum the gradients over ranks in flat buckets (C2)."""
    group = group if group is not None else dist.group.WORLD
    grads = [p.grad for p in params if p.grad is not None]
    bucket, size = [], 0

    def flush():
        nonlocal bucket, size
        if not bucket:
            return
        flat = torch.cat([g.reshape(-1) for g in bucket])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
        off = 0
        for g in bucket:
            n = g.numel()
            g.copy_(flat[off:off + n].view_as(g))
            off += n
        bucket, size = [], 0

    for g in grads:
        bucket.append(g)
        size += g.numel() * g.element_size()
        if size >= bucket_bytes:
            flush()
    flush()


def global_loss(loss_share, group=None):
    """Sum of the per-rank loss shares = the reference's full-batch loss (for logging)."""
    out = loss_share.detach().clone()
    dist.all_reduce(out, op=dist.ReduceOp.SUM, group=group if group is not None else dist.group.WORLD)
    return out


def shard_molecules(mols, rank, world):
    """Contiguous, equally sized molecule ranges of the (already shuffled) global batch."""
    per = len(mols) // world
    return mols[rank * per:(rank + 1) * per]
