"""Host and GPU cost of the data-parallel exchange points at world size 1 over RCCL (the per-call overheads that do not
depend on the number of ranks): C1 all-gather / reduce-scatter of [512, 256], C2 all-reduce of the flat gradient buffer,
and the multi-tensor copy of the gradients into it.   python tools/dp_cost.py
"""
import importlib
import os
import sys
import time

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def timed(fn, reps=200):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    host = (time.perf_counter() - t0) / reps * 1e6
    torch.cuda.synchronize()
    return host, e0.elapsed_time(e1) * 1e3 / reps


if __name__ == '__main__':
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29533')
    dev = torch.device('cuda:0')
    torch.cuda.set_device(dev)
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
    adist = importlib.import_module('3dinfomax_amd.dist')
    z = torch.randn(512, 256, device=dev)
    flat = torch.randn(5_000_000, device=dev)
    sizes = [200 * 200] * 60 + [200] * 50
    srcs = [torch.randn(s, device=dev) for s in sizes]
    dsts = list(flat[:sum(sizes)].split(sizes))
    for name, fn in (('all_gather_rows [512,256]', lambda: adist.all_gather_rows(z)),
                     ('reduce_scatter_rows [512,256]', lambda: adist.reduce_scatter_rows(z)),
                     ('all_reduce 20 MB', lambda: adist.all_reduce_sum(flat)),
                     ('all_reduce 4 B', lambda: adist.all_reduce_sum(z[0, :1])),
                     ('foreach_copy 110 tensors', lambda: torch._foreach_copy_(dsts, srcs)),
                     ('elementwise add [512,256] (a plain torch op for scale)', lambda: z + z)):
        h, g = timed(fn)
        print(f'{name:58s} host {h:7.1f} us   stream {g:7.1f} us per call', flush=True)
    dist.destroy_process_group()
