"""Does RCCL (torch.distributed backend "nccl" on ROCm) accept TWO ranks on ONE device?  (what one GPU can still
prove about the N > 1 path.)  Two processes, both on cuda:0: communicator creation + one all-reduce + one all-gather, then the
library's own RCCL provider of synchronised BatchNorm (dist.enable_native_sync(provider='rccl')).  Prints one JSON line per rank.

    python tools/rccl_one_gpu_probe.py
"""
import json
import os
import sys
import traceback

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def worker(rank, world, port):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0', NCCL_DEBUG='WARN')
    out = dict(rank=rank, world=world, device='cuda:0')
    try:
        torch.cuda.set_device(0)
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda:0'))
        t = torch.full((1024,), float(rank + 1), device='cuda:0')
        dist.all_reduce(t)
        torch.cuda.synchronize()
        out['all_reduce'] = float(t[0].item())
        out['all_reduce_ok'] = out['all_reduce'] == world * (world + 1) / 2
        sys.path.insert(0, ROOT)
        import importlib
        adist = importlib.import_module('3dinfomax_amd.dist')
        ok = adist.enable_native_sync(dist.group.WORLD, torch.device('cuda:0'), provider='rccl')
        out['native_sync_rccl_provider'] = bool(ok)
    except Exception as e:      # noqa: BLE001
        out['error'] = f'{type(e).__name__}: {str(e)[:400]}'
        out['trace_tail'] = traceback.format_exc()[-600:]
    print(json.dumps(out), flush=True)
    try:
        dist.destroy_process_group()
    except Exception:      # noqa: BLE001
        pass


if __name__ == '__main__':
    import socket
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    mp.start_processes(worker, args=(2, port), nprocs=2, join=True, start_method='spawn')
