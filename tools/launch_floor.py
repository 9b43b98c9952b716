"""Host cost of the kernel launches alone: the layer composites (one C call = ~12 forward / ~33 backward launches) replayed
back to back with the argument structs of a real step, no Python in between.  Tells what a whole-step C sequence could reach.
    python tools/launch_floor.py"""
import ctypes
import importlib
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    amd = importlib.import_module('3dinfomax_amd')
    ln = importlib.import_module('3dinfomax_amd.layer_native')
    _lib = importlib.import_module('3dinfomax_amd._lib')
    ops = importlib.import_module('3dinfomax_amd.ops')
    dev = torch.device('cuda:0')
    mols = amd.synth.make_dataset(512, seed=1000)
    g2 = amd.batch([amd.bond_graph(m) for m in mols]).to(dev)
    torch.manual_seed(123)
    pna = amd.PNA(avg_d=1.0, device=dev, **bench.PNA_KW).to(dev).train()
    ln.KEEP_LAST_ARGS = []
    out = pna(g2.local_copy())
    out.sum().backward()
    torch.cuda.synchronize()
    args = [a for a, _ in ln.KEEP_LAST_ARGS]
    L = _lib.load()
    st = ops._stream()
    for name, fn, launches in (('layer fwd', L.i3d_pna_layer_fwd, None), ('layer bwd', L.i3d_pna_layer_bwd, None)):
        for reps in (1, 50):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                for a in args:
                    rc = fn(ctypes.byref(a), st)
                    assert rc == 0
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            n = reps * len(args)
            print(f'{name}: {n} calls, host {1e6 * (t1 - t0) / n:7.1f} us per call, incl. GPU drain {1e6 * (t2 - t0) / n:7.1f} us per call')


if __name__ == '__main__':
    main()
