"""The fine-tune step of BASELINE.json configs[4] (reference configs_clean/tune_QM9_homo.yml:47-75: PNA only, batch 1024, depth 7,
readout min / max / mean / sum, L1 loss on one target, Adam) as a loop of its own - for rocprofv3 traces and A/B runs:
    python tools/finetune_step.py [--steps 40] [--warmup 10] [--batch 1024]"""
import argparse
import importlib
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=40)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--batch', type=int, default=1024)
    a = ap.parse_args()
    amd = importlib.import_module('3dinfomax_amd')
    dev = torch.device('cuda:0')
    mols = amd.synth.make_dataset(a.batch, seed=4000)
    g2 = amd.batch([amd.bond_graph(m) for m in mols]).to(dev)
    targets = torch.randn(a.batch, 1, device=dev)
    torch.manual_seed(123)
    kw = dict(bench.PNA_KW, target_dim=1, batch_norm_momentum=0.1, propagation_depth=7, readout_aggregators=['min', 'max', 'mean', 'sum'])
    pna = amd.PNA(avg_d=1.0, device=dev, **kw).to(dev).train()
    optim = amd.Adam(list(pna.parameters()), lr=7e-5, weight_decay=1e-11, fused=True)
    l1 = torch.nn.L1Loss()

    def step():
        l1(pna(g2.local_copy()), targets).backward()
        optim.step()
        optim.zero_grad()
    for _ in range(a.warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    host = time.perf_counter() - t0
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(json.dumps(dict(workload='configs[4] fine-tune: PNA only, depth 7, L1 loss', batch=a.batch, atoms=int(g2.number_of_nodes()),
                          ms_per_step=round(dt / a.steps * 1e3, 3), host_enqueue_ms_per_step=round(host / a.steps * 1e3, 3),
                          molecules_per_s=round(a.batch * a.steps / dt, 1))))


if __name__ == '__main__':
    main()
