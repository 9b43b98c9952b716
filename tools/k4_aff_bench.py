"""K4 forward / backward with and without the messages normalised on load (the form the step launches), back to back, batch 512 and 8192.
    python tools/k4_aff_bench.py"""
import importlib, os, sys, torch
sys.path.insert(0, os.getcwd())
amd = importlib.import_module('3dinfomax_amd'); ops = importlib.import_module('3dinfomax_amd.ops')
dev = torch.device('cuda:0')
aggs, ident = ops.agg_codes(['mean', 'max', 'min', 'std']), ops.scaler_codes(['identity'])
for B in (512, 8192):
    mols = amd.synth.make_dataset(B, seed=1000)
    idx = amd.batch([amd.bond_graph(m) for m in mols]).index().to(dev)
    N, E, F = idx.num_nodes, idx.num_edges, 200
    e = torch.randn(E, F, device=dev); gout4 = torch.randn(N, 4 * F, device=dev)
    aff = torch.stack([torch.zeros(F), torch.ones(F), torch.zeros(F)]).to(dev).contiguous()
    byts = 4.0 * N * 4 * F + 2 * 4.0 * E * F + 4.0 * (N + 1)
    for name, fn in (('bwd 4F aff', lambda: ops.pna_aggregate_bwd_aff(gout4, e, aff, idx.in_ptr, N, aggs, ident, 1.0)),
                     ('bwd 4F    ', lambda: ops.pna_aggregate_bwd(gout4, e, idx.in_ptr, N, aggs, ident)),
                     ('fwd 4F aff', lambda: ops.pna_aggregate_fwd_aff(e, aff, idx.in_ptr, N, aggs, ident, 1.0)),
                     ('fwd 4F    ', lambda: ops.pna_aggregate_fwd(e, idx.in_ptr, N, aggs, ident))):
        for _ in range(3): fn()
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); t0.record()
        for _ in range(100): fn()
        t1.record(); torch.cuda.synchronize()
        us = t0.elapsed_time(t1) * 10
        print(f'B={B} {name} {us:7.2f} us' + (f'  {byts / us * 1e-3 / 8000:.3f} of 8 TB/s' if name.startswith('bwd') else ''))
