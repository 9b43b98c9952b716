"""The BatchNorm-backward family and K4 at the headline step's shapes (batch 512, hidden 200), a few launches each: the
target of the rocprofv3 --pmc passes of tools/bn_pmc.sh; `--summarise <db> ...` prints every collected counter per kernel.

    python tools/bn_pmc.py [--reps 5]
    python tools/bn_pmc.py --summarise a_results.db b_results.db ... [--trace trace_results.db]
"""
import importlib
import os
import re
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PATTERN = r'colreduce_partial|bn_bwd|bn_finalize|pna_aggregate|bn_apply|pair_final'


def run(reps):
    import torch
    amd = importlib.import_module('3dinfomax_amd')
    ops = importlib.import_module('3dinfomax_amd.ops')
    dev = torch.device('cuda:0')
    F = 200
    g2 = amd.batch([amd.bond_graph(m) for m in amd.synth.make_dataset(512, seed=1000)]).to(dev)
    idx = g2.index()
    N, E = idx.num_nodes, idx.num_edges
    rnd = lambda *s: torch.randn(*s, device=dev)
    gamma, beta, bias = torch.ones(F, device=dev), torch.zeros(F, device=dev), rnd(F)
    P, Q = rnd(N, 2 * F), rnd(60, F)
    code = torch.randint(0, 60, (E,), device=dev, dtype=torch.int32)
    xact, partial, tiles = ops.edge_combine_act_stats(P, Q, bias, idx.src_s, idx.dst_s, 'relu', code)
    mean, invstd, _ = ops.bn_finalize_partials(partial, tiles, F, 1e-5, 0.1, gamma, beta)
    dY_e, dY_n, lin_n = rnd(E, F) * 0.1, rnd(N, F) * 0.1, rnd(N, F)
    gb = torch.empty(F, device=dev)
    aff = torch.stack([xact.mean(0), torch.ones(F, device=dev), torch.zeros(F, device=dev)])
    aggs, ident = ops.agg_codes(['mean', 'max', 'min', 'std']), ops.scaler_codes(['identity'])
    gout = rnd(N, 4 * F)
    for _ in range(reps):
        # [E,F] block without an activation (FC2 of the pretrans MLP): reduction + data gradient, exact-zero bias gradient
        ops.bn_bwd(dY_e, xact, None, 'none', None, mean, invstd, gamma, beta, grad_bias=gb)
        # [N,F] block (posttrans)
        ops.bn_bwd(dY_n, lin_n, None, 'none', None, mean, invstd, gamma, beta, grad_bias=gb)
        # ReLU block (the reduction + data gradient + bias column sums form)
        ops.bn_bwd(dY_e, xact, None, 'relu', None, mean, invstd, gamma, beta, grad_bias=gb)
        ops.bn_finalize_partials(partial, tiles, F, 1e-5, 0.1, gamma, beta)
        ops.pna_aggregate_fwd_aff(xact, aff, idx.in_ptr, N, aggs, ident, 1.0)
        ops.pna_aggregate_bwd_aff(gout, xact, aff, idx.in_ptr, N, aggs, ident, 1.0)
    torch.cuda.synchronize()
    print(f'N {N} E {E} F {F}: [E,F] tensor {4e-6 * E * F:.2f} MB, [N,F] {4e-6 * N * F:.2f} MB, [N,4F] {16e-6 * N * F:.2f} MB')


def short(name):
    s = re.sub(r'\(.*', '', name.replace('(anonymous namespace)::', '')).replace('void ', '').replace('i3d::', '')
    return s[:60]


def summarise(dbs, trace=None):
    rows = {}
    for db in dbs:
        cur = sqlite3.connect(db).cursor()
        q = ('select kernel_name, grid_size, counter_name, avg(value), count(*) from counters_collection '
             'group by kernel_name, grid_size, counter_name')
        for name, grid, cname, val, n in cur.execute(q):
            if not re.search(PATTERN, name):
                continue
            rows.setdefault((short(name), grid), {})[cname] = val
    dur = {}
    if trace:
        cur = sqlite3.connect(trace).cursor()
        try:       # the `kernels` view of the rocpd database: name, start, end (ns), grid_x/y/z
            cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
            gx = [c for c in cols if c.lower() in ('grid_x', 'grid_size_x', 'grid_size')]
            grid = ' * '.join(c for c in cols if c.lower() in ('grid_x', 'grid_y', 'grid_z', 'grid_size_x', 'grid_size_y', 'grid_size_z')) or (gx[0] if gx else '0')
            for name, g, ns, n in cur.execute(f'select name, {grid}, avg(end - start), count(*) from kernels group by 1, 2'):
                dur[(short(name), g)] = (ns / 1e3, n)
        except sqlite3.Error as e:
            print('# trace db not readable:', e)
    for key in sorted(rows):
        d = rows[key]
        print(f'{key[0]}   grid {key[1]}' + (f'   avg {dur[key][0]:.2f} us over {dur[key][1]} dispatches (kernel trace, no counters)' if key in dur else ''))
        for c in sorted(d):
            extra = ''
            if c == 'FETCH_SIZE':
                extra = f'   = {2 * d[c] * 1024 / 1e6:.2f} MB (x2: gfx950 wide-read correction)'
            if c == 'WRITE_SIZE':
                extra = f'   = {d[c] * 1024 / 1e6:.2f} MB'
            print(f'    {c:28s} {d[c]:16.1f}{extra}')
        wc = d.get('SQ_WAVE_CYCLES')
        if wc:
            for c in ('SQ_WAIT_INST_ANY', 'SQ_WAIT_ANY', 'SQ_ACTIVE_INST_VMEM', 'SQ_ACTIVE_INST_VALU', 'SQ_ACTIVE_INST_LDS', 'SQ_ACTIVE_INST_ANY'):
                if c in d:
                    print(f'    {c + " / SQ_WAVE_CYCLES":40s} {d[c] / wc:8.3f}')
        if d.get('SQ_WAVES') and wc:
            print(f'    {"wave cycles per wave":40s} {wc / d["SQ_WAVES"]:10.0f}')
        if d.get('SQ_BUSY_CYCLES') and d.get('GRBM_GUI_ACTIVE'):
            print(f'    {"SQ_BUSY_CYCLES / GRBM_GUI_ACTIVE":40s} {d["SQ_BUSY_CYCLES"] / d["GRBM_GUI_ACTIVE"]:8.3f}')


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == '--summarise':
        args = sys.argv[2:]
        trace = None
        if '--trace' in args:
            i = args.index('--trace')
            trace = args[i + 1]
            args = args[:i] + args[i + 2:]
        summarise(args, trace)
    else:
        reps = int(sys.argv[sys.argv.index('--reps') + 1]) if '--reps' in sys.argv else 5
        run(reps)
