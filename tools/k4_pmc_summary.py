"""HBM traffic of the PNA aggregation kernels (K4) from two rocprofv3 --pmc passes over tools/k4_bench.py.

    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/pmc_f -o f -- python tools/k4_bench.py --batches 512 8192 --reps 5 --bwd > k4.log
    rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/pmc_w -o w -- python tools/k4_bench.py --batches 512 8192 --reps 5 --bwd
    python tools/k4_pmc_summary.py <fetch.db> <write.db> k4.log profiles/rNN_k4_pmc

Units / corrections exactly as MI355X_MICROARCH.md (HBM section): FETCH_SIZE and WRITE_SIZE are KiB per dispatch;
on gfx950 FETCH_SIZE counts the 128-B requests of wide (16 B/lane) coalesced reads as 64 B -> doubled; WRITE_SIZE as is.
k4_bench launches, per batch size, the variants in the order fwd 12F, fwd 4F, bwd 12F, bwd 4F; the two backward
variants share one kernel symbol and grid, so they are told apart by dispatch order.
"""
import json
import re
import sqlite3
import sys


def per_dispatch(db, counter):
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute("select dispatch_id, kernel_name, grid_size, value from counters_collection "
                       "where counter_name = ? and kernel_name like '%pna_aggregate_%' order by dispatch_id", (counter,))
    return [(d, re.sub(r'\(.*', '', n).replace('void ', '').replace('i3d::', ''), g, v) for d, n, g, v in rows]


def main(fetch_db, write_db, log, out_prefix):
    algo = {}
    for line in open(log):
        m = re.match(r'K4 (fwd|bwd)\s+(\d+)F B=\s*(\d+) .*algorithmic\s+([\d.]+) MB', line)
        if m:
            algo[(m.group(1), int(m.group(2)), int(m.group(3)))] = float(m.group(4))
    batches = sorted({b for _, _, b in algo})
    res = {}
    lines = ['# rocprofv3 --kernel-trace --pmc <COUNTER> -- python tools/k4_bench.py --batches ... --reps 5 --bwd   (one pass per counter)',
             '# FETCH_SIZE / WRITE_SIZE: KiB per dispatch, averaged over the dispatches of a variant; gfx950 correction',
             '# (MI355X_MICROARCH.md, HBM): FETCH_SIZE x2 for wide coalesced reads, WRITE_SIZE as is.',
             f'{"variant":14s} {"kernel":34s} {"grid":>9s} {"counter":>11s} {"n":>3s} {"avg_KiB":>12s} {"corrected_MB":>13s}']
    for counter, db, corr in (('FETCH_SIZE', fetch_db, 2.0), ('WRITE_SIZE', write_db, 1.0)):
        rows = per_dispatch(db, counter)
        grids = sorted({g for _, _, g, _ in rows})
        assert len(grids) == len(batches), (grids, batches)
        for g, B in zip(grids, batches):
            fwd = [r for r in rows if r[2] == g and 'fwd' in r[1]]
            bwd = [r for r in rows if r[2] == g and 'bwd' in r[1]]
            groups = {('fwd', 12): [r for r in fwd if '<1>' in r[1]], ('fwd', 4): [r for r in fwd if '<2>' in r[1]],
                      ('bwd', 12): bwd[:len(bwd) // 2], ('bwd', 4): bwd[len(bwd) // 2:]}
            for (d, w), rs in groups.items():
                if not rs:
                    continue
                kib = sum(r[3] for r in rs) / len(rs)
                mb = kib * 1024 * corr / 1e6
                key = f'{d}{w}_B{B}'
                res.setdefault(key, {})[counter] = mb
                lines.append(f'{key:14s} {rs[0][1]:34s} {g:9d} {counter:>11s} {len(rs):3d} {kib:12.1f} {mb:13.2f}')
    lines += ['', '# per launch: corrected HBM traffic (fetch + write) vs algorithmic bytes (SURVEY.md 8d)']
    for key in sorted(res):
        d, w, B = re.match(r'(fwd|bwd)(\d+)_B(\d+)', key).groups()
        r = res[key]
        r['traffic_MB'] = r['FETCH_SIZE'] + r['WRITE_SIZE']
        r['algorithmic_MB'] = algo[(d, int(w), int(B))]
        lines.append(f'{key:14s} traffic {r["traffic_MB"]:9.1f} MB   algorithmic {r["algorithmic_MB"]:9.1f} MB   '
                     f'ratio {r["traffic_MB"] / r["algorithmic_MB"]:.3f}')
    text = '\n'.join(lines) + '\n'
    print(text)
    with open(out_prefix + '.txt', 'w') as f:
        f.write(text)
    with open(out_prefix + '.json', 'w') as f:
        json.dump(res, f, indent=1)


if __name__ == '__main__':
    main(*sys.argv[1:5])
