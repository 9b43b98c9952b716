"""HBM traffic per dispatch of the kernels matching a pattern, from two rocprofv3 --pmc passes (one counter per pass:
FETCH_SIZE costs 3 TCC slots, WRITE_SIZE 2 - MI355X_MICROARCH.md, PMC slots):

    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/pmc_f -o f -- python tools/family_bench.py
    rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/pmc_w -o w -- python tools/family_bench.py
    python tools/pmc_summary.py <fetch.db> <write.db> 'bn_|colreduce|edge_combine_act|gemm_f32_fused|pna_aggregate' > profiles/rNN_bn_pmc.txt

Units / corrections as in the guide's HBM section: both counters are KiB per dispatch; on gfx950 FETCH_SIZE counts the
128-B requests of wide (16 B / lane) coalesced reads as 64 B -> x2; WRITE_SIZE as is.  Working sets below the 256 MiB
Infinity Cache (everything at batch 512) are served from it: the counters then show what reaches the HBM controllers."""
import re
import sqlite3
import sys


def per_kernel(db, counter, pattern):
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute("select kernel_name, grid_size, value from counters_collection where counter_name = ?", (counter,))
    out = {}
    for name, grid, v in rows:
        if not re.search(pattern, name):
            continue
        short = re.sub(r'\(.*', '', name.replace('(anonymous namespace)::', '')).replace('void ', '').replace('i3d::', '')
        short = re.sub(r'Shape<([^>]*)>', lambda m: 'Shape<' + m.group(1).replace(' ', '') + '>', short)
        out.setdefault((short[:70], grid), []).append(v)
    return out


def main(fetch_db, write_db, pattern, json_out=None):
    f = per_kernel(fetch_db, 'FETCH_SIZE', pattern)
    w = per_kernel(write_db, 'WRITE_SIZE', pattern)
    res = {}
    print('# rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE -- <command>   (one pass per counter; the command: profiles/README.md)')
    print('# KiB per dispatch averaged over the dispatches of (kernel, grid); fetch x2 (gfx950 wide-read correction), MB = 1e6 bytes')
    print(f'{"kernel":72s} {"grid":>9s} {"n":>4s} {"fetch_MB":>9s} {"write_MB":>9s} {"total_MB":>9s}')
    for key in sorted(set(f) | set(w)):
        fv, wv = f.get(key, []), w.get(key, [])
        fm = 2.0 * sum(fv) / max(len(fv), 1) * 1024 / 1e6
        wm = sum(wv) / max(len(wv), 1) * 1024 / 1e6
        print(f'{key[0]:72s} {key[1]:9d} {max(len(fv), len(wv)):4d} {fm:9.2f} {wm:9.2f} {fm + wm:9.2f}')
        res.setdefault(key[0], []).append(dict(grid=key[1], dispatches=max(len(fv), len(wv)), fetch_MB=round(fm, 3),
                                               write_MB=round(wm, 3), traffic_MB=round(fm + wm, 3)))
    if json_out:
        import json
        with open(json_out, 'w') as fh:
            json.dump(res, fh, indent=1)


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else '.', sys.argv[4] if len(sys.argv) > 4 else None)
