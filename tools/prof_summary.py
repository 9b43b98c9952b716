"""Summarise a rocprofv3 --kernel-trace results .db (rocpd sqlite) as text: per-kernel calls / total / average /
share, like `--stats`.  Usage: python tools/prof_summary.py <results.db> [steps]"""
import sqlite3
import sys


def main(path, steps=None):
    cur = sqlite3.connect(path).cursor()
    rows = list(cur.execute(
        "select name, count(*), sum(end-start)/1000.0, avg(end-start)/1000.0, min(end-start)/1000.0, "
        "max(end-start)/1000.0 from kernels group by name order by sum(end-start) desc"))
    total = sum(r[2] for r in rows)
    print(f'# rocprofv3 --kernel-trace summary of {path}')
    print(f'# total kernel time {total / 1e3:.3f} ms over {sum(r[1] for r in rows)} dispatches'
          + (f'  ({total / 1e3 / steps:.3f} ms per step over {steps} steps)' if steps else ''))
    print(f'{"kernel":72s} {"calls":>6s} {"total_us":>10s} {"avg_us":>8s} {"min_us":>8s} {"max_us":>8s} {"pct":>6s}')
    for name, calls, tot, avg, mn, mx in rows:
        print(f'{name[:72]:72s} {calls:6d} {tot:10.1f} {avg:8.2f} {mn:8.2f} {mx:8.2f} {100 * tot / total:6.2f}')


if __name__ == '__main__':
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else None)
