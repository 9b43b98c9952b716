"""One training step as a time line from a rocprofv3 --kernel-trace results .db: every kernel of the step (between two
consecutive Adam launches, taken from the middle of the trace) with its stream, start offset, duration and the gap to
the previous kernel of the same stream.
    python tools/prof_step_timeline.py <results.db> [which_step]"""
import re
import sqlite3
import sys


def main(path, which=None):
    cur = sqlite3.connect(path).cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    key = 'stream_id' if 'stream_id' in cols else 'queue_id'
    rows = list(cur.execute(f"select start, end, {key}, name from kernels order by start"))
    adam = [i for i, r in enumerate(rows) if 'adam_kernel' in r[3]]
    k = which if which is not None else len(adam) // 2
    lo, hi = adam[k] + 1, adam[k + 1] + 1
    step = rows[lo:hi]
    t0 = step[0][0]
    streams = sorted({r[2] for r in step}, key=lambda q: -sum(1 for r in step if r[2] == q))
    label = {q: chr(ord('A') + i) for i, q in enumerate(streams)}
    last_end = {}
    print(f'# step {k}: {len(step)} kernels, {(step[-1][1] - t0) / 1e3:.1f} us from the first start to the end of Adam; '
          + ', '.join(f'stream {label[q]}: {sum(1 for r in step if r[2] == q)} kernels, '
                      f'{sum(r[1] - r[0] for r in step if r[2] == q) / 1e3:.0f} us busy' for q in streams))
    print(f'{"start_us":>9s} {"dur_us":>7s} {"gap_us":>7s} s kernel')
    for s, e, q, n in step:
        short = re.sub(r'\(.*', '', n.replace('(anonymous namespace)::', '')).replace('void ', '').replace('i3d::', '')
        short = re.sub(r'Shape<([^>]*)>', lambda m: 'S<' + m.group(1).replace(' ', '') + '>', short)
        gap = (s - last_end[q]) / 1e3 if q in last_end else 0.0
        last_end[q] = e
        print(f'{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:7.1f} {gap:7.1f} {label[q]} {short[:100]}')


if __name__ == '__main__':
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else None)
