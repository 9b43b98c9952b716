"""Busy time and inter-kernel gaps per HIP stream from a rocprofv3 --kernel-trace results .db (rocpd sqlite):
    python tools/prof_gaps.py <results.db> [skip_fraction]
For every stream: dispatches, busy time (sum of kernel durations), span, and the histogram of the idle gaps between
consecutive kernels of that stream.  skip_fraction (default 0.35) drops the start of the trace (warm-up)."""
import sqlite3
import sys


def main(path, skip=0.35):
    cur = sqlite3.connect(path).cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    key = 'stream_id' if 'stream_id' in cols else ('queue_id' if 'queue_id' in cols else None)
    rows = list(cur.execute(f"select start, end, {key or '0'}, name from kernels order by start"))
    t0, t1 = rows[0][0], rows[-1][1]
    cut = t0 + skip * (t1 - t0)
    rows = [r for r in rows if r[0] >= cut]
    print(f'# {path}: {len(rows)} dispatches after the first {skip:.0%} of the trace, grouped by {key}')
    streams = {}
    for s, e, q, n in rows:
        streams.setdefault(q, []).append((s, e, n))
    span_all = (rows[-1][1] - rows[0][0]) / 1e3
    # union of busy intervals over all streams
    busy_union, cur_s, cur_e = 0.0, None, None
    for s, e, _, _ in rows:
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                busy_union += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy_union += cur_e - cur_s
    print(f'span {span_all:.1f} us, GPU busy (union over streams) {busy_union / 1e3:.1f} us = {busy_union / 1e3 / span_all:.1%}')
    for q, ks in sorted(streams.items(), key=lambda kv: -len(kv[1])):
        busy = sum(e - s for s, e, _ in ks) / 1e3
        gaps = [(ks[i + 1][0] - ks[i][1]) / 1e3 for i in range(len(ks) - 1)]
        pos = [g for g in gaps if g > 0]
        hist = {}
        for g in pos:
            b = '<1' if g < 1 else '1-2' if g < 2 else '2-4' if g < 4 else '4-8' if g < 8 else '8-16' if g < 16 else '>=16'
            hist[b] = hist.get(b, 0) + 1
        print(f'stream {q}: {len(ks)} kernels, busy {busy:.1f} us ({busy / span_all:.1%} of span), '
              f'gaps: total {sum(pos):.1f} us, median {sorted(pos)[len(pos) // 2] if pos else 0:.2f} us, histogram (us) '
              + ', '.join(f'{b}: {hist[b]}' for b in ('<1', '1-2', '2-4', '4-8', '8-16', '>=16') if b in hist))


if __name__ == '__main__':
    main(sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 0.35)
