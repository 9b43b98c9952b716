"""Look up the HBM traffic per launch of a kernel in a PMC summary written by tools/pmc_summary.py.

bench.py's `roofline.traffic` comes from here.  Kernel names are matched on the BASE name plus the LEADING template
arguments (`pna_aggregate_fwd_kernel` + ('2',) matches `pna_aggregate_fwd_kernel<2>`, `<2, false>`, `<2,false,1>` ...), so
a kernel that gains a trailing template parameter keeps matching; rows are matched to the workload by the launch's grid
size (a K4 launch has one lane per (node, 4 features): grid = 256 * ceil(N * ceil(F / 4) / 256) work items), so a line
for another workload does not report the traffic of this one.  A summary file that exists and matches nothing is an error
the caller has to show (bench.py: stderr + `roofline.traffic_error`), not a silent null."""
import json
import re


class PmcLookupError(RuntimeError):
    pass


def split_kernel_name(name):
    """'foo_kernel<4, 2, false>' -> ('foo_kernel', ('4', '2', 'false')); 'foo_kernel' -> ('foo_kernel', ())."""
    m = re.match(r'^\s*([A-Za-z_][\w:]*)\s*(?:<(.*)>)?\s*$', name)
    if not m:
        return name.strip(), ()
    args = tuple(a.strip() for a in m.group(2).split(',')) if m.group(2) is not None else ()
    return m.group(1), args


def k4_grid(num_nodes, feat):
    """Work items of a K4 launch (csrc/aggregate.hip: one lane per (node, 4 features), workgroups of 256)."""
    items = num_nodes * ((feat + 3) // 4)
    return 256 * ((items + 255) // 256)


def load(path):
    with open(path) as f:
        return json.load(f)


def rows_for(summary, base, leading_args=(), grids=None):
    """Rows ({grid, dispatches, fetch_MB, write_MB, traffic_MB}) of every kernel entry whose base name is `base` and whose
    template arguments start with `leading_args`; `grids`: keep only rows launched with one of these grid sizes."""
    leading_args = tuple(str(a) for a in leading_args)
    out = []
    for key, rows in summary.items():
        if key.startswith('_'):
            continue
        b, args = split_kernel_name(key)
        if b != base or args[:len(leading_args)] != leading_args:
            continue
        for r in rows:
            if grids is None or r['grid'] in grids:
                out.append(r)
    return out


class PmcNotCovered(LookupError):
    """the summary holds the kernel, but no launch with this workload's grids: the PMC pass was over another workload"""


def traffic_bytes(summary, base, leading_args=(), grids=None):
    """Dispatch-weighted mean HBM bytes per launch.  PmcNotCovered: the kernel is in the file but was never launched with one of
    `grids` (another workload's pass: no number to attach, not an error); PmcLookupError: the kernel is not in the file at all (a
    renamed / re-templated kernel: the caller has to show it) - both name what was looked for and what the file holds."""
    rows = rows_for(summary, base, leading_args, grids)
    if not rows and grids is not None and rows_for(summary, base, leading_args, None):
        have = sorted({r['grid'] for r in rows_for(summary, base, leading_args, None)})
        raise PmcNotCovered(f'{base}<{", ".join(map(str, leading_args))}...> is in the summary with grids {have[:4]}'
                            f'{"..." if len(have) > 4 else ""}, none of {sorted(grids)}: the PMC pass was over another workload')
    if not rows:
        names = sorted(k for k in summary if not k.startswith('_'))
        have = sorted({r['grid'] for k in names if split_kernel_name(k)[0] == base for r in summary[k]})
        raise PmcLookupError(f'no PMC row for {base}<{", ".join(map(str, leading_args))}...> '
                             f'with grid in {sorted(grids) if grids is not None else "any"}; kernels in the file: {names}; '
                             f'grids of {base}: {have[:6]}{"..." if len(have) > 6 else ""}')
    n = sum(r['dispatches'] for r in rows)
    return int(1e6 * sum(r['traffic_MB'] * r['dispatches'] for r in rows) / n)
