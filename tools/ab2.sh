#!/bin/bash
# as tools/ab.sh, printing ms_per_step AND the host enqueue time per step:  tools/ab2.sh N "ENV=1 --flag" ...
N=$1; shift
for i in $(seq $N); do
  for v in "$@"; do
    envs=""; flags=""
    for w in $v; do case "$w" in *=*) envs="$envs $w";; *) flags="$flags $w";; esac; done
    echo -n "[$v]: "
    env $envs timeout 300 python bench.py --steps 300 --warmup 60 --no-cpu-baseline --no-families --no-extra-workloads $flags 2>&1 | grep "^{" | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print(j['ms_per_step'], j['config']['host_enqueue_ms_per_step'])"
  done
done | python -c "
import sys, collections
d = collections.defaultdict(list)
for l in sys.stdin:
    k, v = l.rsplit(':', 1); a, b = v.split(); d[k].append((float(a), float(b)))
for k, v in d.items():
    v.sort(); print(f'{k:60s} step min {v[0][0]:.3f} median {v[len(v) // 2][0]:.3f}   host min {min(x[1] for x in v):.3f} median {sorted(x[1] for x in v)[len(v) // 2]:.3f}')
"
