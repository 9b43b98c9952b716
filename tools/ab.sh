#!/bin/bash
# A/B of environment switches on ONE box, interleaved, min and median of N runs each (box-to-box and minute-to-minute
# host noise is +-10 %):   tools/ab.sh N "ENV_A=1" "ENV_B=1 --depth 7" ...   (each argument: env assignments and/or bench flags)
N=$1; shift
for i in $(seq $N); do
  for v in "$@"; do
    envs=""; flags=""
    for w in $v; do case "$w" in *=*) envs="$envs $w";; *) flags="$flags $w";; esac; done
    echo -n "[$v]: "
    env $envs timeout 300 python bench.py --steps 300 --warmup 60 --no-cpu-baseline $flags 2>&1 | grep "^{" | tail -1 | python -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])"
  done
done | python -c "
import sys, collections
d = collections.defaultdict(list)
for l in sys.stdin:
    k, v = l.rsplit(':', 1); d[k].append(float(v))
for k, v in d.items():
    v.sort(); print(f'{k:50s} min {v[0]:.3f}  median {v[len(v) // 2]:.3f}  all {v}')
"
