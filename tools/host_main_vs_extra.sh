#!/bin/bash
# the main timed region's host enqueue time against the extra windows of the same workload, by what the main loop adds
# (K4 timing events on its dispatches, per-step marks), pool size and run length -> profiles/r06_host_main_vs_extra.txt
cd $GRAFT_REPO_ROOT
run() { # label, flags
  label=$1; shift
  python bench.py --no-cpu-baseline --no-families --no-extra-workloads "$@" 2>/dev/null | grep "^{" | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read()); c=j['config']
print('%-62s ms_per_step %.3f  host_enqueue %.3f  median %s' % ('$label', j['ms_per_step'], c['host_enqueue_ms_per_step'], c.get('ms_per_step_median')))"
}
echo "# python bench.py --no-cpu-baseline --no-families --no-extra-workloads <flags>   (batch 512, depth 4, fp32; one box, in this order)"
for rep in 1 2; do
for steps in 25 300; do
  w=$((steps/5))
  run "steps $steps timers-in-main marks pool4 (rounds 4-5)"  --steps $steps --warmup $w --timers-in-main
  run "steps $steps timers-in-main no-marks pool4"            --steps $steps --warmup $w --timers-in-main --no-marks
  run "steps $steps marks pool4 (default now)"                --steps $steps --warmup $w
  run "steps $steps no-marks pool4"                           --steps $steps --warmup $w --no-marks
  run "steps $steps no-marks pool1"                           --steps $steps --warmup $w --no-marks --pool 1
  run "steps $steps marks pool1"                              --steps $steps --warmup $w --pool 1
done
done
python bench.py --no-cpu-baseline --no-families --steps 50 --warmup 10 2>/dev/null | grep "^{" | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read()); c=j['config']
print('default line: ms_per_step %.3f host_enqueue %.3f' % (j['ms_per_step'], c['host_enqueue_ms_per_step']))
for k,v in c['extra_workloads'].items():
    if isinstance(v, dict) and 'host_enqueue_ms' in v: print('  extra window %-48s ms_per_step %.3f host_enqueue %.3f' % (k, v['ms_per_step'], v['host_enqueue_ms']))"
