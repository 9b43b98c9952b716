#!/bin/bash
# rocprofv3 kernel traces of the step workloads -> gpurun_out/prof_<tag>/ + text summaries gpurun_out/<tag>_{trace,timeline}.txt
#   tools/collect_traces.sh  (on the GPU box; cd /tmp && TMPDIR=/tmp as the guide prescribes for rocprofv3)
ROOT=$(pwd)
OUT=$ROOT/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
run() {   # tag, steps-for-summary, command...
  tag=$1; shift
  (cd /tmp && rm -rf /tmp/prof_$tag && rocprofv3 --kernel-trace -d /tmp/prof_$tag -o $tag -- "$@" > $OUT/${tag}_cmd.log 2>&1)
  db=$(find /tmp/prof_$tag -name "*_results.db" | head -1)
  if [ -n "$db" ]; then
    python $ROOT/tools/prof_summary.py $db > $OUT/${tag}_trace.txt 2>&1
    python $ROOT/tools/prof_step_timeline.py $db > $OUT/${tag}_timeline.txt 2>&1
  else
    echo "no results db for $tag" > $OUT/${tag}_trace.txt
  fi
}
run step python $ROOT/bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-families --no-extra-workloads
run qmugs_fp32 python $ROOT/bench.py --workload qmugs --steps 10 --warmup 5 --no-cpu-baseline
run qmugs_bf16 python $ROOT/bench.py --workload qmugs --dtype bf16 --steps 10 --warmup 5 --no-cpu-baseline
run finetune python $ROOT/tools/finetune_step.py --steps 20 --warmup 5
