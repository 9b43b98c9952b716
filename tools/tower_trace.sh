ROOT=$(pwd); OUT=$ROOT/gpurun_out/tower; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
for b in 128 512; do
  rm -rf /tmp/tw_$b
  rocprofv3 --kernel-trace -d /tmp/tw_$b -o t -- python $ROOT/tools/tower_step.py $b 60 > $OUT/run_$b.log 2>&1
  F=$(find /tmp/tw_$b -name "*_results.db" | head -1)
  python $ROOT/tools/prof_summary.py $F > $OUT/tower_kernel_trace_b$b.txt 2>&1
  python $ROOT/tools/prof_step_timeline.py $F > $OUT/tower_timeline_b$b.txt 2>&1
done
cd $ROOT; python tools/tower_step.py 128 200; python tools/tower_step.py 512 200
