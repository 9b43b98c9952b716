cd /tmp && export TMPDIR=/tmp
for PB in 1024 512 256; do
rm -rf /tmp/prof; I3D_FUSED_FINAL=${FF:-1} I3D_PARTIAL_BLOCKS=$PB rocprofv3 --kernel-trace --stats -d /tmp/prof -o r02 -- python $GRAFT_REPO_ROOT/bench.py --steps 40 --warmup 9 --no-cpu-baseline --loader-workers 0 > /tmp/bench_prof_$PB.log 2>&1
DB=$(find /tmp/prof -name "*.db" | head -1); python $GRAFT_REPO_ROOT/tools/prof_summary.py $DB 69 > $GRAFT_REPO_ROOT/gpurun_out/r02_trace_pb$PB.txt
echo "== PB $PB"; grep -E "total kernel|colreduce|bn_bwd_apply|bn_apply|finalize|_final_kernel" $GRAFT_REPO_ROOT/gpurun_out/r02_trace_pb$PB.txt | cut -c1-150
grep -o '"ms_per_step": [0-9.]*' /tmp/bench_prof_$PB.log
done
