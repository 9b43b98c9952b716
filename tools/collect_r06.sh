#!/bin/bash
# End-of-round artefacts on the GPU box -> gpurun_out/r06/  (bench lines, PMC passes over the step's K4 / weight-gradient kernels,
# microbenchmarks).  Kernel traces: tools/collect_traces.sh.
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/r06
mkdir -p $OUT
export TMPDIR=/tmp
B="python $ROOT/bench.py"
$B > $OUT/bench.json 2> $OUT/bench.err
$B --depth 7 --no-cpu-baseline --no-families --no-extra-workloads > $OUT/bench_depth7.json 2>/dev/null
$B --batch 4096 --no-cpu-baseline --no-families --no-extra-workloads > $OUT/bench_B4096.json 2>/dev/null
$B --dtype bf16 --no-cpu-baseline > $OUT/bench_bf16.json 2>/dev/null
$B --workload qmugs --no-cpu-baseline > $OUT/bench_qmugs.json 2>/dev/null
$B --workload qmugs --dtype bf16 --no-cpu-baseline > $OUT/bench_qmugs_bf16.json 2>/dev/null
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29541 $ROOT/bench.py --gpus 1 --force-dist \
    --no-cpu-baseline --no-families --no-extra-workloads --steps 100 --warmup 20 2>/dev/null | tail -1 > $OUT/bench_force_dist_world1.json
I3D_SYNC_PROVIDER=rccl python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29542 $ROOT/bench.py --gpus 1 \
    --force-dist --no-cpu-baseline --no-families --no-extra-workloads --steps 100 --warmup 20 2>/dev/null | tail -1 > $OUT/bench_force_dist_world1_rccl.json
python $ROOT/bench.py --gpus 2 --backend gloo \
    --no-cpu-baseline --no-families --no-extra-workloads --steps 20 --warmup 5 2>/dev/null | tail -1 > $OUT/bench_dp2_gloo_one_gpu.json
python $ROOT/tools/wgrad_bench.py > $OUT/wgrad_bench.txt 2>&1
python $ROOT/tools/fused_gemm_bench.py > $OUT/fused_gemm_bench.txt 2>&1
python $ROOT/tools/tower_segments.py > $OUT/tower_segments.txt 2>&1
python $ROOT/tools/finetune_step.py --steps 100 --warmup 20 > $OUT/finetune_step.txt 2>&1
# PMC: one counter per pass (FETCH_SIZE 3 TCC slots, WRITE_SIZE 2), kernel trace only next to it
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  rocprofv3 --kernel-trace --pmc $c -d /tmp/pmc_$c -o p -- python $ROOT/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-families --no-extra-workloads > $OUT/pmc_$c.log 2>&1
done
F=$(find /tmp/pmc_FETCH_SIZE -name "*_results.db" | head -1); W=$(find /tmp/pmc_WRITE_SIZE -name "*_results.db" | head -1)
python $ROOT/tools/pmc_summary.py $F $W 'pna_aggregate|wgrad_multi|wgrad_reduce|bn_bwd_fused|panel_gemm' $OUT/step_k4_pmc.json > $OUT/step_k4_pmc.txt 2>&1
cd $ROOT
# the 3D network alone at the configs[3] shape + its PMC pass (tools/n3_bench.py)
python $ROOT/tools/n3_bench.py > $OUT/n3_bench_fp32.txt 2>&1
python $ROOT/tools/n3_bench.py --dtype bf16 > $OUT/n3_bench_bf16.txt 2>&1
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmcn_$c
  rocprofv3 --kernel-trace --pmc $c -d /tmp/pmcn_$c -o p -- python $ROOT/tools/n3_bench.py --steps 3 > /dev/null 2>&1
done
F=$(find /tmp/pmcn_FETCH_SIZE -name "*_results.db" | head -1); W=$(find /tmp/pmcn_WRITE_SIZE -name "*_results.db" | head -1)
python $ROOT/tools/pmc_summary.py $F $W 'n3_|segment_sum' > $OUT/net3d_edge_pmc_qmugs.txt 2>&1
cd $ROOT
