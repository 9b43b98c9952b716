#!/bin/bash
# HBM traffic (PMC) of the tower variant's step kernels at batch 512: two passes, one counter each (guide: separate --pmc passes)
ROOT=$(pwd); OUT=$ROOT/gpurun_out/tower; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/tpmc_$c
  rocprofv3 --kernel-trace --pmc $c -d /tmp/tpmc_$c -o p -- python $ROOT/tools/tower_step.py 512 8 > $OUT/pmc_$c.log 2>&1
done
F=$(find /tmp/tpmc_FETCH_SIZE -name "*_results.db" | head -1); W=$(find /tmp/tpmc_WRITE_SIZE -name "*_results.db" | head -1)
python $ROOT/tools/pmc_summary.py $F $W 'pna_aggregate|gemm_f32_kernel|gemm_f32_rowseg|edge_combine' > $OUT/tower_pmc_b512.txt 2>&1
cd $ROOT
