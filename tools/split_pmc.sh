#!/bin/bash
# PMC passes over tools/gemm_pmc.py: native fp32 products vs the split form
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/split_pmc; mkdir -p $O
for mode in native split; do
  export I3D_FP32_PRODUCTS=$mode
  i=0
  for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" "SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM" "SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_ANY"; do
    i=$((i+1))
    timeout 300 rocprofv3 --kernel-trace --pmc $set -d /tmp/pmc_${mode}_$i -o p -- python tools/gemm_pmc.py > /tmp/pmc_${mode}_$i.log 2>&1
  done
  python tools/gemm_pmc.py --summarise $(find /tmp/pmc_${mode}_* -name '*results.db') > $O/$mode.txt 2>&1
done
