#!/bin/bash
# PMC / SQ evidence for the BatchNorm-backward family next to K4 on the same tensors -> profiles/r06_bn_pmc.txt
# one rocprofv3 pass per counter group (counters only with --kernel-trace: gpurun's rule), then a plain kernel trace for the durations
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
TAG=${1:-r06_bn_pmc}
O=gpurun_out/$TAG; mkdir -p $O
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU" \
           "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $set -d /tmp/${TAG}_$i -o p -- python $GRAFT_REPO_ROOT/tools/bn_pmc.py > /tmp/${TAG}_$i.log 2>&1) || echo "pass $i ($set) failed: $(tail -3 /tmp/${TAG}_$i.log)"
done
(cd /tmp && timeout 300 rocprofv3 --kernel-trace -d /tmp/${TAG}_t -o t -- python $GRAFT_REPO_ROOT/tools/bn_pmc.py --reps 20 > /tmp/${TAG}_t.log 2>&1)
{
  echo "# tools/bn_pmc.sh: rocprofv3 --kernel-trace --pmc <group> -- python tools/bn_pmc.py (one pass per group), durations from a separate plain kernel trace"
  tail -1 /tmp/${TAG}_t.log
  python tools/bn_pmc.py --summarise $(find /tmp/${TAG}_[0-9]* -name '*results.db' | sort) --trace $(find /tmp/${TAG}_t -name '*results.db' | head -1)
} > $O/summary.txt 2>&1
cat $O/summary.txt | head -150
