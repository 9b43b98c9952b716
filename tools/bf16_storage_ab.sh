#!/bin/bash
# g1 (configs[3]: bf16 storage): what bf16 STORAGE buys where the tensors leave the Infinity Cache.  The storage that exists - the [E,F]
# messages of every PNA layer (written by FC2's epilogue, read by K4 forward, K4 backward and the BatchNorm backward) and the 3D edge
# stage's [E3,20] tensors - switched off / on in the bf16 matmul mode at batch 512 and 4096 (interleaved runs, one box)
cd $GRAFT_REPO_ROOT
run() { # label env... flags
  label=$1; shift
  env "$@" 2>/dev/null | grep "^{" | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('%-70s ms_per_step %.3f  molecules/s %.0f' % ('$label', j['ms_per_step'], j['value']))"
}
F="--no-cpu-baseline --no-families --no-extra-workloads"
for rep in 1 2 3; do
  run "batch 4096 bf16 matmul mode, messages fp32 (I3D_MSG_BF16=0)"      I3D_MSG_BF16=0 python bench.py --batch 4096 --dtype bf16 --steps 40 --warmup 10 $F
  run "batch 4096 bf16 matmul mode, messages bf16 (default at this size)" I3D_MSG_BF16=f python bench.py --batch 4096 --dtype bf16 --steps 40 --warmup 10 $F
  run "batch 512  bf16 matmul mode, messages fp32 (default at this size)" I3D_MSG_BF16=0 python bench.py --batch 512 --dtype bf16 --steps 200 --warmup 40 $F
  run "batch 512  bf16 matmul mode, messages bf16 (I3D_MSG_BF16=force)"   I3D_MSG_BF16=f python bench.py --batch 512 --dtype bf16 --steps 200 --warmup 40 $F
done
run "batch 4096 fp32 (split products)" I3D_MSG_BF16=0 python bench.py --batch 4096 --steps 40 --warmup 10 $F
