#!/bin/bash
# main-loop ablation of the fp32 GEMM (tools/probes/ablate/gemm_ablate.hip = csrc/gemm.hip with one piece of the K loop
# compiled out): which of {global loads, MFMAs, LDS stores + barrier} bounds a tile?
for v in BASE NOLOAD NOMFMA NOSTORE; do
  echo "== $v"; I3D_LIB_PATH=$PWD/tools/probes/ablate/lib_$v.so python tools/gemm_bench.py 2>/dev/null | grep -E "^(P |Q |post4|dgrad  dY\[E|P2|wgrad  dY\^T\[F,E\])" | grep -v atomics | cut -c1-110
done
