#!/bin/bash
# main-loop ablation of the fp32 GEMM (tools/probes/ablate/gemm_ablate.hip = csrc/gemm.hip with one piece of the K loop
# compiled out): which of {global loads, MFMAs, LDS stores + barrier} bounds a tile?
# build (in the build container; the .so files travel with the snapshot):
#   python -c "import __graft_entry__ as g; g.build()"
#   for v in BASE NOLOAD NOMFMA NOSTORE; do f=""; [ $v != BASE ] && f="-DI3D_ABLATE_$v"
#     hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off $f -c tools/probes/ablate/gemm_ablate.hip -o /tmp/gemm_$v.o
#     hipcc --offload-arch=gfx950 -shared -fPIC -o tools/probes/ablate/lib_$v.so $(ls 3dinfomax_amd/build/*.o | grep -v gemm.o) /tmp/gemm_$v.o; done
# (gemm_ablate.hip is csrc/gemm.hip with three #ifdef'd pieces of the K loop; regenerate it when gemm.hip changes)
for v in BASE NOLOAD NOMFMA NOSTORE; do
  echo "== $v"; I3D_LIB_PATH=$PWD/tools/probes/ablate/lib_$v.so python tools/gemm_bench.py 2>/dev/null | grep -E "^(P |Q |post4|dgrad  dY\[E|P2|wgrad  dY\^T\[F,E\])" | grep -v atomics | cut -c1-110
done
