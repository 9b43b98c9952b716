#!/bin/bash
# tools/blas_compare.py under rocprofv3 --kernel-trace: the table + the vendor kernels' names (profiles/r04_blas_compare.txt)
export TMPDIR=/tmp; ROOT=$(pwd); cd /tmp; rm -rf /tmp/blas
rocprofv3 --kernel-trace -d /tmp/blas -o b -- python $ROOT/tools/blas_compare.py 2>&1 | grep -v "^W2026\|amdgpu.ids"
F=$(find /tmp/blas -name "*_results.db" | head -1)
python $ROOT/tools/prof_summary.py $F | cut -c1-200 | head -40
python - <<PY
import sqlite3
cur=sqlite3.connect("$F").cursor()
for r in cur.execute("select name, count(*), avg(end-start)/1000.0 from kernels where name like 'Cijk%' group by name order by count(*) desc"):
    print(r[1], round(r[2],1), r[0][:400])
PY
