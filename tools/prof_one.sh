#!/bin/bash
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d gpurun_out/p1_$2 -o r -- python3 tools/step_breakdown.py 2 > /dev/null 2>&1
python3 - <<PY
import sqlite3,re
cur=sqlite3.connect('gpurun_out/p1_$2/r_results.db').cursor()
for r in cur.execute('select name,average from top_kernels'):
    if re.search(r'$1', r[0]): print('$2', r[0][:40], '%.1f us'%r[1])
PY
