#!/bin/bash
# usage (on the GPU box): tools/prof_steps.sh <config> <tag>  -> gpurun_out/prof_<tag>/ + printed per-step kernel table
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$2 -o r -- python3 tools/step_breakdown.py $1 > gpurun_out/prof_$2.log 2>&1
tail -1 gpurun_out/prof_$2.log
python3 - <<PY
import sqlite3
cur=sqlite3.connect('gpurun_out/prof_$2/r_results.db').cursor()
rows=list(cur.execute('select name,total_calls,total_duration,average from top_kernels order by total_duration desc'))
tot=sum(r[2] for r in rows)
print('GPU kernel time per step (23 solves): %.3f ms'%(tot/1e3/23))
for r in rows[:26]:
    print('%-58s %5d calls  %8.1f us avg  %7.3f ms/step'%(r[0][:58],r[1],r[3],r[2]/1e3/23))
PY
