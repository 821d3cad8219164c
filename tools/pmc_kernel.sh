#!/bin/bash
# usage (GPU box): tools/pmc_kernel.sh <kernel-name-pattern> [config]   -- prints average PMC counters per dispatch
export TMPDIR=/tmp
pat=$1; cfg=${2:-2}
for grp in "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS"; do
  tag=$(echo $grp | cut -d' ' -f1)
  rocprofv3 --pmc $grp --kernel-trace -d gpurun_out/pmck_$tag -o r -- python3 tools/step_breakdown.py $cfg > /dev/null 2>&1
  python3 - <<PY
import sqlite3
cur=sqlite3.connect('gpurun_out/pmck_$tag/r_results.db').cursor()
try:
    import re
    for r in cur.execute("select kernel_name,counter_name,count(*),avg(value) from counters_collection group by kernel_name,counter_name"):
        if re.search(r'$pat', r[0]): print(re.sub(r'^void |mvus::', '', r[0])[:28], r[1], r[2], '%.4g'%r[3])
except Exception as e: print('err', e)
PY
done
