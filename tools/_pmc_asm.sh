#!/bin/bash
export TMPDIR=/tmp
for grp in "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES"; do
  tag=$(echo $grp | cut -d' ' -f1)
  rocprofv3 --pmc $grp --kernel-trace -d gpurun_out/pmc_$tag -o r -- python3 tools/_time_asm.py > /dev/null 2>&1
  python3 - <<PY
import sqlite3
cur=sqlite3.connect('gpurun_out/pmc_$tag/r_results.db').cursor()
try:
    for r in cur.execute("select kernel_name,counter_name,count(*),avg(value) from counters_collection where kernel_name like '%assemble_spans%' group by kernel_name,counter_name"):
        print(r[0][:40], r[1], r[2], '%.4g'%r[3])
except Exception as e: print('err', e)
PY
done
