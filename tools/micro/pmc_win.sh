#!/bin/bash
# usage (GPU box): tools/micro/pmc_win.sh [config] -- SQ counters of the window-major assembly kernel (average per dispatch)
export TMPDIR=/tmp
cfg=${1:-2}
for grp in "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_MFMA" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_FLAT" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT"; do
  tag=$(echo $grp | cut -d' ' -f1)
  rm -rf gpurun_out/pmcw_$tag
  rocprofv3 --pmc $grp --kernel-trace -d gpurun_out/pmcw_$tag -o r -- python3 tools/micro/time_win.py $cfg > /dev/null 2>&1
  python3 - <<PY
import sqlite3, re, glob
for db in glob.glob('gpurun_out/pmcw_$tag/**/r_results.db', recursive=True) + glob.glob('gpurun_out/pmcw_$tag/r_results.db'):
    cur = sqlite3.connect(db).cursor()
    try:
        for r in cur.execute("select kernel_name,counter_name,count(*),avg(value) from counters_collection group by kernel_name,counter_name"):
            if re.search(r'k_assemble_windows|k_assemble_spans', r[0]): print(re.sub(r'^void |mvus::', '', r[0])[:28], r[1], r[2], '%.5g' % r[3])
    except Exception as e: print('err', e)
    break
PY
  rm -rf gpurun_out/pmcw_$tag
done
