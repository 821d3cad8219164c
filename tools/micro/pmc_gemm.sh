#!/bin/bash
# usage (GPU box): tools/micro/pmc_gemm.sh [config] -- counters of k_schur_gemm (average per dispatch), one rocprofv3 --pmc pass per group
export TMPDIR=/tmp
cfg=${1:-2}
for grp in "SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES" "SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum" "TCC_REQ_sum TCC_TAG_STALL_sum"; do
  tag=$(echo $grp | cut -d' ' -f1)
  rm -rf gpurun_out/pmcg_$tag
  rocprofv3 --pmc $grp --kernel-trace -d gpurun_out/pmcg_$tag -o r -- python3 tools/step_breakdown.py $cfg > /dev/null 2>&1
  python3 - <<PY
import sqlite3, re, glob
dbs = glob.glob('gpurun_out/pmcg_$tag/**/r_results.db', recursive=True) + glob.glob('gpurun_out/pmcg_$tag/r_results.db')
if not dbs: print('$grp: no result (counter not available?)')
for db in dbs[:1]:
    cur = sqlite3.connect(db).cursor()
    try:
        for r in cur.execute("select kernel_name,counter_name,count(*),avg(value) from counters_collection group by kernel_name,counter_name"):
            if re.search(r'k_schur_gemm', r[0]): print('k_schur_gemm', r[1], r[2], '%.6g' % r[3])
    except Exception as e: print('err', e)
PY
  rm -rf gpurun_out/pmcg_$tag
done
