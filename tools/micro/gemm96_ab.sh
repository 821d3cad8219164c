#!/bin/bash
# (needs tools/experiments/r06_schur_gemm_96x96_lds.patch applied: the kernel is not in the tree)
# usage (GPU box): tools/micro/gemm96_ab.sh  -> k_schur_gemm96 (96 x 96 tiles through LDS) against k_schur_gemm (MVUS_GEMM96=0) and k_rcs_finish behind them,
# rocprofv3 averages over the LM steps of tools/step_breakdown.py; variants/libmvusba_g96s<N>.so = builds with -DMVUS_G96_SETS=N register sets in flight
export TMPDIR=/tmp
run() {  # label, env...
  rm -rf /tmp/pg; env "${@:2}" rocprofv3 --kernel-trace --stats -d /tmp/pg -o r -- python3 tools/step_breakdown.py ${CFG:-2} > /tmp/pg.log 2>&1
  python3 - "$1" <<PY
import sqlite3, sys
cur=sqlite3.connect('/tmp/pg/r_results.db').cursor()
for r in cur.execute("select name,total_calls,average from top_kernels where name like '%k_schur_gemm%' or name like '%k_rcs_finish%'"):
    print(sys.argv[1], r[0][:40], r[1], '%.1f us' % (r[2] / 1e3 if r[2] > 1000 else r[2]))
PY
}
run default
for n in 1 2 3; do [ -f variants/libmvusba_g96s$n.so ] && run sets$n MVUS_LIB_PATH=variants/libmvusba_g96s$n.so; done
run old MVUS_GEMM96=0
for sl in 32 48; do run slabs$sl MVUS_GEMM_SLABS=$sl; done
