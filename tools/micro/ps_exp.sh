export TMPDIR=/tmp
for v in base ps1 ps2 ps3; do
  rm -rf gpurun_out/prof_ps
  MVUS_LIB_PATH=variants/libmvusba_$v.so rocprofv3 --kernel-trace --stats -d gpurun_out/prof_ps -o r -- python3 tools/step_breakdown.py 2 > /dev/null 2>&1
  python3 - <<PY
import sqlite3
cur=sqlite3.connect('gpurun_out/prof_ps/r_results.db').cursor()
for r in cur.execute("select name,total_calls,average from top_kernels where name like '%k_part_solve%' or name like '%k_part_back%' or name like '%k_cholesky_and_rhs%'"):
    print('$v', r[0][:50], r[1], '%.1f us' % r[2])
PY
done
rm -rf gpurun_out/prof_ps
