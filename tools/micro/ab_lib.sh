# usage (GPU box): tools/micro/ab_lib.sh variant... -> LM step time at configs[2], [3], [1], [4] per variants/libmvusba_<variant>.so, and the kernels of the band solve
export TMPDIR=/tmp
for v in "$@"; do export MVUS_LIB_PATH=variants/libmvusba_$v.so
  for c in 2 3 1 4; do python bench.py --config $c --steps 30 --warmup 5 --no-cpu-baseline --no-parity-solver 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$v config $c', round(d['ms_per_step'],4), repr(d['config'].get('cost_last')))"; done
  rm -rf gpurun_out/prof_ab; rocprofv3 --kernel-trace --stats -d gpurun_out/prof_ab -o r -- python3 tools/step_breakdown.py 2 > /dev/null 2>&1
  python3 - <<PY
import sqlite3
cur=sqlite3.connect('gpurun_out/prof_ab/r_results.db').cursor()
for r in cur.execute("select name,total_calls,average from top_kernels where name like '%k_cholesky_and_rhs%' or name like '%k_part_solve%' or name like '%k_part_back%' or name like '%k_sep_bcr_rhs%' or name like '%k_bcr_tail%' or name like '%k_part_reduce%' or name like '%k_sep_bcr_level%'"):
    print('$v', r[0][:50], r[1], '%.1f us' % r[2])
PY
done; rm -rf gpurun_out/prof_ab
