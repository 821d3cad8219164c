export TMPDIR=/tmp
python -m pytest tests/test_gpu_schur.py tests/test_gpu_dist.py -x -q > gpurun_out/gj_tests.log 2>&1; grep -a "passed\|failed" gpurun_out/gj_tests.log
for c in 2 3 1 4; do python bench.py --config $c --steps 30 --warmup 5 --no-cpu-baseline --no-parity-solver 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('config $c', round(d['ms_per_step'],4), repr(d['config'].get('cost_last')))"; done
tools/prof_steps.sh 2 gj2 2>&1 | grep -i "gj_step\|finish\|kernel time"
tools/prof_steps.sh 3 gj3 2>&1 | grep -i "gj_step\|finish\|kernel time"
