#!/bin/bash
# Run on the GPU box: the solver tests, one bench line and the per-kernel stats of an LM run -> gpurun_out/quick/
#   tools/quick_check.sh [pytest targets...]   (default: the Schur / sharding tests)
export TMPDIR=/tmp
O=gpurun_out/quick; mkdir -p $O
T=${@:-tests/test_gpu_schur.py tests/test_gpu_dist.py}
python -m pytest $T -x -q -m gpu > $O/tests.log 2>&1; grep -E "passed|failed|error" $O/tests.log | tail -3
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity-solver > $O/bench.json 2> $O/bench.err
python - <<'PY'
import json
d = json.load(open('gpurun_out/quick/bench.json'))
print('ms/step %.4f  frac %.3f  cost_last %r' % (d['ms_per_step'], d['roofline']['frac'], d.get('cost_last')))
PY
rocprofv3 --kernel-trace --stats -d $O/stats -o r -- python3 tools/micro/host_probe.py > $O/stats.log 2>&1
python3 tools/rocprof_summary.py stats $O/stats/r_results.db | cut -c1-70,111-160 > $O/kernel_stats.txt
rm -rf $O/stats
head -40 $O/kernel_stats.txt
