#!/bin/bash
# Run on the GPU box (gpurun): regenerates everything under profiles/ for the current code -> gpurun_out/refresh/
set -x
export TMPDIR=/tmp
O=gpurun_out/refresh; mkdir -p $O
python bench.py > $O/bench_config2_lm.json 2> $O/bench_config2_lm.err
python bench.py --solver trf --no-cpu-baseline --no-parity-solver > $O/bench_config2_trf.json 2>> $O/bench.err
python bench.py --config 1 --no-cpu-baseline > $O/bench_config1_lm.json 2>> $O/bench.err
python bench.py --config 4 --no-cpu-baseline > $O/bench_config4_calib_lm.json 2>> $O/bench.err
rocprofv3 --kernel-trace --stats -d $O/stats_lm -o r -- python3 bench.py --no-cpu-baseline --no-parity-solver > $O/stats_lm.log 2>&1
python3 tools/rocprof_summary.py stats $O/stats_lm/r_results.db > $O/kernel_stats_config2_lm.txt
rocprofv3 --kernel-trace --stats -d $O/stats_trf -o r -- python3 bench.py --solver trf --no-cpu-baseline --no-parity-solver > $O/stats_trf.log 2>&1
python3 tools/rocprof_summary.py stats $O/stats_trf/r_results.db > $O/kernel_stats_config2_trf.txt
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/pmc_fetch -o r -- python3 bench.py --no-cpu-baseline --no-parity-solver > $O/pmc_fetch.log 2>&1
python3 tools/rocprof_summary.py pmc $O/pmc_fetch/r_results.db > $O/pmc_FETCH_SIZE_config2.txt
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/pmc_write -o r -- python3 bench.py --no-cpu-baseline --no-parity-solver > $O/pmc_write.log 2>&1
python3 tools/rocprof_summary.py pmc $O/pmc_write/r_results.db > $O/pmc_WRITE_SIZE_config2.txt
rm -rf $O/stats_lm $O/stats_trf $O/pmc_fetch $O/pmc_write
ls -la $O
