#!/bin/bash
# Run on the GPU box (gpurun): regenerates everything under profiles/ for the current code -> gpurun_out/refresh/
#   tools/refresh_profiles.sh [round-tag, default r02]
# Kernel traces and PMC passes are separate rocprofv3 runs (the pool refuses --pmc together with sys/hip traces);
# the program after `--` is python3 itself (no wrapper that re-execs).
set -x
export TMPDIR=/tmp
R=${1:-r02}
O=gpurun_out/refresh; mkdir -p $O
python bench.py --steps 20 --warmup 5 > $O/${R}_bench_config2_lm.json 2> $O/bench_config2_lm.err
python bench.py --solver trf --steps 5 --warmup 1 --no-cpu-baseline --no-parity-solver > $O/${R}_bench_config2_trf.json 2>> $O/bench.err
python bench.py --config 1 --no-cpu-baseline > $O/${R}_bench_config1_lm.json 2>> $O/bench.err
python bench.py --config 4 --no-cpu-baseline > $O/${R}_bench_config4_calib_lm.json 2>> $O/bench.err
python bench.py --config 3 --no-cpu-baseline --no-parity-solver > $O/${R}_bench_config3_one_gpu_lm.json 2>> $O/bench.err
rocprofv3 --kernel-trace --stats -d $O/stats_lm -o r -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity-solver > $O/stats_lm.log 2>&1
python3 tools/rocprof_summary.py stats $O/stats_lm/r_results.db > $O/${R}_kernel_stats_config2_lm.txt
rocprofv3 --kernel-trace --stats -d $O/stats_trf -o r -- python3 bench.py --solver trf --steps 5 --warmup 1 --no-cpu-baseline --no-parity-solver > $O/stats_trf.log 2>&1
python3 tools/rocprof_summary.py stats $O/stats_trf/r_results.db > $O/${R}_kernel_stats_config2_trf.txt
rocprofv3 --kernel-trace --stats -d $O/stats_c3 -o r -- python3 bench.py --config 3 --no-cpu-baseline --no-parity-solver > $O/stats_c3.log 2>&1
python3 tools/rocprof_summary.py stats $O/stats_c3/r_results.db > $O/${R}_kernel_stats_config3_one_gpu_lm.txt
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/pmc_fetch -o r -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity-solver > $O/pmc_fetch.log 2>&1
python3 tools/rocprof_summary.py pmc $O/pmc_fetch/r_results.db > $O/${R}_pmc_FETCH_SIZE_config2.txt
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/pmc_write -o r -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity-solver > $O/pmc_write.log 2>&1
python3 tools/rocprof_summary.py pmc $O/pmc_write/r_results.db > $O/${R}_pmc_WRITE_SIZE_config2.txt
python3 tools/rocprof_summary.py traffic $O/${R}_pmc_FETCH_SIZE_config2.txt $O/${R}_pmc_WRITE_SIZE_config2.txt > $O/pmc_traffic.json
rm -rf $O/stats_lm $O/stats_trf $O/stats_c3 $O/pmc_fetch $O/pmc_write
ls -la $O
