export TMPDIR=/tmp
bash tools/micro/pmc_gemm.sh 2 > gpurun_out/pmc_gemm2.log 2>&1
MVUS_LIB_PATH=variants/libmvusba_gjprobe.so python tools/micro/gj_probe.py 2>&1 | grep -i "cycles" | sort | uniq -c | sort -rn | head -40 > gpurun_out/gj_probe.log
for seed in 1 2 3; do echo "== seed $seed trf motion_weights 1e4"; python3 tools/incremental_loop.py --solver trf --motion-weights 1e4 --seed $seed 2>&1 | grep -v "^Number\|^Doing\|^$" | tail -8; done > gpurun_out/loop_mw1e4.log 2>&1
python3 tools/incremental_loop.py --solver trf --obs 20000 --cpu-all 2>&1 | grep -v "^Number\|^Doing\|^$" > gpurun_out/loop_cpu_all.log 2>&1
python3 tools/micro/time_scene_ba.py > gpurun_out/scene_ba.log 2>&1
