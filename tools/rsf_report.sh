#!/bin/bash
# Run on the GPU box: the three tables of DESIGN section 2 item 3 (rs_F_2int_3cam) -> gpurun_out/rsf/
O=gpurun_out/rsf; mkdir -p $O
python tools/micro/rsf_trace.py rs_F_2int_3cam 20 2>&1 | grep -v amdgpu > $O/r04_rsf_trace_per_evaluation.txt
python tools/micro/rsf_lsmr_cap.py 2>&1 | grep -v amdgpu > $O/r04_rsf_lsmr_cap.txt
python tools/micro/rsf_clusters.py rs_F_2int_3cam 12 2>&1 | grep -v amdgpu > $O/r04_rsf_clusters.txt
RSF_ONLY_E3=1 MVUS_HOST_EXACT_SUMS=1 python tools/micro/rsf_clusters.py rs_F_2int_3cam 12 2>&1 | grep -v "amdgpu\|^#" >> $O/r04_rsf_clusters.txt
tail -n 30 $O/*.txt
