#!/bin/bash
# usage (GPU box): tools/micro/shard_rank_probe.sh  -> per-kernel times of ONE rank of an 8-rank time-sharded run, alone on the GPU
# (callbacks: no-ops): configs[3] cut over 8 ranks (strong scaling), two-level and one-level separators; configs[2] x 8 detections (bench.py's weak scaling)
export TMPDIR=/tmp
run() {   # env-assignment cfg mult
  rm -rf /tmp/sp; env $1 rocprofv3 --kernel-trace --stats -d /tmp/sp -o r -- python3 tools/micro/shard_rank_probe.py $2 8 3 8 $3 > /tmp/sp.log 2>&1
  echo "== $1  configs[$2] x $3"; grep "configs\[" /tmp/sp.log; python3 tools/rocprof_summary.py stats /tmp/sp/r_results.db | cut -c1-60,111-160 | head -46
}
run MVUS_SEP_TWO_LEVEL=1 3 1
run MVUS_SEP_TWO_LEVEL=0 3 1
run MVUS_SEP_TWO_LEVEL=1 2 8
