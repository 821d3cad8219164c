export TMPDIR=/tmp
for tl in 1 0; do
  rm -rf /tmp/sp; MVUS_SEP_TWO_LEVEL=$tl rocprofv3 --kernel-trace --stats -d /tmp/sp -o r -- python3 tools/micro/shard_rank_probe.py 3 8 3 8 > /tmp/sp.log 2>&1
  echo "== MVUS_SEP_TWO_LEVEL=$tl"; grep "configs\[" /tmp/sp.log; python3 tools/rocprof_summary.py stats /tmp/sp/r_results.db | cut -c1-60,111-160 | head -45
done
