#!/bin/bash
# per-kernel averages of LM steps (tools/micro/host_probe.py) for library builds: tools/micro/ab_stats.sh <kernel-name-pattern> libA.so libB.so ...
export TMPDIR=/tmp
pat=$1; shift
for lib in "$@"; do
  rm -rf /tmp/abst; MVUS_LIB_PATH=$PWD/$lib rocprofv3 --kernel-trace --stats -d /tmp/abst -o r -- python3 tools/micro/host_probe.py > /tmp/abst.log 2>&1
  echo "== $lib  $(grep 'python wall' /tmp/abst.log)"; python3 tools/rocprof_summary.py stats /tmp/abst/r_results.db | grep -E "$pat" | cut -c1-60,111-160
done
