#!/bin/bash
# usage (GPU box): tools/micro/shard_rank_timeline.sh [cfg=2] [mult=8]  -> the kernels and copies of ONE solve + trial + linearisation of rank 3 of 8
# (shard_rank_probe.py: alone on the GPU, no-op sums) in launch order, with start offsets, durations and gaps
export TMPDIR=/tmp
rm -rf /tmp/st; rocprofv3 --kernel-trace --memory-copy-trace -d /tmp/st -o r -- python3 tools/micro/shard_rank_probe.py ${1:-2} 8 3 4 ${2:-8} > /tmp/st.log 2>&1
grep "configs\[" /tmp/st.log | cut -c1-200
python3 - <<'PY'
import sqlite3, glob
cur = sqlite3.connect(glob.glob('/tmp/st/*results.db')[0]).cursor()
ev = [(r[1], r[2], r[0]) for r in cur.execute('select name, start, end from kernels')]
try:
    ev += [(r[1], r[2], 'COPY ' + str(r[0])) for r in cur.execute('select name, start, end from memory_copies')]
except Exception as e:
    print('no memory copy table:', e)
ev.sort()
idx = [i for i, e in enumerate(ev) if 'k_band_pack' in e[2]]
a, b = idx[-3], idx[-2]
t0, prev_end, tot = ev[a][0], ev[a - 1][1], 0
for s, e, n in ev[a:b]:
    print('%9.1f us  +%6.1f gap  %7.1f us  %s' % ((s - t0) / 1e3, (s - prev_end) / 1e3, (e - s) / 1e3, n[:70]))
    prev_end = max(prev_end, e); tot += e - s
print('busy %.1f us of %.1f us, %d launches' % (tot / 1e3, (ev[b][0] - t0) / 1e3, b - a))
PY
