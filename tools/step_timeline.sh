#!/bin/bash
# usage (on the GPU box): tools/step_timeline.sh [config]  -> the ordered kernel list of ONE LM step (start offsets, durations, gaps)
export TMPDIR=/tmp
O=gpurun_out/timeline; mkdir -p $O
rocprofv3 --kernel-trace --memory-copy-trace -d $O/t -o r -- python3 tools/step_breakdown.py ${1:-2} > $O/run.log 2>&1
tail -1 $O/run.log
python3 - <<'PY'
import sqlite3, glob
db = glob.glob('gpurun_out/timeline/t/*results.db')[0]
cur = sqlite3.connect(db).cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
ev = []
kt = [t for t in tabs if t == 'kernels'] or [t for t in tabs if 'kernel' in t.lower()]
for row in cur.execute('select name, start, end from kernels'):
    ev.append((row[1], row[2], row[0]))
try:
    for row in cur.execute('select name, start, end from memory_copies'):
        ev.append((row[1], row[2], 'COPY ' + str(row[0])))
except Exception as e:
    print('no memory copy table:', e, [t for t in tabs if 'cop' in t.lower()])
ev.sort()
# the last complete step: from the last k_lm_trial backwards to the previous one
idx = [i for i, e in enumerate(ev) if 'k_lm_trial(' in e[2]]
a, b = idx[-2] + 1, idx[-1] + 1
t0 = ev[a][0]
prev_end = ev[a - 1][1]
tot = 0
for s, e, n in ev[a:b + 3]:
    print('%9.1f us  +%6.1f gap  %7.1f us  %s' % ((s - t0) / 1e3, (s - prev_end) / 1e3, (e - s) / 1e3, n[:90]))
    prev_end = max(prev_end, e); tot += e - s
print('busy %.1f us of %.1f us' % (tot / 1e3, (ev[b + 2][1] - t0) / 1e3))
PY
rm -rf $O/t
