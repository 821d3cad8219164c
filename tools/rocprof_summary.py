#!/usr/bin/env python3
"""Summarise rocprofv3 sqlite outputs (kernel stats / one PMC counter) into the text kept under profiles/.

    python tools/rocprof_summary.py stats gpurun_out/prof_stats/r1_results.db  > profiles/r01_kernel_stats.txt
    python tools/rocprof_summary.py pmc   gpurun_out/prof_fetch/r1_results.db  > profiles/r01_pmc_FETCH_SIZE.txt
"""
import sqlite3
import sys


def stats(db):
    cur = sqlite3.connect(db).cursor()
    rows = list(cur.execute('select name, total_calls, total_duration, average, percentage from top_kernels'))
    print('# rocprofv3 --kernel-trace --stats  (durations in microseconds)')
    print('%-110s %8s %14s %12s %8s' % ('kernel', 'calls', 'total_us', 'avg_us', 'pct'))
    for name, calls, total, avg, pct in rows:
        print('%-110s %8d %14.3f %12.3f %8.2f' % (name[:110], calls, total, avg, pct))


def pmc(db):
    cur = sqlite3.connect(db).cursor()
    q = ('select kernel_name, counter_name, count(*), avg(value), min(value), max(value), avg(duration) '
         'from counters_collection group by kernel_name, counter_name order by sum(duration) desc')
    print('# rocprofv3 --pmc <counter> --kernel-trace ; value as reported by rocprofv3 (FETCH_SIZE/WRITE_SIZE in KiB),')
    print('# per dispatch; duration in ns (profiled run, clocks differ from the un-profiled bench)')
    print('%-100s %-12s %6s %14s %14s %14s %12s' % ('kernel', 'counter', 'n', 'avg', 'min', 'max', 'avg_dur_ns'))
    for name, cname, n, avg, mn, mx, dur in cur.execute(q):
        print('%-100s %-12s %6d %14.3f %14.3f %14.3f %12.0f' % (name[:100], cname, n, avg, mn, mx, dur))


def traffic(fetch_txt, write_txt, key='config2_calib0', merge_into=None):
    """profiles/pmc_traffic.json: HBM bytes per launch of the residual+Jacobian kernel from the two PMC summaries, corrected as
    /opt/skills/guides/MI355X_MICROARCH.md (HBM section) prescribes for gfx950: FETCH_SIZE doubled (128-B read requests are
    tallied at 64 B), WRITE_SIZE as reported; both are KiB per dispatch in the summaries.  `key` names the workload
    (config<i>_calib<0|1>, what bench.py looks up); `merge_into`: an existing json whose other keys are kept."""
    import json
    import os
    kname = 'k_observations<true, true>' if key.endswith('calib1') else 'k_observations<false, true>'

    def avg(path, counter):
        for line in open(path):
            if kname in line and counter in line:
                parts = line.split(counter)[1].split()
                return int(parts[0]), float(parts[1]) * 1024.0
        raise SystemExit('kernel not found in ' + path)
    nf, fb = avg(fetch_txt, 'FETCH_SIZE')
    nw, wb = avg(write_txt, 'WRITE_SIZE')
    out = {}
    if merge_into and os.path.exists(merge_into):
        out = json.load(open(merge_into))
    out[key] = {
        'kernel': kname, 'dispatches': nw,
        'WRITE_SIZE_bytes': wb, 'FETCH_SIZE_bytes_reported': fb, 'bytes_per_launch': wb + 2.0 * fb,
        'note': 'rocprofv3 --pmc WRITE_SIZE and --pmc FETCH_SIZE in separate passes of `python3 bench.py` (%s, %s); '
                'the dispatches are the 100 launches with rotating outputs (>= 1 GiB in rotation) plus 20 into one buffer set; FETCH_SIZE '
                'doubled per the gfx950 correction of the guide, WRITE_SIZE as reported' % (os.path.basename(fetch_txt), os.path.basename(write_txt))}
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    {'stats': stats, 'pmc': pmc, 'traffic': traffic}[sys.argv[1]](*sys.argv[2:])
