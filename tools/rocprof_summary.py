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


if __name__ == '__main__':
    {'stats': stats, 'pmc': pmc}[sys.argv[1]](sys.argv[2])
