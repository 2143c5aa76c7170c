#!/usr/bin/env python
"""Summarise a rocprofv3 --kernel-trace --stats run (rocpd sqlite .db) into a text table:
per-kernel launch count, total / average / min / max duration.  Usage:
    python tools/prof_summary.py gpurun_out/prof1/r01_results.db "command line" > profiles/rNN_kernel_stats.txt
"""
import sqlite3
import sys


def main():
    db, cmd = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else '')
    c = sqlite3.connect(db)
    rows = c.execute("select name, count(*), sum(end-start)/1e3, avg(end-start)/1e3, min(end-start)/1e3, "
                     "max(end-start)/1e3 from kernels group by name order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows)
    span = c.execute("select min(start), max(end) from kernels").fetchone()
    print('# rocprofv3 --kernel-trace --stats -- %s' % cmd)
    print('# kernels: %d launches, %.3f ms busy, %.3f ms first-to-last span' % (sum(r[1] for r in rows), tot / 1e3,
                                                                              (span[1] - span[0]) / 1e6))
    print('%-100s %7s %12s %10s %9s %10s %6s' % ('kernel', 'calls', 'total_us', 'avg_us', 'min_us', 'max_us', '%'))
    for r in rows:
        print('%-100s %7d %12.1f %10.2f %9.2f %10.2f %6.2f' % (r[0][:100], r[1], r[2], r[3], r[4], r[5], 100 * r[2] / tot))


if __name__ == '__main__':
    main()
