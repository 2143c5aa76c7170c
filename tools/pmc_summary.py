#!/usr/bin/env python
"""Per-kernel averages of the counters in a rocprofv3 --pmc run (rocpd sqlite .db).
    python tools/pmc_summary.py gpurun_out/pmc_fetch/f_results.db [name-substring]
Prints: kernel, counter, launches, average value per launch."""
import sqlite3
import sys


def main():
    db = sys.argv[1]
    like = sys.argv[2] if len(sys.argv) > 2 else ''
    c = sqlite3.connect(db)
    rows = c.execute("select kernel_name, counter_name, count(*), avg(value), sum(value) from counters_collection "
                     "where kernel_name like ? group by 1, 2 order by 5 desc", ('%' + like + '%',)).fetchall()
    for name, ctr, n, avg, tot in rows[:40]:
        print('%-90s %-12s %6d %16.1f' % (name[:90], ctr, n, avg))


if __name__ == '__main__':
    main()
