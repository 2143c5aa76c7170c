#!/usr/bin/env python
"""Per kernel, how long a launch of the pipelined loop waits behind its predecessor ON ITS OWN QUEUE before it starts, from a rocprofv3
--kernel-trace db of bench.py:  python tools/queue_gaps.py results.db [lo hi] > profiles/rNN_queue_gaps.txt
(lo, hi: the window of the trace in fractions of its span, default 0.35 0.65 -- the pipelined region of a short bench run).  A launch
whose workgroups fit beside other pairs' wide tiles (few registers, little LDS) starts as soon as its predecessor has drained; one that
needs a CU of its own waits for a tile of another stream to end.  Columns: launches, mean / median gap in front, mean duration."""
import re
import sqlite3
import statistics
import sys


def main():
    c = sqlite3.connect(sys.argv[1])
    lo, hi = (float(sys.argv[2]), float(sys.argv[3])) if len(sys.argv) > 3 else (0.35, 0.65)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)").fetchall()]
    qcol = 'queue_id' if 'queue_id' in cols else 'stream_id'
    rows = c.execute("select name, start, end, %s from kernels order by start" % qcol).fetchall()
    t0, t1 = rows[0][1], max(r[2] for r in rows)
    a, b = t0 + lo * (t1 - t0), t0 + hi * (t1 - t0)
    last = {}
    agg = {}
    for name, s, e, q in rows:
        prev = last.get(q)
        last[q] = e if prev is None else max(prev, e)
        if prev is None or not (a <= s <= b):
            continue
        short = name.replace('(anonymous namespace)::', '').replace('void ', '')
        short = re.sub(r'\(.*$', '', short) if '<' not in short.split('(')[0] else short[:short.index('>') + 1]
        agg.setdefault(short, []).append((max(0, s - prev) / 1e3, (e - s) / 1e3))
    print('# window %.2f-%.2f of a %.1f ms trace; gap = start - end of the previous launch on the same queue (%s)' % (lo, hi, (t1 - t0) / 1e6, qcol))
    print('%-64s %7s %10s %10s %10s %12s' % ('kernel', 'calls', 'gap_mean', 'gap_median', 'dur_mean', 'gap_total_ms'))
    for k, v in sorted(agg.items(), key=lambda kv: -sum(g for g, _ in kv[1])):
        gaps = [g for g, _ in v]
        print('%-64s %7d %10.1f %10.1f %10.1f %12.2f' % (k[:64], len(v), statistics.mean(gaps), statistics.median(gaps),
                                                        statistics.mean(d for _, d in v), sum(gaps) / 1e3))


if __name__ == '__main__':
    main()
