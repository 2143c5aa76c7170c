#!/usr/bin/env python
"""Timeline of the LAST native training step in a rocprofv3 --kernel-trace db of tools/train_native_probe.py (PROBE_ONLY=native):
every dispatch from the step's k_weight_relayout_batch on -- offset, duration, queue (main / side stream), kernel -- and per queue the
busy time and the time only that queue was running.    python tools/train_timeline.py results.db > profiles/rNN_train_timeline.txt"""
import re
import sqlite3
import sys


def main():
    c = sqlite3.connect(sys.argv[1])
    cols = [r[1] for r in c.execute("pragma table_info(kernels)").fetchall()]
    qcol = 'queue_id' if 'queue_id' in cols else ('stream_id' if 'stream_id' in cols else None)
    rows = c.execute("select name, start, end, grid_x, workgroup_x, grid_y, workgroup_y%s from kernels order by start" % ((', ' + qcol) if qcol else '')).fetchall()
    firsts = [i for i, r in enumerate(rows) if 'k_weight_relayout_batch' in r[0]]
    # the last COMPLETE step: between the last two re-layouts that are followed by a k_epe3d
    starts = [i for i in firsts if any('k_epe3d' in r[0] for r in rows[i:i + 400])]
    starts = [i for k, i in enumerate(starts) if k == 0 or i - starts[k - 1] > 60]       # (a step re-lays its images in two launches)
    lo = starts[-2] if len(starts) >= 2 else starts[-1]
    hi = starts[-1] if len(starts) >= 2 else len(rows)
    sel = rows[lo:hi]
    t0 = sel[0][1]
    queues = {}
    print('# %d dispatches, span %.1f us' % (len(sel), (max(r[2] for r in sel) - t0) / 1e3))
    for r in sel:
        name, s, e, gx, wx, gy, wy = r[:7]
        q = r[7] if qcol else 0
        queues.setdefault(q, []).append((s, e))
        short = name.replace('(anonymous namespace)::', '').replace('void ', '')
        short = re.sub(r'\(.*$', '', short) if '<' not in short.split('(')[0] else short[:short.index('>') + 1]
        print('%9.1f  +%7.1f us  q%-3s wgs %6d  %s' % ((s - t0) / 1e3, (e - s) / 1e3, q, (gx // max(1, wx)) * max(1, gy // max(1, wy)), short[:70]))
    for q, iv in sorted(queues.items()):
        print('# queue %s: %d dispatches, busy %.1f us' % (q, len(iv), sum(e - s for s, e in iv) / 1e3))


if __name__ == '__main__':
    main()
