#!/usr/bin/env python
"""Launch-by-launch timeline of the LAST single-stream step of tools/chain_run.py from a rocprofv3 --kernel-trace db:
    python tools/forward_trace.py results.db > profiles/rNN_forward_timeline.txt
offset, duration, idle gap in front, grid (workgroups) and kernel of every dispatch behind the last k_fused_begin."""
import re
import sqlite3
import sys


def main():
    c = sqlite3.connect(sys.argv[1])
    rows = c.execute("select name, start, end, grid_x, workgroup_x, grid_y, workgroup_y from kernels order by start").fetchall()
    firsts = [i for i, r in enumerate(rows) if 'k_fused_begin' in r[0] or 'k_lattice_keys_pair' in r[0]]
    sel = rows[firsts[-1]:]
    t0, prev = sel[0][1], sel[0][1]
    busy = gap_sum = 0.0
    print('# %d dispatches, span %.1f us' % (len(sel), (sel[-1][2] - t0) / 1e3))
    for name, s, e, gx, wx, gy, wy in sel:
        short = name.replace('(anonymous namespace)::', '').replace('void ', '')
        short = re.sub(r'\(.*$', '', short) if '<' not in short.split('(')[0] else short[:short.index('>') + 1]
        gap = max(0, s - prev) / 1e3
        prev = max(prev, e)
        busy += (e - s) / 1e3
        gap_sum += gap
        print('%9.1f  +%7.1f us  gap %5.1f  wgs %6d  %s' % ((s - t0) / 1e3, (e - s) / 1e3, gap, (gx // max(1, wx)) * max(1, gy // max(1, wy)), short[:70]))
    print('# busy %.1f us, gaps %.1f us' % (busy, gap_sum))


if __name__ == '__main__':
    main()
