#!/usr/bin/env python
"""Where a single-stream step's time goes, launch by launch: from a rocprofv3 --kernel-trace db of
    python tools/chain_run.py [frustum|surface]
take the LAST step (from its first k_init_ws / lattice kernel to its last kernel) and print, per kernel class:
launches, busy time, and the idle gap in front of each launch (dependent-launch latency of the in-order stream).
Usage: python tools/chain_trace.py results.db [out.txt] [levels=7]
"""
import re
import sqlite3
import sys


def main():
    c = sqlite3.connect(sys.argv[1])
    rows = c.execute("select name, start, end from kernels order by start").fetchall()
    # steps begin with the lattice build's first kernel (k_minmax of level 0 follows k_init_ws)
    firsts = [i for i, r in enumerate(rows) if 'k_lattice_keys_pair' in r[0]]
    nlev = int(sys.argv[3]) if len(sys.argv) > 3 else 7
    lo, hi = firsts[-nlev], len(rows)                          # the last step (tools/chain_run.py ends with it)
    # walk back to the build's first kernel (init / minmax precede the keys kernel)
    while lo > 0 and any(k in rows[lo - 1][0] for k in ('k_init_ws', 'k_minmax', 'k_zero_i32')):
        lo -= 1
    step = rows[lo:hi]
    span = (step[-1][2] - step[0][1]) / 1e3
    per = {}
    prev_end = step[0][1]
    lat_n = lat_busy = lat_gap = 0
    fwd_n = fwd_busy = fwd_gap = 0
    fwd_names = ('k_gconv', 'k_splat', 'k_slice', 'k_copy_cols', 'k_gconv_finish', 'k_transpose', 'k_level')
    gaps = []
    for name, s, e in step:
        short = name.replace('(anonymous namespace)::', '')
        short = short[5:] if short.startswith('void ') else short
        short = re.sub(r'\(.*$', '', short) if '<' not in short.split('(')[0] else short[:short.index('>') + 1]
        gap = max(0, s - prev_end) / 1e3
        prev_end = max(prev_end, e)
        d = per.setdefault(short, [0, 0.0, 0.0])
        d[0] += 1
        d[1] += (e - s) / 1e3
        d[2] += gap
        gaps.append(gap)
        if any(short.startswith(f) for f in fwd_names):
            fwd_n, fwd_busy, fwd_gap = fwd_n + 1, fwd_busy + (e - s) / 1e3, fwd_gap + gap
        else:
            lat_n, lat_busy, lat_gap = lat_n + 1, lat_busy + (e - s) / 1e3, lat_gap + gap
    out = ['# one single-stream step (lattice build + forward): %d launches, span %.1f us' % (len(step), span),
           '# lattice build: %d launches, busy %.1f us, idle in front of them %.1f us' % (lat_n, lat_busy, lat_gap),
           '# forward:       %d launches, busy %.1f us, idle in front of them %.1f us' % (fwd_n, fwd_busy, fwd_gap),
           '# gap per launch: median %.2f us, mean %.2f us' % (sorted(gaps)[len(gaps) // 2], sum(gaps) / len(gaps)),
           '%-60s %6s %10s %10s %10s' % ('kernel', 'calls', 'busy_us', 'avg_us', 'gap_us')]
    for k, (n, b, g) in sorted(per.items(), key=lambda kv: -(kv[1][1] + kv[1][2])):
        out.append('%-60s %6d %10.1f %10.2f %10.1f' % (k[:60], n, b, b / n, g))
    txt = '\n'.join(out) + '\n'
    if len(sys.argv) > 2:
        open(sys.argv[2], 'w').write(txt)
    print(txt)


if __name__ == '__main__':
    main()
