#!/usr/bin/env python
"""Kernel-trace durations of the HBM-bound gathers inside a real forward (what the review asked `kernels.*.frac` to be quoted
from, instead of bench.py's back-to-back replay on one buffer set, which stays in the 256-MiB Infinity Cache):
    rocprofv3 --kernel-trace -d out -o ft -- python tools/chain_run.py frustum 8192
    python tools/trace_hbm.py out/.../ft_results.db profiles/trace_hbm.json
Per class of bench.py's `kernels` table -- splat / slice (the three launches of levels 0-2: the three with the largest grids),
splat_deep / slice_deep (the rest) -- the mean duration of a launch over the last three single-stream steps of the trace, stamped
with the kernel sources' hash (bench.source_stamp): bench.py uses the file only when the stamp is this tree's."""
import json
import os
import sqlite3
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    db, out = sys.argv[1:3]
    c = sqlite3.connect(db)
    rows = c.execute("select name, start, end, grid_x, workgroup_x from kernels order by start").fetchall()
    firsts = [i for i, r in enumerate(rows) if 'k_fused_begin' in r[0]]
    steps = [rows[a:b] for a, b in zip(firsts, firsts[1:] + [len(rows)])][-3:]
    res = {}
    for kern, big, deep in (('k_splat', 'splat', 'splat_deep'), ('k_slice', 'slice', 'slice_deep')):
        acc = {big: [], deep: []}
        for st in steps:
            ls = [((e - s) / 1e3, gx // max(1, wx)) for n, s, e, gx, wx in st if kern + '<' in n]
            if len(ls) < 4:
                continue
            order = sorted(range(len(ls)), key=lambda i: -ls[i][1])[:3]
            acc[big] += [ls[i][0] for i in order]
            acc[deep] += [ls[i][0] for i in range(len(ls)) if i not in order]
        for k, v in acc.items():
            if v:
                res[k] = {'us_per_launch': sum(v) / len(v), 'launches_per_step': len(v) / float(len(steps)), 'min_us': min(v), 'max_us': max(v)}
    import bench
    d = {'_comment': 'rocprofv3 --kernel-trace of tools/chain_run.py frustum 8192 (single-stream lattice build + native forward), last three '
                     'steps: mean kernel duration per launch of the splat / slice classes of bench.py (tools/trace_hbm.py)',
         'classes': res, 'stamp': bench.source_stamp()}
    with open(out, 'w') as f:
        json.dump(d, f, indent=1)
    print(json.dumps(res))


if __name__ == '__main__':
    main()
