#!/usr/bin/env python
"""HBM-side traffic of the dominant gather-GEMM from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; rocpd .db):
    python tools/pmc_traffic.py gpurun_out/pmc_fetch/f_results.db gpurun_out/pmc_write/w_results.db profiles/pmc_traffic.json
Per MI355X_MICROARCH.md (HBM section): the counters are in KiB; on gfx950 FETCH_SIZE reports half the bytes of wide
coalesced reads, so the read side is doubled; WRITE_SIZE is taken as is."""
import json
import sqlite3
import sys

import os
import re
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
DOMINANT = re.compile(r'k_gconv3w<8, 4, \d(, false)?>' if os.environ.get('HPL_MATH', 'f16x2') != 'f32' else r'k_gconv<64, 128, 2, 4, true, (8|15)\b')


def per_launch(db, ctr):
    c = sqlite3.connect(db)
    rows = c.execute("select kernel_name, count(*), avg(value) from counters_collection where counter_name=? group by 1",
                     (ctr,)).fetchall()
    return {k: (n, v) for k, n, v in rows}


def main():
    fdb, wdb, out = sys.argv[1:4]
    f, w = per_launch(fdb, 'FETCH_SIZE'), per_launch(wdb, 'WRITE_SIZE')
    fk = max((k for k in f if DOMINANT.search(k)), key=lambda k: f[k][0] * f[k][1])
    wk = max((k for k in w if DOMINANT.search(k)), key=lambda k: w[k][0] * w[k][1])
    fetch_kib, write_kib = f[fk][1], w[wk][1]
    total = int((2 * fetch_kib + write_kib) * 1024)
    try:
        old = json.load(open(out))
    except Exception:
        old = {}
    hist = old.get('history', {})
    hist.pop('round_1_final', None)
    hist.setdefault('round_2_fp32_kernel_64x128_tiles', 1549000000)
    hist.setdefault('round_3_split_kernel_one_barrier_form', 1110000000)
    hist.setdefault('round_5_pair_form_column_major_xcd_order', 930231343)
    d = {'_comment': 'rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (two separate passes) of `python bench.py '
                     '--steps 3 --warmup 1 --no-cpu-baseline --no-overlap` (tools/pmc_traffic.py). Per launch of '
                     'the dominant stencil instance (k_gconv3w<8,4,planes>; HPL_MATH=f32: k_gconv<64,128,2,4,true,...>), averaged over its four launches per step. Counter unit KiB; FETCH_SIZE '
                     'doubled per MI355X_MICROARCH.md (gfx950 reports half the bytes of 16-B/lane reads), WRITE_SIZE as is.',
         'fetch_size_kib_per_launch_raw': fetch_kib, 'write_size_kib_per_launch': write_kib, 'launches_fetch_pass': f[fk][0],
         'k_gconv_64x128_bytes_per_launch': total, 'dominant_bytes_per_launch': total, 'dominant_kernel': fk,
         # every operand once: activation matrix + split weight image of the pass + output written (+ read back by the second
         # tap-group pass): bcn1_ 60 + 28.5 + 106 / 60 + 25 + 212 MB, bcn2_ 45 + 8 + 71 / 45 + 7 + 142 MB -> mean of the four launches
         'algorithmic_bytes_per_launch': 202000000, 'history': hist,
         'note': 'fetch > algorithmic: L2-miss traffic of the gathered activation rows (every use of a row by another tap is a slice list apart in time) '
                 'from the 60 / 45 MB matrix, which lives in the 256 MB Infinity Cache, and of the weight panels (round 6: row-major XCD '
                 'order -- the column tiles of a tile-row share their gathered rows in one L2, every L2 streams all weight panels; rounds 2-5: '
                 'column-major). DESIGN.md section 4.1.'}
    for nm in ('k_splat', 'k_slice'):       # the HBM-bound gathers: average over all their launches of a step (all levels)
        fk2 = [k for k in f if nm in k]
        wk2 = [k for k in w if nm in k]
        if fk2 and wk2:
            d['%s_bytes_per_launch_all_levels' % nm] = int((2 * f[fk2[0]][1] + w[wk2[0]][1]) * 1024)
            d['%s_launches' % nm] = f[fk2[0]][0]
    import bench
    d['stamp'] = bench.source_stamp()
    json.dump(d, open(out, 'w'), indent=1)
    print(json.dumps({k: d[k] for k in ('fetch_size_kib_per_launch_raw', 'write_size_kib_per_launch', 'k_gconv_64x128_bytes_per_launch')}))
    for name, tab in (('FETCH_SIZE', f), ('WRITE_SIZE', w)):
        for k, (n, v) in sorted(tab.items(), key=lambda kv: -kv[1][0] * kv[1][1])[:8]:
            print('%-90s %-11s %6d %14.1f' % (k[:90], name, n, v))


if __name__ == '__main__':
    main()
