#!/usr/bin/env python
"""Executed MFMA work per kernel from a rocprofv3 --pmc run of bench.py (rocpd sqlite .db).

    rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE \
        -d gpurun_out/pmc_mfma -o m -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-overlap
    python tools/pmc_mfma.py gpurun_out/pmc_mfma/m_results.db profiles/r02_mfma_pmc

writes <prefix>.txt (per-kernel table) and <prefix>.json (what bench.py reads: executed GFLOP per launch of the
dominant gather-GEMM).  The unit of SQ_INSTS_VALU_MFMA_MOPS_F32 is calibrated inside the same run on
`k_mfma_probe`, whose flop count is known exactly (bench.mfma_ceiling: 1024 blocks x 4 waves x 1500 x 64
v_mfma_f32_32x32x2_f32 of 4096 flop)."""
import json
import sqlite3
import sys

PROBE_FLOP = 1024 * 4.0 * 1500 * 64 * 4096
import os
import re
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
# the wide tap-group passes: the split-operand kernel (default) or, with HPL_MATH=f32, the fp32-MFMA 64 x 128 stencil class
DOMINANT = re.compile(r'k_gconv3w<8, 4, \d(, false)?>' if os.environ.get('HPL_MATH', 'f16x2') != 'f32' else r'k_gconv<64, 128, 2, 4, true, (8|15)\b')


def main():
    db, prefix = sys.argv[1], sys.argv[2]
    c = sqlite3.connect(db)
    rows = c.execute("select kernel_name, counter_name, count(*), avg(value), sum(value) from counters_collection "
                     "group by 1, 2").fetchall()
    per = {}
    for name, ctr, n, avg, tot in rows:
        per.setdefault(name, {})[ctr] = (n, avg, tot)
    probe = [v for k, v in per.items() if 'k_mfma_probe' in k and 'chain' not in k]
    unit = None
    if probe and 'SQ_INSTS_VALU_MFMA_MOPS_F32' in probe[0]:
        unit = PROBE_FLOP / probe[0]['SQ_INSTS_VALU_MFMA_MOPS_F32'][1]
    unit_used = unit if unit else 512.0
    lines = ['# rocprofv3 --pmc: per-kernel averages per launch; flop per MOPS count calibrated on k_mfma_probe: %s'
             % ('%.1f' % unit if unit else 'n/a (512 assumed)')]
    ctrs = sorted({ctr for v in per.values() for ctr in v})
    lines.append('%-92s %7s ' % ('kernel', 'calls') + ' '.join('%26s' % x for x in ctrs) + ' %14s' % 'GFLOP_executed')
    order = sorted(per.items(), key=lambda kv: -kv[1].get('SQ_INSTS_VALU_MFMA_MOPS_F32', (0, 0, 0))[2])
    out = {'flop_per_mops_count': unit_used, 'calibrated': bool(unit), 'kernels': {}}
    for name, v in order:
        n = max(x[0] for x in v.values())
        mops = v.get('SQ_INSTS_VALU_MFMA_MOPS_F32', (0, 0.0, 0.0))[1]
        gf = mops * unit_used / 1e9
        lines.append('%-92s %7d ' % (name[:92], n) + ' '.join('%26.1f' % v.get(x, (0, 0.0, 0.0))[1] for x in ctrs)
                     + ' %14.3f' % gf)
        if gf > 0 or v.get('SQ_VALU_MFMA_BUSY_CYCLES', (0, 0.0, 0.0))[1] > 0:
            out['kernels'][name] = {'launches': n, 'executed_gflop_per_launch': gf,
                                    **{x: v[x][1] for x in v}}
        if DOMINANT.search(name):
            out['dominant_kernel'] = name
            out['dominant_executed_gflop_per_launch'] = gf
            out['dominant_launches'] = n
            busy = v.get('SQ_VALU_MFMA_BUSY_CYCLES')
            if busy:
                # matrix-pipe cycles per launch, summed over the SIMDs: 64 per v_mfma_f32_32x32x2_f32, 32 per
                # v_mfma_f32_32x32x16_bf16 -- the dtype-independent measure bench.py prices against 1024 SIMDs x 2.4 GHz
                out['dominant_busy_cycles_per_launch'] = busy[1]
            gui = v.get('GRBM_GUI_ACTIVE')
            if busy and gui and gui[1] > 0:
                # busy cycles are summed over the SIMDs that report (per-XCD sampling): quote the ratio only
                out['dominant_mfma_busy_cycles_per_gui_cycle'] = busy[1] / gui[1]
    if out.get('dominant_launches'):
        # whole step: every MFMA kernel's executed flops, per step.  Forwards in the run = launches of the one kernel
        # instance that runs exactly once per forward (the first conv1 layer: C = 3, not vectorisable)
        once = [k['launches'] for n_, k in out['kernels'].items() if re.search(r'k_gconv<64, 32, 2, 1, false, 1\b', n_)]
        steps = float(once[0]) if once else out['dominant_launches'] / 4.0
        out['dominant_launches_per_step'] = out['dominant_launches'] / steps
        out['busy_cycles_per_step'] = sum(v_.get('SQ_VALU_MFMA_BUSY_CYCLES', (0, 0.0, 0.0))[2] for n_, v_ in per.items()
                                          if 'k_mfma_probe' not in n_) / steps
        out['executed_gflop_per_step'] = sum(k['launches'] * k['executed_gflop_per_launch'] for n_, k in out['kernels'].items()
                                             if 'k_mfma_probe' not in n_) / steps
        lines.append('# executed MFMA work of the whole step (all kernels): %.1f GF' % out['executed_gflop_per_step'])
    import bench
    out['stamp'] = bench.source_stamp()
    open(prefix + '.txt', 'w').write('\n'.join(lines) + '\n')
    json.dump(out, open(prefix + '.json', 'w'), indent=1)
    print('\n'.join(lines[:14]))


if __name__ == '__main__':
    main()
