#!/usr/bin/env python
"""Regenerate the marked regions of DESIGN.md (<!--KEY-->...<!--/KEY-->) from the round's profile files (profiles/rNN_*):
    python tools/design_fill.py r06
Every figure DESIGN.md quotes in those places is read from a committed file; run it again after a new profile round."""
import json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
R = sys.argv[1] if len(sys.argv) > 1 else 'r06'
P = lambda n: os.path.join(ROOT, 'profiles', '%s_%s' % (R, n))
J = lambda n: json.load(open(P(n)))


def main():
    d, pl = J('bench_driver_cmd_detail.json'), J('bench_plain_detail.json')
    r, k = d['roofline'], d['kernels']
    lat = d['single_pair_latency_ms']
    tr = d.get('train') or {}
    pw = d.get('power') or {}
    runs = ''
    if os.path.exists(P('driver_cmd_runs.txt')):          # tools/gpu/driver_cmd_runs.sh: five runs on one box, the median run is the record
        m = re.search(r'median run: \d+ \(([\d.]+) pairs/s\); min ([\d.]+), max ([\d.]+)', open(P('driver_cmd_runs.txt')).read())
        if m:
            runs = ' (the median of five runs of that command on one box: %s–%s, `%s_driver_cmd_runs.txt`)' % (m.group(2), m.group(3), R)
    head = ('**%.1f point-pairs/s** over the driver\'s 20 steps%s (%.3f ms per step), %.1f over the ≥ 1 s behind them (`steady`), %.1f over 200 steps '
            '(`python bench.py`, `%s_bench_plain.json`; steady %.1f), forward alone %.1f pairs/s; one pair alone: lattice %.2f ms + forward %.2f ms; '
            'exact bf16 triples %.1f pairs/s; training step %.2f ms; CPU port %.3f pairs/s on %d threads (× %.0f); EPE3D differs from the CPU oracle '
            'by %.1e; board %s W of %s at %s MHz.  Boxes of the pool differ by ± 5 %% (this file\'s A/Bs: 435–478 pairs/s for the same '
            'build over 300 steps); round 5\'s driver record: 428.9.'
            % (d['value'], runs, d['ms_per_step'], d['steady']['value'], pl['value'], R, pl['steady']['value'], d['forward_only']['pairs_per_s'],
               lat['lattice_build_ms'], lat['forward_ms'], d['exact_bf16x3']['value'], tr.get('ms_per_step', float('nan')),
               d['cpu_baseline']['value'], d['cpu_baseline']['cores'], d['value'] / d['cpu_baseline']['value'], d['epe3d']['abs_delta'],
               pw.get('package_w'), pw.get('limit_w'), pw.get('sclk_mhz')))
    rows = ['| class (launches per step) | isolated launch µs | back-to-back µs | bound | achieved | frac |', '|---|---|---|---|---|---|']
    names = {'gconv3_128x256_g': 'wide stencil, tap-group pass `k_gconv3w<8,4,2>` (dominant)', 'gconv3_128x256_g_mid': 'wide stencil, one pass, split-K `k_gconv3w<15,4,2>`',
             'gconv3_128x256_d': 'wide dense 1×1 `k_gconv3w<1,4,2>`', 'gconv3_128x256_d_mid': 'wide dense 1×1, mid-size'}
    for n in sorted(k, key=lambda n: -k[n].get('launches_per_step', 0) * k[n].get('avg_launch_us', 0)):
        v = k[n]
        unit = 'TF/s' if v.get('bound') == 'mfma' else 'GB/s'
        rows.append('| %s (%g) | %.1f | %.1f | %s | %.0f %s | %.3f |' % (names.get(n, '`%s`' % n), v.get('launches_per_step', 0), v.get('avg_launch_us', 0),
                                                                       v.get('kernel_us', 0), v.get('bound'), v.get('achieved', 0), unit, v.get('frac', 0)))
    ws = r.get('whole_step') or {}
    il = r.get('in_loop') or {}
    rows.append('')
    rows.append('`roofline` of the line: %s, frac **%.3f** of the fp16 matrix pipe (%.0f of 2 516.6 TF/s executed; %s), isolated launch %.1f µs; inside the timed '
                'loop %.1f µs (frac %.3f); whole step %s; rocprofv3 average of the same kernel over the default bench run: see `%s_kernel_stats.txt`.'
                % (r.get('kernel', 'k_gconv3w<8,4,2>')[:40], r['frac'], r['achieved'], r.get('executed_source', '')[:60], r['avg_launch_us'], il.get('avg_launch_us', float('nan')),
                   il.get('frac', float('nan')), ('frac %.3f' % ws['frac']) if ws.get('frac') else 'n/a', R))
    t = r.get('traffic')
    tb = t.get('bytes_per_launch') if isinstance(t, dict) else t
    traffic = ('%.0f MB per launch (`profiles/pmc_traffic.json`: FETCH_SIZE × 2 + WRITE_SIZE, two `--pmc` passes) against ≈ 200 MB algorithmic (A once + weights + Y); '
               'round 5, column-major order: 930 MB.' % (tb / 1e6)) if isinstance(tb, (int, float)) else 'not collected (`traffic` null)'
    th = json.load(open(P('trace_hbm.json'))).get('classes', {})
    ss = []
    for n in ('splat', 'slice', 'splat_deep', 'slice_deep'):
        v, tv = k.get(n, {}), th.get(n, {})
        ss.append('%s: %.1f µs per launch in the forward trace (%.1f–%.1f), `trace_frac` %s; back to back %.1f µs, frac %.3f'
                  % (n, tv.get('us_per_launch', 0), tv.get('min_us', 0), tv.get('max_us', 0), ('%.3f' % v['trace_frac']) if 'trace_frac' in v else 'n/a',
                     v.get('kernel_us', 0), v.get('frac', 0)))
    avg = None
    for ln in open(P('kernel_stats.txt')):
        if 'k_gconv3w<8, 4, 2' in ln and 'true>' not in ln:
            avg = float(ln[100:].split()[2])
            break
    dom = ('rocprofv3 average of `k_gconv3w<8,4,2>` over the default bench run %.1f µs (round 5: 351.8), fabric traffic %s per launch (930), %.1f pairs/s over the '
           "driver's 20 steps, %.1f over ≥ 1 s" % (avg, ('%.0f MB' % (tb / 1e6)) if isinstance(tb, (int, float)) else 'n/a', d['value'], d['steady']['value']))
    text = open(os.path.join(ROOT, 'DESIGN.md')).read()
    gp = open(P('pytest_gpu.txt')).read()
    m = re.search(r'(\d+) passed', gp)
    for key, val in (('HEADLINE', head), ('KERNEL_TABLE', '\n'.join(rows)), ('TRAFFIC', traffic), ('SPLAT_SLICE', ';\n'.join(ss) + '.'),
                     ('LATTICE_MS', '%.2f' % lat['lattice_build_ms']), ('NGPU', m.group(1) if m else '?'), ('DOM', dom)):
        text, n = re.subn(r'<!--%s-->.*?<!--/%s-->' % (key, key), lambda _m: '<!--%s-->%s<!--/%s-->' % (key, val, key), text, flags=re.S)
        assert n >= 1, key
    open(os.path.join(ROOT, 'DESIGN.md'), 'w').write(text)
    print(head)
    print('\n'.join(rows))
    print(traffic)
    print('\n'.join(ss))


if __name__ == '__main__':
    main()
