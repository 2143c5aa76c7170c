#!/usr/bin/env python
"""GPU: bench.py over cloud sizes (full HPLFlowNet inference incl. the lattice build) + the shallow model at N = 4096 (BASELINE config 2)
+ dense-surface pairs: one table for profiles/rNN_point_count_sweep.txt."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
def run(extra):
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--steps', '60', '--warmup', '5', '--no-cpu-baseline', '--no-train-probe'] + extra,
                         stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True).stdout
    return json.loads([l for l in out.splitlines() if l.startswith('{')][0])
print('# python bench.py --points N --steps 60 --warmup 5 --no-cpu-baseline --no-train-probe   (one MI355X; pipelined output checked against the single-stream forward)')
print('%-28s %9s %8s %14s %12s %12s %8s %6s' % ('workload', 'pairs/s', 'ms/step', 'fwd-only p/s', 'lattice ms', 'forward ms', 'board W', 'check'))
for name, extra in [('HPLFlowNet N=%d' % n, ['--points', str(n)]) for n in (500, 1024, 2048, 4096, 8192, 16384, 32768)] + \
        [('HPLFlowNetShallow N=4096', ['--arch', 'HPLFlowNetShallow', '--points', '4096']), ('HPLFlowNet N=8192 surface', ['--data', 'surface'])]:
    d = run(extra)
    lat = d.get('single_pair_latency_ms') or {}
    print('%-28s %9.1f %8.3f %14.1f %12.3f %12.3f %8s %6s' % (name, d['value'], d['ms_per_step'], (d.get('forward_only') or {}).get('pairs_per_s', 0), lat.get('lattice_build_ms', 0),
          lat.get('forward_ms', 0), (d.get('power') or {}).get('package_w'), (d.get('pipelined_output_check') or {}).get('max_abs_diff')))
    sys.stdout.flush()
