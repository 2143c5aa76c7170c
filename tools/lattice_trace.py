#!/usr/bin/env python
"""Per-launch timeline of one lattice build (run under rocprofv3 --kernel-trace):
    python tools/lattice_trace.py run [N]            builds 6 lattices of an N-point frustum pair (the last two are the sample)
    python tools/lattice_trace.py show results.db    start offset / duration / grid of every dispatch of the last build
"""
import os
import sqlite3
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run(n):
    import types
    import torch
    import hplflownet_amd as H
    from hplflownet_amd.synthetic import SCALES_FILTER_MAP, synthetic_pair
    shallow = bool(os.environ.get('LT_SHALLOW'))        # the 5-level model (BASELINE config 2)
    args = types.SimpleNamespace(dim=3, scales_filter_map=SCALES_FILTER_MAP[:5] if shallow else SCALES_FILTER_MAP, evaluate=True, use_leaky=True,
                                 bcn_use_bias=True, bcn_use_norm=True, last_relu=False, DEVICE='cuda')
    m = (H.HPLFlowNetShallow if shallow else H.HPLFlowNet)(args)
    gen = H.GenerateDataUnsymmetric(args, device='cuda', wide_up=m.lattice_hint())
    pc1, pc2, _ = synthetic_pair(n, 0)
    t1, t2 = torch.from_numpy(pc1.T.copy()).cuda(), torch.from_numpy(pc2.T.copy()).cuda()
    import time
    for i in range(int(os.environ.get('LT_BUILDS', 6))):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        lat = gen.build_native(t1, t2)
        torch.cuda.synchronize()
        print('build %d: %.3f ms  H=%s' % (i, (time.perf_counter() - t0) * 1e3, [h[0] for h in lat.H]))
    nb = gen.native_builder()
    print('fused', nb.fused, 'launches', getattr(nb, 'launches', None), 'bounds', nb.bounds, 'arena MB', lat.arena.numel() / 2 ** 20)


def show(db):
    c = sqlite3.connect(db)
    rows = c.execute("select name, start, end, grid_x, workgroup_x from kernels order by start").fetchall()
    # the last build = the dispatches after the last gap > 200 us
    cut = 0
    for i in range(1, len(rows)):
        if rows[i][1] - rows[i - 1][2] > 200e3:
            cut = i
    sel = rows[cut:]
    t0 = sel[0][1]
    busy = 0.0
    print('# last build: %d dispatches, span %.1f us' % (len(sel), (sel[-1][2] - t0) / 1e3))
    for name, s, e, g, w in sel:
        busy += (e - s) / 1e3
        print('%9.1f us  +%7.1f us  grid %6d  %s' % ((s - t0) / 1e3, (e - s) / 1e3, g // max(1, w), name[:60]))
    print('# busy %.1f us' % busy)


if __name__ == '__main__':
    if sys.argv[1] == 'run':
        run(int(sys.argv[2]) if len(sys.argv) > 2 else 8192)
    else:
        show(sys.argv[2])
