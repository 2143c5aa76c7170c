#!/usr/bin/env python
"""Where the wall time of the four dominant launches goes BETWEEN the tiles (csrc/gconv3.hip built with -DHPL_PHASE_PROBE=2, see
tools/gpu/tile_probe.sh): every workgroup of k_gconv3w<8,4> leaves its entry / exit wall time, the shader cycles of its main loop,
its slice count and the CU it ran on.  Printed per launch: launch time, CU-slot occupancy over the launch (sum of workgroup
residence / (256 x span)), the share of residence outside the main loop (prologue + epilogue), shader cycles per half-step,
when the CUs go idle at the end, and what a perfectly balanced schedule of the same tiles would take."""
import os, sys, types, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import hplflownet_amd as H
from hplflownet_amd import ops
from hplflownet_amd.synthetic import SCALES_FILTER_MAP, synthetic_pair
dev = 'cuda'
N = int(os.environ.get('POINTS', 8192))
pc1, pc2, sf = synthetic_pair(N, 0)
gen = H.GenerateDataUnsymmetric(types.SimpleNamespace(dim=3, scales_filter_map=SCALES_FILTER_MAP), device=dev)
t1 = torch.from_numpy(pc1.T.copy()).to(dev); t2 = torch.from_numpy(pc2.T.copy()).to(dev)
lat = gen.build(t1, t2)
for name, lvl, C, O in [('bcn1_', 0, 580, 1024), ('bcn2_', 1, 324, 512)]:
    tb = lat.levels[lvl].blur[0]
    tb.vertices_per_point = 3.0
    groups, gtiles = tb.groups(), tb.group_tiles()
    M = tb.t.shape[1]
    A = torch.randn(M, C, device=dev)
    y = torch.empty(M, O, device=dev)
    for gi, ((f0, f1, perm), tiles) in enumerate(zip(groups, gtiles)):
        F = f1 - f0
        nbr = tb.t[f0:f1]
        Wt = torch.zeros(ops.round_up(F * C, 32), O, device=dev); Wt[:F * C] = torch.randn(F * C, O, device=dev) / (F * C) ** 0.5
        W3 = ops.weight_split3(Wt)
        fn = lambda: ops.gconv_raw(A, nbr, M, C, F, Wt, O, out=y, row_perm=perm, tiles=tiles, split_k=False, Wt3=W3)
        for _ in range(3): fn()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record(); torch.cuda.synchronize()
        us = 1e3 * s.elapsed_time(e)
        probe = torch.zeros(64 + 4 * 4096, dtype=torch.int64, device=dev)
        ops.CLOCK_PROBE = probe
        fn(); torch.cuda.synchronize()
        ops.CLOCK_PROBE = None
        r = probe.cpu().numpy()[64:].reshape(-1, 4)
        r = r[r[:, 1] > 0]
        w0, w1, loop = r[:, 0].astype(np.float64), r[:, 1].astype(np.float64), r[:, 2].astype(np.float64)
        nsl = (r[:, 3] & 0xffff).astype(np.float64)
        hw = (r[:, 3] >> 16) & 0xffffffff
        xcc = (r[:, 3] >> 48) & 15
        cu = (xcc << 8) | (((hw >> 13) & 7) << 5) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 15)      # XCC | SE | SH | CU
        t0 = w0.min(); span = w1.max() - t0
        res = w1 - w0
        cus = np.unique(cu)
        ok = nsl > 0
        clk = np.median(loop[ok] / np.maximum(1.0, (res[ok]))) / 10.0       # cycles per 10 ns tick -> GHz (loop ~ residence)
        hs = 2 * nsl[ok]
        cyc = loop[ok] / hs
        loop_ticks = loop / (clk * 10.0)
        print('%s group %d (F=%d, M=%d, C=%d, O=%d): %.1f us alone; %d workgroups with work on %d CUs' % (name, gi, F, M, C, O, us, len(r), len(cus)))
        print('   span of the probed launch %.1f us; CU-slot occupancy %.3f; main loop / residence %.3f; half-steps %d, cycles per half-step '
              'median %.0f mean %.0f (1 536 = MFMA issue only); clock ~%.2f GHz'
              % (span / 100, res.sum() / (len(cus) * span), loop_ticks.sum() / res.sum(), hs.sum(), np.median(cyc), (loop[ok].sum() / hs.sum()), clk))
        last = np.array([w1[cu == c].max() - t0 for c in cus])
        busy = np.array([res[cu == c].sum() for c in cus])
        print('   per CU: tiles %.2f mean (min %d max %d); busy / span mean %.3f min %.3f; last exit at %.2f of the span on average (p10 %.2f, min %.2f)'
              % (len(r) / len(cus), min((cu == c).sum() for c in cus), max((cu == c).sum() for c in cus), (busy / span).mean(), (busy / span).min(),
                 (last / span).mean(), np.percentile(last / span, 10), (last / span).min()))
        print('   balanced schedule of the same residences: %.1f us (= sum / CUs) -> the launch loses %.1f %% to the tail; heaviest tile %.1f us, lightest %.1f us'
              % (res.sum() / len(cus) / 100, 100 * (1 - res.sum() / (len(cus) * span)), res.max() / 100, res.min() / 100))
        print('   per XCC (workgroups, sum of residences / 32 CUs in us, last exit in us): ' +
              '  '.join('%d: %d %.0f %.0f' % (x, (xcc == x).sum(), res[xcc == x].sum() / 3200, (w1[xcc == x].max() - t0) / 100) for x in range(8)))
        # what a greedy queue would reach with the same residences: every CU takes the next tile of ONE heaviest-first list
        # (by slice count) the moment it is free; "per XCC" = eight such lists (the tiles each XCC ran), 32 CUs each
        import heapq
        def greedy(rs, n):
            h = [0.0] * n
            for x in rs:
                heapq.heappush(h, heapq.heappop(h) + x)
            return max(h)
        order = np.argsort(-nsl, kind='stable')
        g_all = greedy(res[order], len(cus))
        g_xcc = max(greedy(res[order][xcc[order] == x], 32) for x in range(8))
        print('   greedy queue on the same residences: one list %.1f us, one list per XCC %.1f us (measured span %.1f, balanced %.1f)'
              % (g_all / 100, g_xcc / 100, span / 100, res.sum() / len(cus) / 100))
        if os.environ.get('TP_SAVE'):
            np.savez(os.path.join(os.environ['TP_SAVE'], 'tile_probe_%s_g%d.npz' % (name, gi)), w0=w0, w1=w1, loop=loop, nsl=nsl, xcc=xcc, cu=cu)
        if os.environ.get('TP_DUMP'):
            sel = np.where(xcc == 0)[0]
            sel = sel[np.argsort(w0[sel], kind='stable')]
            print('   XCC 0 in entry order (entry us, SE.SH.CU, residence us): ' +
                  ' | '.join('%.0f %d.%d.%d %.0f' % ((w0[i] - t0) / 100, (hw[i] >> 13) & 7, (hw[i] >> 12) & 1, (hw[i] >> 8) & 15, res[i] / 100) for i in sel))
        bins = np.linspace(0, span, 21)
        occ = [np.clip(np.minimum(w1 - t0, b) - np.maximum(w0 - t0, a), 0, None).sum() / (b - a) for a, b in zip(bins[:-1], bins[1:])]
        print('   resident workgroups per 5 %% of the span: ' + ' '.join('%3.0f' % o for o in occ))
