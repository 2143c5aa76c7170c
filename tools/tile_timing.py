#!/usr/bin/env python
"""Where the cycles of the dominant gather-GEMM go (diagnostic library with cycle stamps: tools/build_timing_lib.sh,
run with HPL_LIB=hplflownet_amd/libhplbcl_timing.so).  One launch of the bcn1_ / bcn2_ blur conv (first tap group,
the model's row order); per workgroup: prologue / main loop / epilogue cycles, cycles per executed slice, share of
the loop parked at the end-of-step wait + barrier."""
import ctypes, os, sys, types
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import hplflownet_amd as H
from hplflownet_amd import _lib, ops
from hplflownet_amd.bcl import NbrTable
from hplflownet_amd.synthetic import SCALES_FILTER_MAP, synthetic_pair

dev = 'cuda'
L = _lib.load()
if not hasattr(L, 'hpl_timing_read'):
    sys.exit('needs the timing build: HPL_LIB=hplflownet_amd/libhplbcl_timing.so')
pc1, pc2, sf = synthetic_pair(8192, 0)
gen = H.GenerateDataUnsymmetric(types.SimpleNamespace(dim=3, scales_filter_map=SCALES_FILTER_MAP), device=dev)
_, _, _, lat = gen([pc1, pc2, sf])
NW = 8192
if os.environ.get('ABL'):
    assert L.hpl_timing_ablate(int(os.environ['ABL'])) == 0
for name, lvl, C, O in (('bcn1_ blur', 0, 580, 1024), ('bcn2_ blur', 1, 324, 512), ('dense', -1, 8704, 1024)):
    if lvl >= 0:
        base = lat.levels[lvl].blur[0].t.contiguous()
        F, M = base.shape
        tbl = NbrTable(base)
        tbl.vertices_per_point = 3.0
        f0, f1, perm = tbl.groups()[0]
        nbr = base[f0:f1]
        Fg = f1 - f0
        tiles = None if os.environ.get('HPL_NO_TILES') else tbl.group_tiles()[0]
    else:
        nbr, perm, Fg, M, tiles = None, None, 1, 25841, None
    A = torch.randn(M, C, device=dev)
    W = torch.randn(O, C, Fg, device=dev) / (C * Fg) ** 0.5
    Wt = ops.weight_relayout(W, C, O, Fg, Fg, C * Fg, 1)
    y = torch.empty(M, O, device=dev)
    for _ in range(3):
        ops.gconv_raw(A, nbr, M, C, Fg, Wt, O, out=y, row_perm=perm, tiles=tiles)
    torch.cuda.synchronize()
    assert L.hpl_timing_reset() == 0
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    ops.gconv_raw(A, nbr, M, C, Fg, Wt, O, out=y, row_perm=perm, tiles=tiles)
    e.record()
    torch.cuda.synchronize()
    ms = s.elapsed_time(e)
    buf = np.zeros(NW * 8 * 8, np.int64)
    assert L.hpl_timing_read(buf.ctypes.data_as(ctypes.c_void_p)) == 0
    r = buf.reshape(NW, 8, 8)
    nwg = int((r[:, 0, 3] > 0).sum())
    r = r[:nwg]
    w0 = r[:, 0, :]
    total = (r[:, :, 3].max(1) - r[:, :, 0].min(1)).astype(np.float64)
    pro = (w0[:, 1] - w0[:, 0]).astype(np.float64)
    loop = (w0[:, 2] - w0[:, 1]).astype(np.float64)
    epi = (w0[:, 3] - w0[:, 2]).astype(np.float64)
    w_start = r[:, :, 6].min(1).astype(np.float64)
    w_end = (r[:, :, 7] >> 4).max(1).astype(np.float64)
    xcc = (w0[:, 7] & 15)
    nsl = w0[:, 4].astype(np.float64)
    parked = r[:, :, 5].mean(1).astype(np.float64)
    ok = nsl > 0
    per_slice = loop[ok] / nsl[ok]
    span = float(r[:, :, 3].max() - r[:, :, 0].min())
    print('%-11s %.3f ms  %d workgroups' % (name, ms, nwg))
    print('   per workgroup (median / mean): total %.0f / %.0f  prologue %.0f / %.0f  loop %.0f / %.0f  epilogue %.0f / %.0f cycles'
          % (np.median(total), total.mean(), np.median(pro), pro.mean(), np.median(loop), loop.mean(), np.median(epi), epi.mean()))
    t0w = w_start.min()
    wall_us = (w_end.max() - t0w) / 100.0
    clk = total / np.maximum(1.0, (w_end - w_start)) * 100e6 / 1e9
    print('   wall span %.1f us; shader clock from cycle / wall stamps: median %.2f GHz (p10 %.2f p90 %.2f)' %
          (wall_us, np.median(clk), np.percentile(clk, 10), np.percentile(clk, 90)))
    # occupancy timeline: resident workgroups over 20 equal time bins
    bins = np.linspace(0, w_end.max() - t0w, 21)
    occ = []
    for a_, b_ in zip(bins[:-1], bins[1:]):
        ov = np.clip(np.minimum(w_end - t0w, b_) - np.maximum(w_start - t0w, a_), 0, None).sum() / (b_ - a_)
        occ.append(ov)
    print('   resident workgroups per 5 %% of the span: ' + ' '.join('%3.0f' % o for o in occ))
    print('   mean residency %.1f of 512 slots = %.3f' % (np.mean(occ), np.mean(occ) / 512))
    for x in range(8):
        sel = xcc == x
        if sel.any():
            print('   XCC %d: %4d workgroups, last end %.1f us, busy slot-time %.1f us' %
                  (x, sel.sum(), (w_end[sel].max() - t0w) / 100.0, (w_end[sel] - w_start[sel]).sum() / 100.0 / 64))
    order = np.argsort(w0[:, 0])
    first, later = order[:512], order[512:]
    print('   first 512 workgroups: prologue mean %.0f; later ones: %.0f; epilogue first %.0f later %.0f' %
          (pro[first].mean(), pro[later].mean() if len(later) else 0, epi[first].mean(), epi[later].mean() if len(later) else 0))
    print('   slices per workgroup: median %.0f mean %.1f; cycles per slice: median %.0f mean %.0f p10 %.0f p90 %.0f '
          '(4 waves/SIMD x 16 MFMA x 64 = 4096 when the matrix pipe is the limit)' %
          (np.median(nsl), nsl.mean(), np.median(per_slice), per_slice.mean(), np.percentile(per_slice, 10), np.percentile(per_slice, 90)))
    print('   share of workgroup time: prologue %.1f %%  loop %.1f %%  epilogue %.1f %%; of the loop parked at wait+barrier: %.1f %%'
          % (100 * pro.sum() / total.sum(), 100 * loop.sum() / total.sum(), 100 * epi.sum() / total.sum(),
             100 * parked[ok].sum() / loop[ok].sum()))
    # occupancy over the kernel: workgroup-cycles / (span * slots)
