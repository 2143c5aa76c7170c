#!/usr/bin/env python
"""Cycle accounting of the persistent gather-GEMM (timing build): per workgroup total / main loops / staging /
epilogue+switch cycles, slices, clock."""
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
pc1, pc2, sf = synthetic_pair(8192, 0)
gen = H.GenerateDataUnsymmetric(types.SimpleNamespace(dim=3, scales_filter_map=SCALES_FILTER_MAP), device=dev)
_, _, _, lat = gen([pc1, pc2, sf])
for name, lvl, C, O in (('bcn1_ blur', 0, 580, 1024), ('bcn2_ blur', 1, 324, 512)):
    base = lat.levels[lvl].blur[0].t.contiguous()
    tbl = NbrTable(base); tbl.vertices_per_point = 3.0
    f0, f1, perm = tbl.groups()[0]
    nbr = base[f0:f1]; Fg = f1 - f0; M = base.shape[1]
    tiles = tbl.group_tiles()[0]
    A = torch.randn(M, C, device=dev)
    W = torch.randn(O, C, Fg, device=dev) / (C * Fg) ** 0.5
    Wt = ops.weight_relayout(W, C, O, Fg, Fg, C * Fg, 1)
    y = torch.empty(M, O, device=dev)
    for _ in range(3):
        ops.gconv_raw(A, nbr, M, C, Fg, Wt, O, out=y, row_perm=perm, tiles=tiles)
    torch.cuda.synchronize()
    assert L.hpl_timing_reset() == 0
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); ops.gconv_raw(A, nbr, M, C, Fg, Wt, O, out=y, row_perm=perm, tiles=tiles); e.record(); torch.cuda.synchronize()
    buf = np.zeros(8192 * 64, np.int64)
    assert L.hpl_timing_read(buf.ctypes.data_as(ctypes.c_void_p)) == 0
    r = buf.reshape(8192, 8, 8)
    n = int((r[:, 0, 3] > 0).sum()); r = r[:n, 0, :].astype(np.float64)
    w0 = r[:, 6]; w1 = (r[:, 7].astype(np.int64) >> 4).astype(np.float64)
    base = w0.min()
    for x in range(8):
        sel = (np.arange(n) & 7) == x
        print('   queue %d: start %.1f..%.1f us, end %.1f..%.1f us, slices/wg %.0f (sum %d), tiles/wg %.1f, cyc/slice %.0f' %
              (x, (w0[sel].min() - base) / 100, (w0[sel].max() - base) / 100, (w1[sel].min() - base) / 100, (w1[sel].max() - base) / 100,
               r[sel, 4].mean(), r[sel, 4].sum(), (r[sel, 7].astype(np.int64) & 15).mean(), r[sel, 1].sum() / r[sel, 4].sum()))
    total = r[:, 3] - r[:, 0]
    wall = ((r[:, 7].astype(np.int64) >> 4) - r[:, 6]).astype(np.float64)
    print('%s %.3f ms, %d workgroups: total %.0f kcyc (min %.0f max %.0f), loop %.1f %% stage %.1f %% epilogue+switch %.1f %%, slices %.0f (min %.0f max %.0f), cycles/slice %.0f, clock %.2f GHz, wall min %.0f max %.0f us'
          % (name, s.elapsed_time(e), n, total.mean() / 1e3, total.min() / 1e3, total.max() / 1e3, 100 * r[:, 1].sum() / total.sum(),
             100 * r[:, 2].sum() / total.sum(), 100 * r[:, 5].sum() / total.sum(), r[:, 4].mean(), r[:, 4].min(), r[:, 4].max(),
             r[:, 1].sum() / r[:, 4].sum(), np.median(total / wall) * 0.1, wall.min() / 100, wall.max() / 100))
