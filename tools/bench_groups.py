#!/usr/bin/env python
"""GPU: the wide stencil convs (bcn1_/bcn2_ blur) as one pass vs one pass per tap group."""
import os, sys, types
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import hplflownet_amd as H
from hplflownet_amd import ops
from hplflownet_amd.bcl import NbrTable
from hplflownet_amd.synthetic import SCALES_FILTER_MAP, synthetic_pair
import bench

dev = 'cuda'
pc1, pc2, sf = synthetic_pair(8192, 0)
gen = H.GenerateDataUnsymmetric(types.SimpleNamespace(dim=3, scales_filter_map=SCALES_FILTER_MAP), device=dev)
_, _, _, lat = gen([pc1, pc2, sf])
reps = int(os.environ.get('REPS', '10'))
for name, lvl, C, O in (('bcn1_ blur', 0, 580, 1024), ('bcn2_ blur', 1, 324, 512)):
    base = lat.levels[lvl].blur[0].t.contiguous()
    F, M = base.shape
    A = torch.randn(M, C, device=dev)
    W = (torch.randn(O, C, F, 1, device=dev) / (C * F) ** 0.5)
    for G in (1, 2, 3, 5):
        tbl = NbrTable(base)
        tbl.TAP_GROUPS = G
        groups = tbl.groups()
        y = torch.empty(M, O, device=dev)
        with torch.no_grad():
            fn = lambda: ops.gconv(A, W, None, tbl.t, M, F, row_perm=tbl.perm if not groups else None, tap_groups=groups, out=y,
                                   tiles=tbl.group_tiles() if groups else tbl.perm_tiles)
            fn(); torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(reps):
                fn()
            e.record(); torch.cuda.synchronize()
        ms = s.elapsed_time(e) / reps
        if groups:
            fr = sum((f1 - f0) * bench.needed_slice_fraction(types.SimpleNamespace(t=tbl.t[f0:f1], perm=p), C, BM=64)
                     for f0, f1, p in groups) / F
        else:
            fr = bench.needed_slice_fraction(tbl, C, BM=64)
        print('%-11s groups=%d  %7.3f ms  %6.1f TF algorithmic  executed slices %.3f -> %5.1f TF executed'
              % (name, G, ms, 2.0 * M * F * C * O / ms / 1e9, fr, 2.0 * M * F * C * O * fr / ms / 1e9))
