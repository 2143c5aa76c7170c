#!/usr/bin/env python
"""GPU micro-benchmark of the gather-GEMM on the Up-BCL shapes of the full model at N=8192
(real level tables from the device lattice).  Prints TFLOP/s per shape."""
import os, sys, types
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import hplflownet_amd as H
from hplflownet_amd import ops
from hplflownet_amd.synthetic import SCALES_FILTER_MAP, synthetic_pair

dev = 'cuda'
pc1, pc2, sf = synthetic_pair(8192, 0)
gen = H.GenerateDataUnsymmetric(types.SimpleNamespace(dim=3, scales_filter_map=SCALES_FILTER_MAP), device=dev)
_, _, _, lat = gen([pc1, pc2, sf])
shapes = [  # name, level, C_in, C_out, F
    ('bcn1_ blur', 0, 580, 1024, 15), ('bcn1_ 1x1', 0, 1024, 1024, 1),
    ('bcn2_ blur', 1, 324, 512, 15), ('bcn2_ 1x1', 1, 512, 512, 1),
    ('bcn3_ blur', 2, 388, 256, 15), ('bcn1 blur', 0, 68, 64, 15), ('bcn2 blur', 1, 68, 64, 15),
    ('conv2', -1, 1024, 1024, 1), ('conv3', -1, 1024, 512, 1),
    ('bcn4_ blur', 3, 260, 256, 15), ('corr1 B-term', -2, 64, 32, 15),
    ('dense longK', -1, 8704, 1024, 1),          # uniform tiles, no skipping: the loop's own efficiency
    ('pair bcn1 blur', -10, 68, 64, 15), ('pair bcn2 blur', -11, 68, 64, 15), ('pair bcn3 blur', -12, 68, 64, 15),
]
only = os.environ.get('SHAPES')          # comma-separated substrings
if only:
    shapes = [s_ for s_ in shapes if any(o in s_[0] for o in only.split(','))]
brief = bool(os.environ.get('BRIEF'))
reps = int(os.environ.get('REPS', '5'))
for name, lvl, C, O, F in shapes:
    if lvl >= 0:
        tbl = lat.levels[lvl].blur[0].t
        M = tbl.shape[1]
    elif lvl <= -10:                     # Down BCL of the stacked pair: level -10 - lvl, pair blur table
        tbl = lat.levels[-10 - lvl].blur.pair.t
        M = tbl.shape[1]
    elif lvl == -2:                      # corr B-term of level 2: 15*H1 virtual vertices, permuted corr2 table
        tbl = lat.levels[2].corr2.t
        M = tbl.shape[1]
    elif lvl == -1 and C > 4096:
        tbl, M = None, 25841
    else:
        tbl, M = None, 8192
    A = torch.randn(M if lvl != -2 else lat.levels[2].H[1], C, device=dev)
    W = torch.randn(O, C, F, device=dev) / (C * F) ** 0.5
    Wt = ops.weight_relayout(W, C, O, F, F, C * F, 1)
    nbr = tbl if F > 1 else None
    y = ops.gconv_raw(A, nbr, M, C, F, Wt, O)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        ops.gconv_raw(A, nbr, M, C, F, Wt, O, out=y)
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / reps
    fl = 2.0 * M * F * C * O
    valid = float((tbl >= 0).float().mean()) if nbr is not None else 1.0
    if brief and nbr is not None and M >= 16384:     # what the model launches: rows sorted by tap mask
        perm = ops.tap_order(nbr)
        ops.gconv_raw(A, nbr, M, C, F, Wt, O, out=y, row_perm=perm)
        torch.cuda.synchronize()
        s.record()
        for _ in range(reps):
            ops.gconv_raw(A, nbr, M, C, F, Wt, O, out=y, row_perm=perm)
        e.record(); torch.cuda.synchronize()
        ms = s.elapsed_time(e) / reps
    print('%-12s M=%6d K=%5d N=%5d  %8.3f ms  %6.1f TFLOP/s  (valid taps %.2f)' % (name, M, F * C, O, ms, fl / ms / 1e9, valid))
    if nbr is not None and not brief:
        perm = ops.tap_order(nbr)
        for nm, pm in (('  +row_perm(mask)', perm), ('  +row_perm(random)', torch.randperm(M, device=dev).to(torch.int32))):
            ops.gconv_raw(A, nbr, M, C, F, Wt, O, out=y, row_perm=pm)
            torch.cuda.synchronize()
            s.record()
            for _ in range(reps):
                ops.gconv_raw(A, nbr, M, C, F, Wt, O, out=y, row_perm=pm)
            e.record(); torch.cuda.synchronize()
            ms2 = s.elapsed_time(e) / reps
            print('%-20s %8.3f ms  %6.1f TFLOP/s (nominal)' % (nm, ms2, fl / ms2 / 1e9))
