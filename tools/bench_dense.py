#!/usr/bin/env python
"""GPU micro-benchmark of the gather-GEMM on the DENSE launches: the 1x1 convs of the head / Up layers (F = 1) and the
level-0 / level-1 stencil convs on a dense-surface lattice (every tap present: nothing to skip).  One tile
configuration per process (HPL_TILE=... ; unset = the library's choice).  Prints us and TFLOP/s per shape."""
import os, sys, types
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import hplflownet_amd as H
from hplflownet_amd import ops
from hplflownet_amd.synthetic import SCALES_FILTER_MAP, surface_pair

dev = 'cuda'
pc1, pc2, sf = surface_pair(8192, 0)
gen = H.GenerateDataUnsymmetric(types.SimpleNamespace(dim=3, scales_filter_map=SCALES_FILTER_MAP), device=dev)
_, _, _, lat = gen([pc1, pc2, sf])
lat.prepare()
shapes = [('conv2 8192x1024x1024', None, 8192, 1024, 1024, 1), ('conv3 8192x1024x512', None, 8192, 1024, 512, 1),
          ('bcn2_ 1x1 25841x512x512', None, 25841, 512, 512, 1), ('surface bcn1_ blur', 0, None, 580, 1024, 15),
          ('surface bcn2_ blur', 1, None, 324, 512, 15), ('surface bcn3_ blur', 2, None, 388, 256, 15)]
reps = int(os.environ.get('REPS', '20'))
out = []
for name, lvl, M, C, O, F in shapes:
    nbr = perm = tiles = None
    if lvl is not None:
        up = lat.levels[lvl].blur[0]
        nbr, perm, tiles = up.t, up.perm, up.perm_tiles
        M = nbr.shape[1]
    A = torch.randn(M, C, device=dev)
    W = torch.randn(O, C, F, device=dev) / (C * F) ** 0.5
    Wt = ops.weight_relayout(W, C, O, F, F, C * F, 1)
    kw = dict(row_perm=perm, tiles=tiles) if perm is not None else {}
    y = ops.gconv_raw(A, nbr, M, C, F, Wt, O, **kw)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        ops.gconv_raw(A, nbr, M, C, F, Wt, O, out=y, **kw)
    e.record(); torch.cuda.synchronize()
    us = 1e3 * s.elapsed_time(e) / reps
    out.append('%s M=%d: %.1f us %.1f TF' % (name, M, us, 2.0 * M * F * C * O / us / 1e6))
print('HPL_TILE=%s | ' % os.environ.get('HPL_TILE', '-') + ' | '.join(out))
