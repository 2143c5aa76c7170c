#!/usr/bin/env python
"""Weight gradient of the wide layers on the real tables of the N=8192 frustum: fp32-MFMA kernel vs split operands on the
bf16 MFMA (csrc/wgrad3.hip).  One process per mode (HPL_WGRAD3 is read once); TF = executed fp32-equivalent flops / time."""
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
t1 = torch.from_numpy(pc1.T.copy()).to(dev); t2 = torch.from_numpy(pc2.T.copy()).to(dev)
lat = gen.build(t1, t2)
reps = int(os.environ.get('REPS', '5'))
cases = [('bcn1_ blur', 0, 580, 1024), ('bcn2_ blur', 1, 324, 512), ('dense 25841x1024x1024', -1, 1024, 1024),
         ('dense 34631x512x512', -2, 512, 512), ('dense 8192x1024x1024', -3, 1024, 1024)]
for name, lvl, C, O in cases:
    if lvl >= 0:
        nbr = lat.levels[lvl].blur[0].t
        F, M = nbr.shape
        taps = ops.tap_lists(nbr)
        valid = float((nbr >= 0).float().mean())
    else:
        nbr, taps, F, valid = None, None, 1, 1.0
        M = {-1: 25841, -2: 34631, -3: 8192}[lvl]
    A = torch.randn(M, C, device=dev)
    dY = torch.randn(M, O, device=dev)
    fn = lambda: ops.wgrad_raw(A, nbr, M, C, F, dY, O, taps=taps)
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(reps):
            fn()
        e.record(); torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) / reps)
    fl = 2.0 * M * F * C * O * valid
    print('%-24s M=%6d C=%5d F=%2d N=%5d taps %.2f | %8.1f us %6.1f TF (incl. the zero fill of dWt)' % (name, M, C, F, O, valid, best * 1e3, fl / best / 1e9), flush=True)
