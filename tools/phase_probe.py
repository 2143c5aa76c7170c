#!/usr/bin/env python
"""Where a half-step of the ping-pong split-operand kernel spends its cycles (csrc/gconv3.hip built with -DHPL_PHASE_PROBE=1, see
tools/gpu/phase_probe.sh): per wave row and half-step, shader cycles in the memory phase (and the part of it spent issuing), at the
first barrier, in the compute phase, at the second barrier.  Timing only -- the stamps are scalar memory reads and add waits."""
import os, sys, types, torch
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
for name, lvl, C, O, f0, f1 in [('bcn1_ g0', 0, 580, 1024, 0, 8), ('dense', -1, 4640, 1024, 0, 1)]:
    F = f1 - f0
    if lvl >= 0:
        tb = lat.levels[lvl].blur[0]
        nbr = tb.t[f0:f1]; M = nbr.shape[1]
        perm = ops.tap_order(nbr, tb.keys); t128 = ops.tile_index(nbr, perm, BM=128)
    else:
        nbr, M, perm, t128 = None, 25841, None, None
    A = torch.randn(M, C, device=dev)
    Wt = torch.zeros(ops.round_up(F * C, 32), O, device=dev); Wt[:F * C] = torch.randn(F * C, O, device=dev) / (F * C) ** 0.5
    W3 = ops.weight_split3(Wt)
    y = torch.empty(M, O, device=dev)
    fn = lambda: ops.gconv_raw(A, nbr, M, C, F, Wt, O, out=y, row_perm=perm, tiles=t128, split_k=False, Wt3=W3)
    fn(); torch.cuda.synchronize()
    probe = torch.zeros(32, dtype=torch.int64, device=dev)
    ops.CLOCK_PROBE = probe
    fn(); torch.cuda.synchronize()
    ops.CLOCK_PROBE = None
    v = probe.cpu().tolist()
    for g in range(2):
        hs = max(1, v[16 + g]) * 4.0       # half-steps x 4 waves of the row
        ph = [v[8 + 4 * g + k] / hs for k in range(4)]; iss = v[20 + g] / hs
        print('%-10s wave row %d: issue %5.0f | memory phase %6.0f | barrier 1 %6.0f | compute phase %6.0f | barrier 2 %6.0f | sum %6.0f cycles (%d half-steps sampled)'
              % (name, g, iss, ph[0], ph[1], ph[2], ph[3], sum(ph), v[16 + g]))
