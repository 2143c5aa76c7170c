#!/usr/bin/env python
"""A/B of the two gather-GEMM kernels on the wide Up-conv launches of the N=8192 frustum (real tables of the device
lattice): fp32 MFMA (csrc/gconv.hip) vs three-way bf16 split on the bf16 MFMA (csrc/gconv3.hip).  Interleaved rounds in
one process; TF = algorithmic fp32-equivalent flops / time."""
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
rounds = int(os.environ.get('ROUNDS', '3'))
cases = [('dgrad bcn1_ g0', 0, 1024, 580, 0, 8), ('dgrad bcn2_ g0', 1, 512, 324, 0, 8), ('bcn1_ g0', 0, 580, 1024, 0, 8), ('bcn1_ g1', 0, 580, 1024, 8, 15), ('bcn2_ g0', 1, 324, 512, 0, 8),
         ('bcn2_ g1', 1, 324, 512, 8, 15), ('dense 25841x4640x1024', -1, 4640, 1024, 0, 1), ('1x1 25841x1024x1024', -1, 1024, 1024, 0, 1)]


def timeit(fn):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps


only = os.environ.get('CASES')
if only:
    cases = [c for c in cases if any(o in c[0] for o in only.split(','))]
for name, lvl, C, O, f0, f1 in cases:
    F = f1 - f0
    if lvl >= 0:
        tb = lat.levels[lvl].blur[0]
        nbr = tb.t[f0:f1]
        M = nbr.shape[1]
        perm = ops.tap_order(nbr)
        t64 = ops.tile_index(nbr, perm, BM=64)
        t128 = ops.tile_index(nbr, perm, BM=128)
        valid = float((nbr >= 0).float().mean())
    else:
        nbr, M, perm, t64, t128, valid = None, 25841, None, None, None, 1.0
    A = torch.randn(M, C, device=dev)
    Wt = torch.zeros(ops.round_up(F * C, 32), O, device=dev)
    Wt[:F * C] = torch.randn(F * C, O, device=dev) / (F * C) ** 0.5
    W3 = ops.weight_split3(Wt)
    y = torch.empty(M, O, device=dev)
    f32 = lambda: ops.gconv_raw(A, nbr, M, C, F, Wt, O, out=y, row_perm=perm, tiles=t64, split_k=False)
    sp3 = lambda: ops.gconv_raw(A, nbr, M, C, F, Wt, O, out=y, row_perm=perm, tiles=t128, split_k=False, Wt3=W3)
    res = {'f32': [], 'split3': []}
    for _ in range(rounds):
        res['f32'].append(timeit(f32))
        res['split3'].append(timeit(sp3))
    fl = 2.0 * M * F * C * O
    a, b = min(res['f32']), min(res['split3'])
    print('%-24s M=%6d K=%5d N=%5d taps %.2f | fp32 %8.1f us %6.1f TF | split3 %8.1f us %6.1f TF | x%.2f' %
          (name, M, F * C, O, valid, a * 1e3, fl / a / 1e9, b * 1e3, fl / b / 1e9, a / b), flush=True)
