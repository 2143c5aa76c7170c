#!/usr/bin/env python
"""GPU: the pc2 half of the patch correlation (models/bnn_flow.py:195-202) two ways on a real N-point lattice, levels 2 and 3:
  (a) gather-GEMM over the 15*H1 virtual vertices (K = 15 taps x 64 channels, N = 32): what rounds 1-4 ran;
  (b) per-tap projection Z = f2 . [W_0 | ... | W_14] (dense GEMM, 64 -> 480) + hpl_gather_sum of 32-float rows;
and their gradients: (a) GEMM + atomic scatter of 960-float rows, weight gradient through the table; (b) hpl_gather_sum of the
output gradient through the inverse table (hpl_table_invert) into dZ, then two small dense GEMMs.  Prints times (HIP events, 20 repetitions) and the largest difference of the results."""
import os, sys, types, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import hplflownet_amd as H
from hplflownet_amd import ops, _lib
from hplflownet_amd._lib import check, ptr, stream
from hplflownet_amd.synthetic import SCALES_FILTER_MAP, synthetic_pair
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
dev = 'cuda'
a = types.SimpleNamespace(dim=3, scales_filter_map=SCALES_FILTER_MAP, evaluate=False, use_leaky=True, bcn_use_bias=True,
                          bcn_use_norm=True, last_relu=False, DEVICE='cuda')
gen = H.GenerateDataUnsymmetric(a, device=dev)
pc1, pc2, sf = synthetic_pair(N, 0)
t1, t2 = [torch.from_numpy(x.T.copy()).to(dev) for x in (pc1, pc2)]
lat = gen.build(t1, t2)
L = _lib.load()

def timed(fn, n=20):
    for _ in range(3): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return 1e3 * s.elapsed_time(e) / n

torch.manual_seed(0)
for lev in (2, 3):
    lv = lat.levels[lev]
    H0, H1 = lv.H
    tbl = lv.corr2.t                       # [15, 15*H0]
    K, M = tbl.shape
    C, O = 64, 32
    f2 = torch.randn(H1, C, device=dev)
    W = torch.randn(O, C, K, device=dev) * 0.05          # (o, c, k)
    bias = torch.randn(O, device=dev)
    res = torch.randn(H0, O, device=dev)
    Wt = ops.weight_relayout(W, C, O, K, K, C * K, 1)    # [(k*C + c), o]
    Wz = W.permute(1, 2, 0).reshape(C, K * O).contiguous()          # [c, k*O + o]
    Wz_img = ops.weight_relayout(Wz, C, K * O, 1, K * O, 1, 1)
    ya = torch.empty(M, O, device=dev); yb = torch.empty(M, O, device=dev)
    def fa():
        ops.gconv_raw(f2, tbl, M, C, K, Wt, O, bias=bias, act=1, res=res, res_mod=H0, out=ya)
    Z = torch.empty(H1, K * O, device=dev)
    def fb1():
        ops.gconv_raw(f2, None, H1, C, 1, Wz_img, K * O, out=Z)
    def fb2():
        ops.gather_sum_raw(Z, tbl, M, K, O, O, bias=bias, res=res, res_mod=H0, act=1, slope=0.1, out=yb)
    ta, tb1, tb2 = timed(fa), timed(fb1), timed(fb2)
    err = float((ya - yb).abs().max()) / float(ya.abs().max())
    print('level %d  H0 %d H1 %d  M %d: forward gather-GEMM %.1f us | projection GEMM %.1f + gather-sum %.1f us   rel diff %.2e' % (lev, H0, H1, M, ta, tb1, tb2, err))
    # backward
    g = torch.randn(M, O, device=dev)
    WtS = ops.weight_relayout(W.permute(0, 2, 1).reshape(O, K * C).contiguous(), O, K * C, 1, K * C, 1, 1)
    gA = torch.zeros(H1, C, device=dev)
    def ba1():
        gA.zero_()
        ops.gconv_raw(g, None, M, O, 1, WtS, K * C, out=gA, scat=tbl, scat_c=C)
    def ba2():
        ops.wgrad_raw(f2, tbl, M, C, K, g, O)
    dZ = torch.zeros(H1, K * O, device=dev)
    F = M // H0
    inv = [None]
    def bb0():
        inv[0] = ops.table_invert(tbl, H0, F, H1)
    def bb1():
        ops.gather_sum_raw(g, inv[0], K * H1, F, O, 0, out=dZ.view(K * H1, O))
    WzT = ops.weight_relayout(Wz.t().contiguous(), K * O, C, 1, C, 1, 1)
    gB = torch.empty(H1, C, device=dev)
    def bb2():
        ops.gconv_raw(dZ, None, H1, K * O, 1, WzT, C, out=gB)
    def bb3():
        ops.wgrad_raw(f2, None, H1, C, 1, dZ, K * O)
    t0 = timed(bb0)
    t = [timed(f) for f in (ba1, ba2, bb1, bb2, bb3)]
    ba1(); bb1(); bb2()
    err = float((gA - gB).abs().max()) / float(gA.abs().max())
    print('          backward: GEMM+scatter %.1f us, table wgrad %.1f us | invert %.1f + gather-sum %.1f + dZ GEMM %.1f + dense wgrad %.1f us   rel diff %.2e'
          % (t[0], t[1], t0, t[2], t[3], t[4], err))
