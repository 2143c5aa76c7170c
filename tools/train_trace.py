#!/usr/bin/env python
"""GPU: every hot-kernel call of ONE training step (forward + backward, BASELINE config 4 shape: one N=8192 pair) in issue order,
with its shape and its time (HIP events around the call: small launches read ~8 us too long)."""
import os, sys, types, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import hplflownet_amd as H
from hplflownet_amd import ops
from hplflownet_amd.synthetic import SCALES_FILTER_MAP, fill_module_, synthetic_pair
dev = 'cuda'
a = types.SimpleNamespace(dim=3, scales_filter_map=SCALES_FILTER_MAP, evaluate=False, use_leaky=True, bcn_use_bias=True,
                          bcn_use_norm=True, last_relu=False, DEVICE='cuda')
model = H.HPLFlowNet(a); fill_module_(model, 1.0, 'hash'); model = model.to(dev).train()
model.native_forward = False
ops.enable_weight_bank(True)
gen = H.GenerateDataUnsymmetric(a, device=dev, wide_up=model.lattice_hint())
pc1, pc2, sf = synthetic_pair(8192, 0)
t1 = torch.from_numpy(pc1.T.copy()).to(dev); t2 = torch.from_numpy(pc2.T.copy()).to(dev); tsf = torch.from_numpy(sf.T.copy()).to(dev)
opt = torch.optim.Adam(model.parameters(), lr=1e-4)
rec, on = [], [False]
def wrap(name, fn, desc):
    def inner(*x, **k):
        if not on[0]:
            return fn(*x, **k)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); out = fn(*x, **k); e.record()
        rec.append((name, desc(*x, **k), s, e))
        return out
    return inner
ops.gconv_raw = wrap('gconv', ops.gconv_raw, lambda A, nbr, M, C, F, Wt, N, **k: 'M=%-6d N=%-5d C=%-4d F=%-2d %s%s%s' % (
    M, N, C, F, 'split3 ' if k.get('Wt3') is not None else '', 'scatter ' if k.get('scat') is not None else '', 'perm' if k.get('row_perm') is not None else ''))
ops.wgrad_raw = wrap('wgrad', ops.wgrad_raw, lambda A, nbr, M, C, F, dY, N, **k: 'M=%-6d N=%-5d C=%-4d F=%-2d %s' % (M, N, C, F, 'taps' if k.get('taps') is not None else ''))
ops.splat_raw = wrap('splat', ops.splat_raw, lambda feat, csr, Hh, *x, **k: 'N=%d C=%d H=%d' % (feat.shape[0], feat.shape[1], Hh))
ops.slice_raw = wrap('slice', ops.slice_raw, lambda Y, bary, off, N, *x, **k: 'H=%d C=%d N=%d' % (Y.shape[0], Y.shape[1], N))
ops.leaky_bwd = wrap('leaky_bwd', ops.leaky_bwd, lambda dY, Y, *x, **k: '%d x %d' % tuple(Y.shape))
ops.colsum = wrap('colsum', ops.colsum, lambda X: '%d x %d' % tuple(X.shape))
def step():
    with torch.no_grad():
        lat = gen.build(t1, t2)
    flow = model(t1[None], t2[None], lat)
    loss = torch.norm(flow - tsf[None], p=2, dim=1).mean()
    opt.zero_grad(set_to_none=True)
    loss.backward()
    opt.step()
for _ in range(3): step()
torch.cuda.synchronize()
s0, s1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
on[0] = True
s0.record(); step(); s1.record(); torch.cuda.synchronize()
on[0] = False
print('step with per-call events: %.2f ms' % s0.elapsed_time(s1))
tot = {}
for name, d, s, e in rec:
    us = 1e3 * s.elapsed_time(e)
    tot[name] = tot.get(name, 0) + us
    print('%-10s %8.1f us  %s' % (name, us, d))
print('sums (us):', {k: round(v) for k, v in tot.items()}, 'all', round(sum(tot.values())))
