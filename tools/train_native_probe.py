#!/usr/bin/env python
"""GPU: the native training step (train_plan.TrainPlan) beside the autograd step on one N-point pair: ms per step (steady state,
same lattice), host enqueue time of a step, launches per step (count of executed ops)."""
import os, sys, time, types, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import hplflownet_amd as H
from hplflownet_amd import ops
from hplflownet_amd.synthetic import SCALES_FILTER_MAP, fill_module_, synthetic_pair
from hplflownet_amd.train_plan import TrainPlan
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
dev = 'cuda'
a = types.SimpleNamespace(dim=3, scales_filter_map=SCALES_FILTER_MAP, evaluate=False, use_leaky=True, bcn_use_bias=True,
                          bcn_use_norm=True, last_relu=False, DEVICE='cuda')
model = H.HPLFlowNet(a); fill_module_(model, 1.0, 'hash'); model = model.to(dev).train()
gen = H.GenerateDataUnsymmetric(a, device=dev, wide_up=model.lattice_hint())
pc1, pc2, sf = synthetic_pair(N, 0)
t1, t2, tsf = [torch.from_numpy(x.T.copy()).to(dev) for x in (pc1, pc2, sf)]
lat = gen.build_native(t1, t2).device_lattice().prepare(True)
opt = torch.optim.Adam(model.parameters(), lr=1e-4, fused=not os.environ.get('PROBE_UNFUSED_ADAM'))      # (engine.Trainer and bench.py use the fused optimizer)

def timed(fn, n=10, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter(); host = 0.0
    for _ in range(n):
        h0 = time.perf_counter(); fn(); host += time.perf_counter() - h0
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3 / n, host * 1e3 / n

ONLY = os.environ.get('PROBE_ONLY')          # 'native': skip the autograd leg (kernel traces of the native step alone)
ops.enable_weight_bank(True)
def auto():
    flow = model(t1[None], t2[None], lat)
    loss = torch.norm(flow - tsf[None], p=2, dim=1).mean()
    opt.zero_grad(set_to_none=True)
    loss.backward()
    opt.step()
if ONLY != 'native':
    ms, host = timed(auto)
    print('autograd step: %.2f ms (host enqueue %.2f ms)' % (ms, host))
ops.enable_weight_bank(False)
for side in ((True,) if ONLY == 'native' else (False, True)):
    plan = TrainPlan(model, side_stream=side)
    def nat():
        plan.step(t1, t2, tsf, lat); plan.finish()
        if not plan.adam_step(opt):              # one launch over the flat arrays, as engine.Trainer does
            opt.step()
    ms, host = timed(nat)
    if side:
        hi = torch.cuda.Stream(priority=-1)
        hi.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(hi):
            msh, hosth = timed(nat)
        torch.cuda.current_stream().wait_stream(hi)
        print('native step on a high-priority stream (side stream at normal priority): %.2f ms (host %.2f)' % (msh, hosth))
    nfwd = plan.n_fwd
    print('native step (side stream %s): %.2f ms (host enqueue %.2f ms); program: %d forward + %d backward ops, %d un-layout buckets'
          % (side, ms, host, nfwd, len(plan.prog.ops) - nfwd, len(plan.cuts)))
    torch.cuda.synchronize()
    h0 = time.perf_counter(); plan.step(t1, t2, tsf, lat); h1 = time.perf_counter(); plan.finish(); plan.adam_step(opt) or opt.step(); h2 = time.perf_counter()
    torch.cuda.synchronize()
    print('   host time of one step issued to an idle GPU: program %.2f ms, Adam %.2f ms' % ((h1 - h0) * 1e3, (h2 - h1) * 1e3))
    def fwd_only():
        plan.refresh_weights()
    ms2, h2 = timed(fwd_only)
    print('   of which weight refresh: %.2f ms' % ms2)
    del plan
