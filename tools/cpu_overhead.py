#!/usr/bin/env python
"""GPU box: host-side enqueue time of one forward / one lattice build vs their GPU time."""
import os, sys, time, types, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import hplflownet_amd as H
from hplflownet_amd.synthetic import SCALES_FILTER_MAP, fill_module_, synthetic_pair
dev = 'cuda'
a = types.SimpleNamespace(dim=3, scales_filter_map=SCALES_FILTER_MAP, evaluate=True, use_leaky=True, bcn_use_bias=True,
                          bcn_use_norm=True, last_relu=False, DEVICE='cuda')
model = H.HPLFlowNet(a); fill_module_(model, 1.0, 'hash'); model = model.to(dev).eval()
gen = H.GenerateDataUnsymmetric(a, device=dev)
pc1, pc2, sf = synthetic_pair(8192, 0)
t1, t2, _, lat = gen([pc1, pc2, sf]); lat.prepare()
with torch.no_grad():
    for _ in range(3): model(t1[None], t2[None], lat)
    torch.cuda.synchronize()
    for rep in range(3):
        t0 = time.perf_counter(); y = model(t1[None], t2[None], lat); t_enq = time.perf_counter() - t0
        torch.cuda.synchronize(); t_all = time.perf_counter() - t0
        print('forward: enqueue %.2f ms, until done %.2f ms' % (1e3 * t_enq, 1e3 * t_all))
    for rep in range(3):
        t0 = time.perf_counter(); l2 = gen.build(t1, t2).prepare(); t_enq = time.perf_counter() - t0
        torch.cuda.synchronize(); t_all = time.perf_counter() - t0
        print('lattice: host %.2f ms, until done %.2f ms' % (1e3 * t_enq, 1e3 * t_all))
import cProfile, pstats
with torch.no_grad():
    pr = cProfile.Profile(); pr.enable()
    for _ in range(5): model(t1[None], t2[None], lat)
    pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats('cumulative').print_stats(18)
pstats.Stats(pr).sort_stats('tottime').print_stats(22)
with torch.no_grad():
    pr = cProfile.Profile(); pr.enable()
    for _ in range(5): gen.build(t1, t2).prepare()
    pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats('tottime').print_stats(22)
