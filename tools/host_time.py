#!/usr/bin/env python
"""Host wall time of one inference forward's ENQUEUE (GPU idle, nothing to wait for): Python path vs native plan."""
import os, sys, time, types
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import hplflownet_amd as H
from hplflownet_amd.synthetic import SCALES_FILTER_MAP, fill_module_, surface_pair, synthetic_pair

dev = 'cuda'
for arch, nsc, n, kind in (('HPLFlowNet', 7, 8192, 'frustum'), ('HPLFlowNet', 7, 8192, 'surface'),
                           ('HPLFlowNetShallow', 5, 4096, 'frustum')):
    args = types.SimpleNamespace(dim=3, scales_filter_map=SCALES_FILTER_MAP[:nsc], evaluate=True, use_leaky=True,
                                 bcn_use_bias=True, bcn_use_norm=True, last_relu=False, DEVICE='cuda')
    m = getattr(H, arch)(args)
    fill_module_(m, 1.0, 'hash')
    m = m.to(dev).eval()
    gen = H.GenerateDataUnsymmetric(args, device=dev, wide_up=m.lattice_hint())
    pc1, pc2, sf = (surface_pair if kind == 'surface' else synthetic_pair)(n, 0)
    t1, t2, _, lat = gen([pc1, pc2, sf])
    lat.prepare()
    res = {}
    with torch.no_grad():
        for native in (False, True):
            m.native_forward = native
            for _ in range(5):
                m(t1[None], t2[None], lat)
            torch.cuda.synchronize()
            host, gpu = [], []
            for _ in range(30):
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                t0 = time.perf_counter()
                s.record()
                m(t1[None], t2[None], lat)
                e.record()
                host.append((time.perf_counter() - t0) * 1e3)
                torch.cuda.synchronize()
                gpu.append(s.elapsed_time(e))
            res[native] = (sorted(host)[len(host) // 2], sorted(gpu)[len(gpu) // 2])
    print('%-18s N=%5d %-8s python: host %.3f ms gpu %.3f ms | native: host %.3f ms gpu %.3f ms' %
          (arch, n, kind, res[False][0], res[False][1], res[True][0], res[True][1]))
