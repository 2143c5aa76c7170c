#!/usr/bin/env python
"""GPU box: where the host thread's time goes in one pair (lattice build + prepare, forward enqueue): inside the
C ABI calls (argument conversion + hipLaunchKernel), inside torch.empty, inside other torch calls, and the
Python glue around them.  Wrapping adds ~0.2 us per call; totals are per pair, single stream.
    python tools/host_profile.py [--data surface|frustum]
"""
import argparse, collections, os, sys, time, types
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import hplflownet_amd as H
from hplflownet_amd import _lib
from hplflownet_amd.synthetic import SCALES_FILTER_MAP, fill_module_, surface_pair, synthetic_pair

ap = argparse.ArgumentParser()
ap.add_argument('--data', default='surface')
a = ap.parse_args()
T = collections.defaultdict(float)
Nc = collections.defaultdict(int)
pc = time.perf_counter


def Timed(fn, key):
    def wrapper(*args, **kw):
        t = pc()
        try:
            return fn(*args, **kw)
        finally:
            T[key] += pc() - t
            Nc[key] += 1
    return wrapper


lib = _lib.load()
class LibProxy(object):
    pass
proxy = LibProxy()
for name in _lib.EXPORTS:
    setattr(proxy, name, Timed(getattr(lib, name), 'C ABI calls'))
_lib._lib = proxy
for name in ('empty', 'cat', 'zeros', 'empty_like'):
    setattr(torch, name, Timed(getattr(torch, name), 'torch.' + ('empty' if 'empty' in name else name)))
for name in ('copy_', 'contiguous', 'view', 'reshape', '__getitem__', 'narrow', 'float', 't'):
    setattr(torch.Tensor, name, Timed(getattr(torch.Tensor, name), 'tensor methods (copy_/view/slice/...)'))

args = types.SimpleNamespace(dim=3, scales_filter_map=SCALES_FILTER_MAP, evaluate=True, use_leaky=True, bcn_use_bias=True,
                             bcn_use_norm=True, last_relu=False, DEVICE='cuda')
model = H.HPLFlowNet(args); fill_module_(model, 1.0, 'hash'); model = model.to('cuda').eval()
gen = H.GenerateDataUnsymmetric(args, device='cuda')
p1, p2, _ = (surface_pair if a.data == 'surface' else synthetic_pair)(8192, 0)
t1 = torch.from_numpy(p1.T.copy()).cuda(); t2 = torch.from_numpy(p2.T.copy()).cuda()
with torch.no_grad():
    for _ in range(3):
        lat = gen.build(t1, t2).prepare(); model(t1[None], t2[None], lat)
    torch.cuda.synchronize()
    for phase in ('lattice', 'forward'):
        T.clear(); Nc.clear()
        reps = 20
        t0 = pc()
        for _ in range(reps):
            if phase == 'lattice':
                lat = gen.build(t1, t2).prepare()
            else:
                model(t1[None], t2[None], lat)
        tot = pc() - t0
        torch.cuda.synchronize()
        print('%s (%s data): %.3f ms per pair on the host' % (phase, a.data, 1e3 * tot / reps))
        acc = 0.0
        for k in sorted(T, key=lambda k: -T[k]):
            print('   %-42s %6.1f calls  %.3f ms' % (k, Nc[k] / reps, 1e3 * T[k] / reps))
            acc += T[k]
        print('   %-42s %13s  %.3f ms' % ('Python glue (rest)', '', 1e3 * (tot - acc) / reps))
