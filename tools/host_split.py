#!/usr/bin/env python
"""GPU box: where the host thread spends a pipelined inference step -- lattice launches, lattice
read-back waits (the vertex counts of each level), forward enqueue, waiting for old steps."""
import collections, os, sys, time, types, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import hplflownet_amd as H
from hplflownet_amd.synthetic import SCALES_FILTER_MAP, fill_module_, synthetic_pair
dev = torch.device('cuda:0')
a = types.SimpleNamespace(dim=3, scales_filter_map=SCALES_FILTER_MAP, evaluate=True, use_leaky=True, bcn_use_bias=True,
                          bcn_use_norm=True, last_relu=False, DEVICE='cuda')
model = H.HPLFlowNet(a); fill_module_(model, 1.0, 'hash'); model = model.to(dev).eval()
gen = H.GenerateDataUnsymmetric(a, device=dev)
pairs = []
for s in range(8):
    pc1, pc2, sf = synthetic_pair(8192, s)
    pairs.append((torch.from_numpy(pc1.T.copy()).to(dev), torch.from_numpy(pc2.T.copy()).to(dev)))
side = torch.cuda.Stream(priority=-1)
fwd = [torch.cuda.Stream() for _ in range(3)]
T = collections.defaultdict(float)
orig = torch.Tensor.tolist
def timed_tolist(self):
    t = time.perf_counter(); r = orig(self); T['readback_wait'] += time.perf_counter() - t; return r
torch.Tensor.tolist = timed_tolist

def build(i):
    t = time.perf_counter()
    with torch.cuda.stream(side), torch.no_grad():
        lat = gen.build(*pairs[i % 8]); t1 = time.perf_counter(); lat.prepare(); ev = torch.cuda.Event(); ev.record(side)
    T['build_total'] += t1 - t; T['prepare'] += time.perf_counter() - t1
    return lat, ev

def run(n):
    keep = collections.deque(); nxt = build(0)
    for i in range(n):
        lat, ev = nxt; m = fwd[i % 3]; m.wait_event(ev)
        t = time.perf_counter()
        with torch.cuda.stream(m), torch.no_grad():
            out = model(pairs[i % 8][0][None], pairs[i % 8][1][None], lat)
        T['forward_enqueue'] += time.perf_counter() - t
        fin = torch.cuda.Event(); fin.record(m); keep.append((lat, out, fin))
        nxt = build(i + 1)
        t = time.perf_counter()
        while len(keep) > 4:
            keep[0][2].synchronize(); keep.popleft()
        T['wait_old'] += time.perf_counter() - t
run(10); torch.cuda.synchronize(); T.clear()
t0 = time.perf_counter(); n = 60; run(n); torch.cuda.synchronize(); tot = time.perf_counter() - t0
print('step %.3f ms' % (1e3 * tot / n))
for k, v in T.items():
    print('%-16s %.3f ms/step' % (k, 1e3 * v / n))
print('lattice launches (build_total - readback_wait): %.3f ms/step' % (1e3 * (T['build_total'] - T['readback_wait']) / n))

# the same loop fed by LatticePipeline (several pairs under construction, asynchronous read-backs)
from hplflownet_amd.lattice import LatticePipeline
torch.Tensor.tolist = orig
DEPTH = int(os.environ.get('DEPTH', '4'))


def run2(n):
    pipe = LatticePipeline(gen, lambda i: pairs[i % 8], 0, n, depth=DEPTH, stream=side)
    keep = collections.deque()
    for _ in range(n):
        t = time.perf_counter()
        (i, _), lat, ev = pipe.get()
        T['pipe.get'] += time.perf_counter() - t
        m = fwd[i % 3]; m.wait_event(ev)
        t = time.perf_counter()
        with torch.cuda.stream(m), torch.no_grad():
            out = model(pairs[i % 8][0][None], pairs[i % 8][1][None], lat)
        T['forward_enqueue'] += time.perf_counter() - t
        fin = torch.cuda.Event(); fin.record(m); keep.append((lat, out, fin))
        t = time.perf_counter()
        while len(keep) > 4:
            keep[0][2].synchronize(); keep.popleft()
        T['wait_old'] += time.perf_counter() - t


run2(10); torch.cuda.synchronize(); T.clear()
t0 = time.perf_counter(); run2(n); torch.cuda.synchronize(); tot = time.perf_counter() - t0
print('pipeline depth %d: step %.3f ms' % (DEPTH, 1e3 * tot / n))
for k, v in T.items():
    print('%-16s %.3f ms/step' % (k, 1e3 * v / n))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable(); run2(40); pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats('tottime').print_stats(28)
