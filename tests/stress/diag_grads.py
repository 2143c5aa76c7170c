"""GPU diagnostic: whole-model gradient norms, HIP ops vs a torch float64 restatement of the same
ops under the SAME model code, vs the golden fixture."""
import os, sys, types
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import hplflownet_amd as H
from hplflownet_amd import ops
from common import GOLD, oracle_lattice
from hplflownet_amd.synthetic import *
from test_gpu_layers import model_args, gd_batched_device, T
from test_gpu_autograd import ref_gconv

def t_gconv(A, weight, bias, nbr, M, F, act=0, c0=0, C=None, res=None, res_mod=0, bwd_mode='scatter', out=None, slope=0.1):
    O = weight.shape[0]; Ctot = weight.numel() // (O * F); C = Ctot if C is None else C
    y = ref_gconv(A, weight.view(O, Ctot, F), bias, nbr, M, c0, C, F, act != 0, res, res_mod or (res.shape[0] if res is not None else 0), slope).float()
    if out is not None:
        out.copy_(y); return out
    return y
def t_splat(x, cloud, use_norm):
    offl = cloud.off.long().view(4, -1); b = cloud.bary.double()
    S = torch.zeros(cloud.H, x.shape[1], dtype=torch.float64, device=x.device); w = torch.zeros(cloud.H, dtype=torch.float64, device=x.device)
    for r in range(4):
        S = S.index_add(0, offl[r], b[r][:, None] * x.double()); w = w.index_add(0, offl[r], b[r])
    if use_norm: S = S / (w + 1e-5)[:, None]
    return S.float()
def t_slice(y, cloud, bias):
    offl = cloud.off.long().view(4, -1); b = cloud.bary.double()
    o = sum(b[r][:, None] * y.double()[offl[r]] for r in range(4))
    if bias is not None: o = o + bias.double()[None]
    return o.float()
class FakeFn:
    def __init__(self, f): self.apply = f

z = np.load(os.path.join(GOLD, 'models.npz'))
tag, cls, n, nsc = 'shallow_n256', 'HPLFlowNetShallow', 256, 5
pc1, pc2, sf, gd = oracle_lattice(n)
res = {}
for variant in ('hip', 'torch'):
    if variant == 'torch':
        ops.gconv = t_gconv
        ops.SplatFn = FakeFn(t_splat); ops.SliceFn = FakeFn(t_slice)
    m = getattr(H, cls)(model_args(nsc)); fill_module_(m, 1.0, 'hash'); m = m.to('cuda')
    y = m(T(pc1.T)[None], T(pc2.T)[None], gd_batched_device(gd[:nsc]))
    loss = torch.norm(y - T(sf.T)[None], p=2, dim=1).mean(); loss.backward()
    print(variant, 'loss', loss.item(), 'fixture', float(z[tag + '_loss']), 'max flow err', np.abs(y.detach().cpu().numpy()[0] - z[tag + '_flow']).max())
    res[variant] = {k: float(p.grad.norm()) for k, p in m.named_parameters()}
names = bytes(z[tag + '_gradnames']).decode().split('\n')
want = dict(zip(names, z[tag + '_gradnorm']))
for k in res['hip']:
    a, b, w = res['hip'][k], res['torch'][k], want[k]
    print('%-45s hip %.6g torch %.6g want %.6g | hip-vs-want %.1e torch-vs-want %.1e' % (k, a, b, w, abs(a - w) / w, abs(b - w) / w))
