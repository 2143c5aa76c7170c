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

