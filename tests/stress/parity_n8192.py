#!/usr/bin/env python
"""GPU box: full HPLFlowNet at the benchmark size (N=8192) against the numpy/C oracle on several seeds --
device lattice bit-exact, flow and EPE3D within the north-star tolerance (1e-4).  ~10 s of CPU per seed.
    python tests/stress/parity_n8192.py [--seeds 1 2 3]
"""
import argparse, os, sys, types
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import hplflownet_amd as H
from hplflownet_amd.synthetic import SCALES_FILTER_MAP, fill_module_, synthetic_pair
from oracle import bcl_oracle as BO, lattice_oracle as LO

ap = argparse.ArgumentParser()
ap.add_argument('--seeds', type=int, nargs='*', default=[1, 2, 3])
ap.add_argument('--points', type=int, default=8192)
a = ap.parse_args()
args = types.SimpleNamespace(dim=3, scales_filter_map=SCALES_FILTER_MAP, evaluate=True, use_leaky=True, bcn_use_bias=True,
                             bcn_use_norm=True, last_relu=False, DEVICE='cuda')
m = H.HPLFlowNet(args)
fill_module_(m, 1.0, 'hash')
sd = {k: v.numpy().copy() for k, v in m.state_dict().items()}
m = m.to('cuda').eval()
gen = H.GenerateDataUnsymmetric(args, device='cuda')
bad = 0
for seed in a.seeds:
    pc1, pc2, sf = synthetic_pair(a.points, seed)
    t1, t2, _, lat = gen([pc1, pc2, sf])
    with torch.no_grad():
        y = m(t1[None], t2[None], lat)[0].cpu().numpy()
    gd = LO.generate_data(pc1, pc2, SCALES_FILTER_MAP)
    same = all(np.array_equal(np.asarray(x[k].cpu().numpy() if torch.is_tensor(x[k]) else x[k]), np.asarray(g[k]))
               for x, g in zip(H.to_reference_format(lat), gd) for k in g)
    yo = BO.hplflownet_forward(sd, pc1.T, pc2.T, gd)
    e_gpu, e_cpu = BO.epe3d(y, sf.T), BO.epe3d(yo, sf.T)
    ok = same and abs(e_gpu - e_cpu) < 1e-4 and np.abs(y - yo).max() < 2e-4 * max(1.0, np.abs(yo).max())
    bad += not ok
    print('%s seed %d: lattice bit-exact %s, EPE3D gpu %.6f oracle %.6f (delta %.1e), max|flow diff| %.1e of max %.1f'
          % ('ok ' if ok else 'BAD', seed, same, e_gpu, e_cpu, abs(e_gpu - e_cpu), np.abs(y - yo).max(), np.abs(yo).max()))
print('DONE, %d bad' % bad)
sys.exit(1 if bad else 0)
