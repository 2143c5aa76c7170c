#!/usr/bin/env python
"""GPU box: whole-model forwards on awkward inputs (1 .. 700 points, ragged pairs, duplicated points) against
the numpy oracle on the same device-built lattice; inference path, training path, loss and EPE3D.
    python tests/stress/stress_models.py [--cases 24]
"""
import argparse, os, sys, types
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import hplflownet_amd as H
from hplflownet_amd.synthetic import SCALES_FILTER_MAP, fill_module_
from oracle import bcl_oracle as BO


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--cases', type=int, default=24)
    a = ap.parse_args()
    rng = np.random.RandomState(0)
    sizes = [1, 2, 3, 5, 17, 33, 64, 100, 257, 700]
    bad = 0
    for case in range(a.cases):
        n1 = sizes[case % len(sizes)]
        n2 = n1 if case % 3 else max(1, n1 - rng.randint(0, max(1, n1 // 2) + 1))
        cls, nsc = (('HPLFlowNet', 7), ('HPLFlowNetShallow', 5))[case % 2]
        p1 = np.stack([rng.uniform(-6, 6, n1), rng.uniform(-4, 4, n1), rng.uniform(1.5, 30, n1)], 1).astype(np.float32)
        p2 = (p1[rng.randint(0, n1, n2)] + rng.normal(0, 0.3, (n2, 3))).astype(np.float32)
        if case % 5 == 4:
            p1[:] = p1[0]
        args = types.SimpleNamespace(dim=3, scales_filter_map=SCALES_FILTER_MAP[:nsc], evaluate=True, use_leaky=True,
                                     bcn_use_bias=True, bcn_use_norm=True, last_relu=False, DEVICE='cuda')
        m = getattr(H, cls)(args)
        fill_module_(m, 1.0, 'hash')
        m = m.to('cuda').eval()
        gen = H.GenerateDataUnsymmetric(args, device='cuda')
        t1, t2, _, lat = gen([p1, p2, np.zeros_like(p1)])
        with torch.no_grad():
            y = m(t1[None], t2[None], lat)
        yt = m(t1[None], t2[None], gen.build(t1, t2).prepare(for_training=True))        # autograd path
        gd = [{k: (v.cpu().numpy() if torch.is_tensor(v) else v) for k, v in d.items()} for d in H.to_reference_format(lat)]
        sd = {k: v.detach().cpu().numpy() for k, v in m.state_dict().items()}
        yo = BO.hplflownet_forward(sd, p1.T.copy(), p2.T.copy(), gd, shallow=(cls == 'HPLFlowNetShallow'))
        sc = max(1.0, float(np.abs(yo).max()))
        e1 = float(np.abs(y.cpu().numpy()[0] - yo).max()) / sc
        e2 = float(np.abs(yt.detach().cpu().numpy()[0] - yo).max()) / sc
        ok = e1 < 2e-4 and e2 < 2e-4 and tuple(y.shape) == (1, 3, n1)
        bad += not ok
        print('%s case %2d %-17s n=(%d,%d) H0=%s  inference %.1e  training-path %.1e' % ('ok ' if ok else 'BAD', case, cls, n1, n2,
                                                                                       lat.levels[0].H, e1, e2))
    print('DONE %d cases, %d bad' % (a.cases, bad))
    return 1 if bad else 0


if __name__ == '__main__':
    sys.exit(main())
