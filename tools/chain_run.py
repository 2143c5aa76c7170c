#!/usr/bin/env python
"""K single-stream steps (native lattice build + native forward) and nothing else, for a kernel trace:
    rocprofv3 --kernel-trace -d out -o k -- python tools/chain_run.py [frustum|surface] [points] [arch]
tools/chain_trace.py then splits the LAST step into lattice / forward launches and the idle time in front of each."""
import os
import sys
import time
import types

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hplflownet_amd as H                                                       # noqa: E402
from hplflownet_amd.synthetic import SCALES_FILTER_MAP, fill_module_, surface_pair, synthetic_pair   # noqa: E402


def main():
    data = sys.argv[1] if len(sys.argv) > 1 else 'frustum'
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
    arch = sys.argv[3] if len(sys.argv) > 3 else 'HPLFlowNet'
    dev = torch.device('cuda:0')
    sfm = SCALES_FILTER_MAP if arch == 'HPLFlowNet' else SCALES_FILTER_MAP[:5]
    margs = types.SimpleNamespace(dim=3, scales_filter_map=sfm, evaluate=True, use_leaky=True, bcn_use_bias=True,
                                  bcn_use_norm=True, last_relu=False, DEVICE='cuda')
    model = getattr(H, arch)(margs)
    fill_module_(model, 1.0, 'hash')
    model = model.to(dev).eval()
    gen = H.GenerateDataUnsymmetric(margs, device=dev, wide_up=model.lattice_hint())
    p1, p2, _ = (surface_pair if data == 'surface' else synthetic_pair)(n, 0)
    p1, p2 = torch.from_numpy(p1.T.copy()).to(dev), torch.from_numpy(p2.T.copy()).to(dev)
    ts = []
    with torch.no_grad():
        for i in range(6):
            torch.cuda.synchronize()
            t = time.perf_counter()
            lat = gen.build_native(p1, p2)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            y = model(p1[None], p2[None], lat)
            torch.cuda.synchronize()
            ts.append((t1 - t, time.perf_counter() - t1))
    print('wall per step (build, forward) ms:', [(round(a * 1e3, 3), round(b * 1e3, 3)) for a, b in ts[2:]])


if __name__ == '__main__':
    main()
