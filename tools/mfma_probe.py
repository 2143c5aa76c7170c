#!/usr/bin/env python
"""GPU: sustained fp32-MFMA rate (no memory traffic) at 1 and 2 workgroups per CU."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hplflownet_amd import _lib
L = _lib.load_diag()
for blocks in (256, 512, 1024):
    out = torch.empty(blocks * 256, device='cuda')
    iters = 4000
    L.hpl_mfma_probe(out.data_ptr(), blocks, 100, _lib.stream())
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); L.hpl_mfma_probe(out.data_ptr(), blocks, iters, _lib.stream()); e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e)
    fl = blocks * 4.0 * iters * 64 * 4096
    print('blocks=%4d  %.3f ms  %.1f TFLOP/s' % (blocks, ms, fl / ms / 1e9))
