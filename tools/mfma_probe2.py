#!/usr/bin/env python
"""MFMA pipe rate vs accumulator count and waves per SIMD (hpl_mfma_probe): 4 independent
accumulators per wave vs one dependent chain, 1..8 waves per SIMD."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hplflownet_amd import _lib

L = _lib.load()
dev = torch.device('cuda:0')
out = torch.empty(8192 * 256, device=dev)
for mode, sign in (('4 accumulators', 1), ('1 accumulator (dependent chain)', -1)):
    for wps in (1, 2, 4, 8):                    # waves per SIMD: blocks of 4 waves, 256 CUs
        blocks = 256 * wps
        iters = 4000 // wps
        best = 0.0
        for _ in range(3):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            L.hpl_mfma_probe(out.data_ptr(), blocks, sign * iters, _lib.stream())
            e.record()
            torch.cuda.synchronize()
            best = max(best, blocks * 4.0 * iters * 64 * 4096 / (s.elapsed_time(e) * 1e-3) / 1e12)
        print('%-34s %d waves/SIMD: %6.1f TFLOP/s' % (mode, wps, best))
