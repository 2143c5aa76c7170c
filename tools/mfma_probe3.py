#!/usr/bin/env python
"""What the fp32 matrix pipe sustains on this chip by operand activity: constant operands (hpl_mfma_probe), changing
register operands, operands streamed from LDS.  Rate in TFLOP/s and the shader clock measured inside the kernel."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hplflownet_amd import _lib
L = _lib.load()
dev = 'cuda'
out = torch.empty(2048 * 256, device=dev)
clk = torch.zeros(2, dtype=torch.int64, device=dev)
blocks, iters = 1024, 1500
flop = blocks * 4.0 * iters * 64 * 4096
def run(fn, name):
    best = None
    for _ in range(4):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record(); torch.cuda.synchronize()
        ms = s.elapsed_time(e)
        c = clk.tolist()
        ghz = c[0] / max(1, c[1]) * 0.1
        if best is None or ms < best[0]:
            best = (ms, ghz)
    print('%-42s %7.3f ms  %6.1f TFLOP/s  clock %.2f GHz' % (name, best[0], flop / best[0] / 1e9, best[1]))
run(lambda: L.hpl_mfma_probe(out.data_ptr(), blocks, iters, _lib.stream()), 'constant operands (no clock stamp)')
run(lambda: L.hpl_mfma_probe_data(out.data_ptr(), blocks, iters, 1, clk.data_ptr(), _lib.stream()), 'changing register operands')
run(lambda: L.hpl_mfma_probe_data(out.data_ptr(), blocks, iters, 2, clk.data_ptr(), _lib.stream()), 'operands from LDS (2 ds_read_b32 / MFMA)')
blocks = 2048
flop = blocks * 4.0 * iters * 64 * 4096
run(lambda: L.hpl_mfma_probe_data(out.data_ptr(), blocks, iters, 1, clk.data_ptr(), _lib.stream()), '... 2 workgroups per CU, registers')
run(lambda: L.hpl_mfma_probe_data(out.data_ptr(), blocks, iters, 2, clk.data_ptr(), _lib.stream()), '... 2 workgroups per CU, LDS')
