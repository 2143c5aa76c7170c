#!/usr/bin/env python
"""GPU: dense (F=1) and gathered GEMM shapes chosen to separate tail effects from kernel efficiency."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hplflownet_amd import ops
dev = 'cuda'
def run(name, M, C, F, O, valid=1.0, reps=5):
    A = torch.randn(M, C, device=dev)
    W = torch.randn(O, C, F, device=dev) / (C * F) ** 0.5
    Wt = ops.weight_relayout(W, C, O, F, F, C * F, 1)
    nbr = None
    if F > 1:
        nbr = torch.randint(0, M, (F, M), device=dev, dtype=torch.int32)
        if valid < 1.0:
            nbr[torch.rand(F, M, device=dev) > valid] = -1
    y = ops.gconv_raw(A, nbr, M, C, F, Wt, O)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        ops.gconv_raw(A, nbr, M, C, F, Wt, O, out=y)
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / reps
    print('%-34s M=%6d K=%5d N=%5d %8.3f ms %6.1f TFLOP/s' % (name, M, F * C, O, ms, 2.0 * M * F * C * O / ms / 1e9))
run('dense 3 full waves', 24576, 8192, 1, 1024)
run('dense 1 full wave', 8192, 8192, 1, 1024)
run('dense 1 block/CU', 4096, 8192, 1, 1024)
run('dense 6 full waves', 49152, 8192, 1, 1024)
run('gather random rows 3 waves', 24576, 576, 15, 1024)
run('gather random rows 42% valid', 24576, 576, 15, 1024, valid=0.42)
run('gather C=580 (K tail, straddle)', 24576, 580, 15, 1024)
run('dense long K 1 wave', 8192, 32768, 1, 1024, reps=3)
