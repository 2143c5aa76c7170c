#!/usr/bin/env python
"""Calibration point for the roofline discussion: what the vendor library's fp32 GEMM (torch.mm -> hipBLASLt / rocBLAS,
fp32 MFMA) reaches on this chip for the plain-GEMM shapes of the model and for a large square problem."""
import torch
dev = 'cuda'
torch.backends.cuda.matmul.allow_tf32 = False
shapes = [('conv2 8192x1024x1024', 8192, 1024, 1024), ('bcn2_ 1x1 25841x512x512', 25841, 512, 512),
          ('long K 25841x8704x1024 (bcn1_ blur, dense)', 25841, 8704, 1024), ('square 8192^3', 8192, 8192, 8192),
          ('square 16384x8192x8192', 16384, 8192, 8192)]
for name, M, K, N in shapes:
    a = torch.randn(M, K, device=dev)
    b = torch.randn(K, N, device=dev)
    c = torch.mm(a, b)
    torch.cuda.synchronize()
    reps = 10
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        torch.mm(a, b, out=c)
    e.record(); torch.cuda.synchronize()
    us = 1e3 * s.elapsed_time(e) / reps
    print('%-46s %9.1f us  %6.1f TFLOP/s  (%.2f of the 157.3 TF fp32-MFMA peak)' % (name, us, 2.0 * M * K * N / us / 1e6,
                                                                                   2.0 * M * K * N / us / 1e6 / 157.3))
