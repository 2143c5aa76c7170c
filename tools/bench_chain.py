#!/usr/bin/env python
"""What a chain of small dependent steps costs on this GPU: as kernel launches on one stream, and as phases of ONE persistent
launch separated by a grid barrier (libhplbcl_diag.so hpl_diag_chain; DESIGN.md section 9, "one launch for levels 3-6").

    python tools/bench_chain.py [--steps 100]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hplflownet_amd import _lib                                # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=100)
    a = ap.parse_args()
    dev = torch.device('cuda:0')
    lib = _lib.load_diag()
    bar = torch.zeros(320, dtype=torch.int32, device=dev)
    print('%-8s %-8s %-40s %10s' % ('grid', 'bytes/WG', 'form', 'us / step'))
    for grid in (32, 64, 128, 256):
        for words in (256, 4096):
            x0 = torch.rand(grid * words, device=dev)
            for mode, name in ((0, 'kernel launches (one stream)'), (1, 'persistent, counter barrier'), (2, 'persistent, XCD-hierarchical barrier')):
                best = None
                for rep in range(5):
                    x, y = x0.clone(), torch.empty_like(x0)
                    torch.cuda.synchronize()
                    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    s.record()
                    rc = lib.hpl_diag_chain(x.data_ptr(), y.data_ptr(), grid, words, a.steps, bar.data_ptr(), mode, _lib.stream())
                    e.record()
                    torch.cuda.synchronize()
                    assert rc == 0, rc
                    out = y if a.steps % 2 else x
                    # every block has travelled `steps` blocks down the ring and gained `steps`
                    want = x0.view(grid, words).roll(-(a.steps % grid), 0).reshape(-1) + a.steps
                    assert torch.allclose(out, want), (grid, words, mode, float((out - want).abs().max()))
                    t = s.elapsed_time(e) * 1e3 / a.steps
                    best = t if best is None else min(best, t)
                print('%-8d %-8d %-40s %10.2f' % (grid, words * 4, name, best))


if __name__ == '__main__':
    main()
