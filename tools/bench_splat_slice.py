"""Splat / slice kernels alone against the HBM roofline: every instance of the
full model at N=8192 (real tables of the synthetic pair), as launched by the model (one pair) and
as a batched variant (B pairs in one launch: points concatenated, vertex ids offset by b*H).

    python tools/bench_splat_slice.py [--batch 16] [--reps 30] [--json out.json]

Algorithmic bytes: splat 4*C*N + 32*N + 4*(C+1)*(H+1); slice 4*C*H + 32*N + 4*C*N.
"""
import argparse
import json
import os
import sys
from types import SimpleNamespace

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hplflownet_amd import _lib, ops, synthetic                                # noqa: E402
from hplflownet_amd.lattice import GenerateDataUnsymmetric               # noqa: E402

HBM_PEAK = 8.0e12


def timed(fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True)
    e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e-3 / reps


def batched(cloud, B):
    """B copies of a cloud's tables with disjoint vertex ranges."""
    N, H = cloud.N, cloud.H
    shift = (torch.arange(B, device=cloud.off.device, dtype=torch.int32) * H).repeat_interleave(N)[None, :]
    off = cloud.off.repeat(1, B) + shift
    bary = cloud.bary.repeat(1, B)
    return ops.CloudTables(bary, off, H * B)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--points', type=int, default=8192)
    ap.add_argument('--batch', type=int, default=16)
    ap.add_argument('--reps', type=int, default=30)
    ap.add_argument('--json', default=None)
    ap.add_argument('--atomic', action='store_true',
                    help='also time the splat as scatter-adds (libhplbcl_diag.so hpl_diag_splat_atomic: global atomics / LDS-staged), '
                         'the A/B of the product CSR reduction')
    a = ap.parse_args()
    dev = torch.device('cuda:0')
    pc1, pc2, _ = synthetic.synthetic_pair(a.points, 0)
    args = SimpleNamespace(dim=3, scales_filter_map=synthetic.SCALES_FILTER_MAP)
    gen = GenerateDataUnsymmetric(args, device=dev)
    lat = gen.build(torch.from_numpy(pc1.T.copy()).to(dev), torch.from_numpy(pc2.T.copy()).to(dev))
    up_c = {0: 1024, 1: 512, 2: 256, 3: 256, 4: 128, 5: 128, 6: 128}     # flownet.HPLFlowNet.UP
    rows = []
    for L, lv in enumerate(lat.levels):
        cloud = lv.clouds[0]
        for B in (1, a.batch):
            cl = cloud if B == 1 else batched(cloud, B)
            N, H = cl.N, cl.H
            # Down BCL splat: 64 features + 4 position features
            C = 68
            feat = torch.randn(N, C, device=dev)
            csr = cl.csr()
            out = torch.empty(H, C, device=dev)
            t = timed(lambda: ops.splat_raw(feat, csr, H, True, out=out), a.reps)
            by = 4.0 * C * N + 32.0 * N + 4.0 * (C + 1) * (H + 1)
            rows.append(dict(kernel='splat', level=L, batch=B, C=C, N=N, H=H, us=t * 1e6, MB=by / 1e6,
                             GBps=by / t / 1e9, frac=by / t / HBM_PEAK))
            if a.atomic:
                diag = _lib.load_diag()
                norm = csr[3]
                ref = out.clone()
                for mode, nm in ((0, 'splat_atomic'), (1, 'splat_atomic_lds')):
                    o3 = torch.empty(H, C, device=dev)

                    def run():
                        rc = diag.hpl_diag_splat_atomic(feat.data_ptr(), C, C, cl.bary.data_ptr(), cl.off.data_ptr(), N, norm.data_ptr(), H,
                                                        o3.data_ptr(), C, mode, _lib.stream())
                        assert rc == 0, rc
                    t = timed(run, a.reps)
                    err = float((o3 - ref).abs().max() / ref.abs().max().clamp_min(1e-30))
                    assert err < 1e-4, (nm, L, B, err)
                    rows.append(dict(kernel=nm, level=L, batch=B, C=C, N=N, H=H, us=t * 1e6, MB=by / 1e6,
                                     GBps=by / t / 1e9, frac=by / t / HBM_PEAK, max_rel_diff_vs_csr=err))
            # Up BCL slice
            C = up_c[L]
            Y = torch.randn(H, C, device=dev)
            bias = torch.randn(C, device=dev)
            o2 = torch.empty(N, C, device=dev)
            t = timed(lambda: ops.slice_raw(Y, cl.bary, cl.off, N, bias=bias, out=o2), a.reps)
            by = 4.0 * C * H + 32.0 * N + 4.0 * C * N
            rows.append(dict(kernel='slice', level=L, batch=B, C=C, N=N, H=H, us=t * 1e6, MB=by / 1e6,
                             GBps=by / t / 1e9, frac=by / t / HBM_PEAK))
            del feat, out, Y, o2
    print('%-16s %3s %5s %5s %9s %9s %9s %8s %9s %6s' % ('kernel', 'lvl', 'batch', 'C', 'N', 'H', 'us', 'MB', 'GB/s',
                                                       'frac'))
    for r in rows:
        print('%-16s %3d %5d %5d %9d %9d %9.1f %8.1f %9.0f %6.3f' % (r['kernel'], r['level'], r['batch'], r['C'],
                                                                    r['N'], r['H'], r['us'], r['MB'], r['GBps'],
                                                                    r['frac']))
    if a.json:
        with open(a.json, 'w') as f:
            json.dump(rows, f, indent=1)


if __name__ == '__main__':
    main()
