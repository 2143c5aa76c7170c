#!/usr/bin/env python
"""Where a tile of the wide (split-operand) launches spends its cycles.
    python tools/tile_phase_probe.py build      (CPU, hipcc) writes a PROBE COPY of csrc/gconv3.hip -- every 64th workgroup adds the shader cycles
        of its prologue / contraction loop / epilogue to hpl_gconv_desc.clock_probe[2..4], the number of workgroups to [5], their slice counts to
        [6] -- and links hplflownet_amd/libhplbcl_probe.so from it and the product's objects (both untracked; delete them afterwards)
    HPL_LIB=$PWD/hplflownet_amd/libhplbcl_probe.so python tools/tile_phase_probe.py      (GPU) the table of profiles/rNN_tile_phase_probe.txt"""
import ctypes, os, subprocess, sys, types
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def build_probe():
    csrc = os.path.join(ROOT, 'hplflownet_amd', 'csrc')
    src = open(os.path.join(csrc, 'gconv3.hip')).read()

    def once(old, new):
        assert src.count(old) >= 1, old
        return src.replace(old, new, 1)
    a = "    using S2 = std::integral_constant<int, 2>;\n    if (nsl > 0) {"
    src = once(a, "    long long probe_p1 = 0; if (probe) probe_p1 = (long long)__builtin_readcyclecounter();\n" + a)
    b = "    if constexpr (PL == 2) {        // undo the operand scales"
    src = once(b, "    long long probe_p2 = 0; if (probe) probe_p2 = (long long)__builtin_readcyclecounter();\n" + b)
    c = "    if (probe) {\n        atomicAdd(reinterpret_cast<unsigned long long *>(p.clock_probe),"
    src = once(c, "    if (probe) {\n        const long long probe_e = (long long)__builtin_readcyclecounter();\n"
               "        atomicAdd(reinterpret_cast<unsigned long long *>(p.clock_probe) + 2, (unsigned long long)(probe_p1 - probe_c));\n"
               "        atomicAdd(reinterpret_cast<unsigned long long *>(p.clock_probe) + 3, (unsigned long long)(probe_p2 - probe_p1));\n"
               "        atomicAdd(reinterpret_cast<unsigned long long *>(p.clock_probe) + 4, (unsigned long long)(probe_e - probe_p2));\n"
               "        atomicAdd(reinterpret_cast<unsigned long long *>(p.clock_probe) + 5, 1ull);\n"
               "        atomicAdd(reinterpret_cast<unsigned long long *>(p.clock_probe) + 6, (unsigned long long)nsl);\n"
               "        atomicAdd(reinterpret_cast<unsigned long long *>(p.clock_probe),")
    open(os.path.join(csrc, '_probe_gconv3.hip'), 'w').write(src)
    # the fp32 kernel (csrc/gconv.hip, k_gconv) the same way
    src = open(os.path.join(csrc, 'gconv.hip')).read()
    a = "    __syncthreads();\n    int cur = 0;\n    // One contraction step t (compile-time flags)."
    src = once(a, "    long long probe_p1 = 0; if (probe) probe_p1 = (long long)__builtin_readcyclecounter();\n" + a)
    b = "    // ---- epilogue.  C/D layout of the 32x32 MFMA"
    src = once(b, "    long long probe_p2 = 0; if (probe) probe_p2 = (long long)__builtin_readcyclecounter();\n" + b)
    c = "            }\n        }\n    if (probe) {\n        atomicAdd(reinterpret_cast<unsigned long long *>(p.clock_probe),"
    src = once(c, "            }\n        }\n    if (probe) {\n        const long long probe_e = (long long)__builtin_readcyclecounter();\n"
               "        atomicAdd(reinterpret_cast<unsigned long long *>(p.clock_probe) + 2, (unsigned long long)(probe_p1 - probe_c));\n"
               "        atomicAdd(reinterpret_cast<unsigned long long *>(p.clock_probe) + 3, (unsigned long long)(probe_p2 - probe_p1));\n"
               "        atomicAdd(reinterpret_cast<unsigned long long *>(p.clock_probe) + 4, (unsigned long long)(probe_e - probe_p2));\n"
               "        atomicAdd(reinterpret_cast<unsigned long long *>(p.clock_probe) + 5, 1ull);\n"
               "        atomicAdd(reinterpret_cast<unsigned long long *>(p.clock_probe) + 6, (unsigned long long)nsl);\n"
               "        atomicAdd(reinterpret_cast<unsigned long long *>(p.clock_probe),")
    open(os.path.join(csrc, '_probe_gconv.hip'), 'w').write(src)
    from hplflownet_amd import build
    build.build()
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    for n in ('_probe_gconv3', '_probe_gconv'):
        subprocess.check_call([hipcc] + build.FLAGS + ['-c', os.path.join(csrc, n + '.hip'), '-o', os.path.join(csrc, n + '.o')])
    objs = [os.path.join(csrc, f.replace('.hip', '.o')) for f in build.SOURCES if f not in ('gconv3.hip', 'gconv.hip')] + \
        [os.path.join(csrc, '_probe_gconv3.o'), os.path.join(csrc, '_probe_gconv.o')]
    out = os.path.join(ROOT, 'hplflownet_amd', 'libhplbcl_probe.so')
    subprocess.check_call([hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', out] + objs)
    print(out)


if len(sys.argv) > 1 and sys.argv[1] == 'build':
    build_probe()
    sys.exit(0)
import torch
import hplflownet_amd as H
from hplflownet_amd import ops, _lib
from hplflownet_amd.synthetic import SCALES_FILTER_MAP, synthetic_pair

dev = 'cuda'
pc1, pc2, sf = synthetic_pair(8192, 0)
gen = H.GenerateDataUnsymmetric(types.SimpleNamespace(dim=3, scales_filter_map=SCALES_FILTER_MAP), device=dev)
t1 = torch.from_numpy(pc1.T.copy()).to(dev); t2 = torch.from_numpy(pc2.T.copy()).to(dev)
lat = gen.build(t1, t2)
PROBE = torch.zeros(8, dtype=torch.int64, device=dev)
lib = _lib.load()
orig = lib.hpl_gconv_forward
on = [False]


def wrapped(dref, st):
    if on[0]:
        dref._obj.clock_probe = PROBE.data_ptr()
    return orig(dref, st)


lib.hpl_gconv_forward = wrapped
cases = [('bcn1_ g0', 0, 580, 1024, 0, 8, None), ('bcn1_ g1', 0, 580, 1024, 8, 15, None), ('bcn2_ g0', 1, 324, 512, 0, 8, None), ('bcn2_ g1', 1, 324, 512, 8, 15, None),
         ('1x1 25841x1024x1024', -1, 1024, 1024, 0, 1, 25841), ('1x1 8192x1024x1024', -1, 1024, 1024, 0, 1, 8192), ('1x1 8192x1024x512', -1, 1024, 512, 0, 1, 8192)]
print('%-22s %8s %6s | per probed tile, shader cycles: %9s %9s %9s %9s | %8s %8s' % ('launch', 'us', 'tiles', 'prologue', 'loop', 'epilogue', 'total', 'slices', 'GHz'))
for name, lvl, C, O, f0, f1, Md in cases:
    F = f1 - f0
    if lvl >= 0:
        nbr = lat.levels[lvl].blur[0].t[f0:f1]
        M = nbr.shape[1]
        perm = ops.tap_order(nbr)
        t128 = ops.tile_index(nbr, perm, BM=128)
    else:
        nbr, M, perm, t128 = None, Md, None, None
    A = torch.randn(M, C, device=dev)
    Wt = torch.zeros(ops.round_up(F * C, 32), O, device=dev)
    Wt[:F * C] = torch.randn(F * C, O, device=dev) / (F * C) ** 0.5
    W3 = ops.weight_split3(Wt)
    y = torch.empty(M, O, device=dev)
    fn = lambda: ops.gconv_raw(A, nbr, M, C, F, Wt, O, out=y, row_perm=perm, tiles=t128, split_k=False, Wt3=W3)
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    PROBE.zero_()
    on[0] = True
    s.record(); fn(); e.record(); torch.cuda.synchronize()
    on[0] = False
    p = PROBE.tolist()
    n = max(1, p[5])
    ghz = p[0] / max(1, p[1]) * 0.1          # cycles per 100-MHz tick
    print('%-22s %8.1f %6d | %43s %9.0f %9.0f %9.0f %9.0f | %8.1f %8.2f' % (name, s.elapsed_time(e) * 1e3, -(-M // 128) * -(-O // 256), '', p[2] / n, p[3] / n, p[4] / n, p[0] / n, p[6] / n, ghz), flush=True)

# ---- the fp32 kernel (csrc/gconv.hip) on the narrow stencils of the Down path: both clouds of a level, 64 -> 64 channels, 15 taps
print()
print('%-22s %8s %6s | per probed tile, shader cycles: %9s %9s %9s %9s | %8s %8s' % ('fp32 launch', 'us', 'wgs', 'prologue', 'loop', 'epilogue', 'total', 'slices', 'GHz'))
for lvl in (0, 1, 2, 3, 4):
    tb = lat.levels[lvl].blur.pair
    nbr = tb.t
    F, M = nbr.shape
    C = O = 64
    perm = ops.tap_order(nbr) if M >= 16384 else None
    t64 = ops.tile_index(nbr, perm, BM=64) if perm is not None else None
    A = torch.randn(M, C, device=dev)
    Wt = torch.zeros(ops.round_up(F * C, 32), O, device=dev)
    Wt[:F * C] = torch.randn(F * C, O, device=dev) / (F * C) ** 0.5
    y = torch.empty(M, O, device=dev)
    for sk in (False, True):
        fn = lambda: ops.gconv_raw(A, nbr, M, C, F, Wt, O, out=y, row_perm=perm, tiles=t64, split_k=sk)
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        PROBE.zero_()
        on[0] = True
        s.record(); fn(); e.record(); torch.cuda.synchronize()
        on[0] = False
        p = PROBE.tolist()
        n = max(1, p[5])
        print('%-22s %8.1f %6s | %43s %9.0f %9.0f %9.0f %9.0f | %8.1f %8.2f' % ('level %d%s M=%d' % (lvl, ' split-K' if sk else '', M), s.elapsed_time(e) * 1e3, '', '',
                                                                               p[2] / n, p[3] / n, p[4] / n, p[0] / n, p[6] / n, p[0] / max(1, p[1]) * 0.1), flush=True)
