"""Build libhplbcl.so (the HIP kernels + C ABI) in-tree with hipcc for gfx950.

hipcc cross-compiles without a GPU.  The shared object is git-ignored but travels to
the GPU box with the working-tree snapshot.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIB = os.path.join(HERE, 'libhplbcl.so')
DIAG_LIB = os.path.join(HERE, 'libhplbcl_diag.so')
SOURCES = ['index_ops.hip', 'row_order.hip', 'splat_slice.hip', 'train_ops.hip', 'gconv.hip', 'gconv3.hip', 'wgrad3.hip', 'lattice.hip', 'lattice_fused.hip', 'executor.hip', 'lattice_builder.hip']
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-munsafe-fp-atomics',
         '-ffp-contract=off',   # every fused multiply-add in the kernels is an explicit fmaf
         '-Wall', '-Wno-unused-function']


def needs_build():
    if not os.path.exists(LIB) or not os.path.exists(DIAG_LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + \
           [os.path.join(os.path.dirname(HERE), 'include', 'hpl_bcl.h'), os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    if not force and not needs_build():
        return LIB
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    objs = []
    procs = []
    for src in SOURCES:
        path = os.path.join(CSRC, src)
        if not os.path.exists(path):
            continue
        obj = os.path.join(CSRC, src.replace('.hip', '.o'))
        cmd = [hipcc] + FLAGS + ['-c', path, '-o', obj]
        if verbose:
            print(' '.join(cmd))
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
        objs.append(obj)
    failed = False
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0 or (verbose and out):
            sys.stderr.write(out.decode())
        failed |= p.returncode != 0
    if failed:
        raise RuntimeError('hipcc failed')
    subprocess.check_call([hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs)
    # measurement helpers of bench.py / tools (include/hpl_diag.h): a library of their own, nothing of the hot path loads it
    subprocess.check_call([hipcc] + FLAGS + ['-shared', os.path.join(CSRC, 'diag.hip'), '-o', DIAG_LIB])
    return LIB


if __name__ == '__main__':
    print(build(force='--force' in sys.argv, verbose=True))
