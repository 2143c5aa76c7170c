"""CPU: the C-ABI library loads and exports every symbol include/hpl_bcl.h declares; the ctypes
mirror of hpl_gconv_desc matches the header; the product path refuses to run without a device."""
import ctypes
import os
import re

import pytest
import torch

from common import ROOT
from hplflownet_amd import _lib


def header_text():
    return open(os.path.join(ROOT, 'include', 'hpl_bcl.h')).read()


def test_header_symbols_exported():
    hdr = header_text()
    body = re.sub(r'/\*.*?\*/', '', hdr, flags=re.S)
    declared = set(re.findall(r'\b(hpl_[a-z0-9_]+)\s*\(', body))
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    lib = _lib.load()                      # resolves each symbol, sets argtypes
    for name in declared:
        assert hasattr(lib, name)
    assert lib.hpl_version() >= 100


def test_library_is_not_older_than_its_sources():
    """The built library travels to the GPU box with the working tree (`gpurun`, the driver's GPU tiers): a kernel edit without a
    rebuild would be measured as the old kernel under the new source's name.  build.needs_build() compares the modification times."""
    from hplflownet_amd import build
    assert not build.needs_build(), 'hplflownet_amd/libhplbcl.so is older than csrc/ or include/hpl_bcl.h: run __graft_entry__.build()'


def test_gconv_desc_layout_matches_header():
    hdr = header_text()
    struct = hdr[hdr.index('typedef struct hpl_gconv_desc {'):hdr.index('} hpl_gconv_desc;')]
    struct = re.sub(r'/\*.*?\*/', '', struct, flags=re.S)
    fields = re.findall(r'(?:const\s+)?(?:float|int32_t|uint32_t|int64_t|void)\s*\*?\s*(\w+);', struct)
    assert fields == [f[0] for f in _lib.GConvDesc._fields_]
    assert ctypes.sizeof(_lib.GConvDesc) == (8 * 3 + 8 * 3 + 8 + 4 + 4 + 8 + 8 + 4 + 4 + 4 + 4 + 8 + 8 + 8 + 8 + 8 + 8
                                             + 8 + 8 + 4 + 4 + 8 + 8 + 8 + 8 + 8 + 4 + 4 + 8 + 8 + 8 + 8 + 8 + 8
                                                 + 8 + 8 + 8 + 8      # wt3_planes (+ padding), a_amax, w_amax, y_amax
                                                 + 8 + 8 + 8)         # a_guard, y_guard, guard_trips (round 6)


def test_relayout_job_layout_matches_header():
    hdr = header_text()
    struct = hdr[hdr.index('typedef struct hpl_relayout_job {'):hdr.index('} hpl_relayout_job;')]
    struct = re.sub(r'/\*.*?\*/', '', struct, flags=re.S)
    names = []
    for decl in re.findall(r'(?:const\s+)?(?:float|int32_t|int64_t|void)\s*\*?\s*([\w\s,]+);', struct):
        names += [n.strip() for n in decl.split(',')]
    assert names == [f[0] for f in _lib.RelayoutJob._fields_]
    assert ctypes.sizeof(_lib.RelayoutJob) == 8 + 4 * 8 + 4 * 4 + 8


def test_no_cpu_fallback():
    import hplflownet_amd as H
    m = H.BilateralConvFlex(3, 1, 8, [8], 'cpu', True, True, True, False, False, False)
    with pytest.raises(_lib.HplError):
        m(torch.zeros(1, 8, 5), None, None, torch.zeros(1, 15, 5, dtype=torch.long), None, None)
    with pytest.raises(_lib.HplError):
        H.sparse_sum(torch.zeros(1, 4, dtype=torch.long), torch.zeros(4, 3), torch.Size([5, 3]), False)


def test_argument_validation_without_gpu():
    lib = _lib.load()
    d = _lib.GConvDesc()
    assert lib.hpl_gconv_forward(ctypes.byref(d), None) == -1          # HPL_EINVAL: null pointers
    assert b'null' in lib.hpl_last_error()
    assert lib.hpl_splat(None, 0, 4, None, None, None, None, 1, None, 0, None) == -1
    assert lib.hpl_lattice_workspace_bytes(8192, 8192) > 0


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, 'hplflownet_amd')
    for fn in os.listdir(pkg):
        if fn.endswith('.py'):
            src = open(os.path.join(pkg, fn)).read()
            assert 'import oracle' not in src and 'from oracle' not in src, fn


def test_executor_struct_layouts_match_header(tmp_path):
    """The ctypes mirrors of the native executor's structs (hpl_ref, hpl_buf, hpl_weight, hpl_op, hpl_level_tables)
    against the C compiler's view of include/hpl_bcl.h: sizes and the offset of every field."""
    import shutil
    import subprocess
    if shutil.which('gcc') is None:
        pytest.skip('no C compiler')
    mirrors = {'hpl_ref': _lib.Ref, 'hpl_buf': _lib.Buf, 'hpl_weight': _lib.Weight, 'hpl_op': _lib.Op,
               'hpl_level_tables': _lib.LevelTables, 'hpl_gconv_desc': _lib.GConvDesc, 'hpl_relayout_job': _lib.RelayoutJob,
               'hpl_lattice_spec': _lib.LatticeSpec, 'hpl_split3_job': _lib.Split3Job}
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "hpl_bcl.h"', 'int main(void) {']
    for cname, cls in mirrors.items():
        lines.append('printf("%s %%zu\\n", sizeof(%s));' % (cname, cname))
        for f in cls._fields_:
            lines.append('printf("%s.%s %%zu\\n", offsetof(%s, %s));' % (cname, f[0], cname, f[0]))
    lines.append('return 0; }')
    src = tmp_path / 'layout.c'
    src.write_text('\n'.join(lines))
    exe = tmp_path / 'layout'
    subprocess.check_call(['gcc', '-I', os.path.join(ROOT, 'include'), str(src), '-o', str(exe)])
    got = dict(ln.split() for ln in subprocess.check_output([str(exe)]).decode().splitlines())
    for cname, cls in mirrors.items():
        assert int(got[cname]) == ctypes.sizeof(cls), cname
        for f in cls._fields_:
            assert int(got['%s.%s' % (cname, f[0])]) == getattr(cls, f[0]).offset, (cname, f[0])
