"""Import the read-only reference (`/root/reference`) inside THIS container.

The reference's lattice module imports `numba` and the CFFI-built `_khash_ffi`
at load time (`transforms/transforms.py:12-24`); neither exists here.  Two
in-memory shims make it importable without touching the reference tree:
  * `numba.njit` -> identity decorator (the njit bodies are plain numpy/Python);
  * `_khash_ffi.lib` -> either a Python dict or the reference's own khash
    compiled by `oracle/Makefile` into `oracle/_ref/libkhash_ref.so` (ctypes).
Only get/set semantics are observable, so both give the same
integers.  This file is tooling for fixture generation; it never ships to the
GPU box as a dependency (the reference does not exist there).
"""
import ctypes
import os
import sys
import types

REF = '/root/reference'


class _DictKhash:
    def __init__(self):
        self.tables = {}
        self.next = 1

    def khash_int2int_init(self):
        h = self.next
        self.next += 1
        self.tables[h] = {}
        return h

    def khash_int2int_get(self, h, k, d):
        return self.tables[h].get(int(k), d)

    def khash_int2int_set(self, h, k, v):
        self.tables[h][int(k)] = int(v)
        return 0

    def khash_int2int_destroy(self, h):
        self.tables.pop(h, None)


class _CtypesKhash:
    """The reference's own khash (models/khash_int2int.h) built into oracle/_ref."""

    def __init__(self, path):
        lib = ctypes.CDLL(path)
        lib.khash_ref_init.restype = ctypes.c_void_p
        lib.khash_ref_get.restype = ctypes.c_int64
        lib.khash_ref_get.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64]
        lib.khash_ref_set.restype = ctypes.c_int
        lib.khash_ref_set.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64]
        lib.khash_ref_destroy.argtypes = [ctypes.c_void_p]
        self.lib = lib

    def khash_int2int_init(self):
        return self.lib.khash_ref_init()

    def khash_int2int_get(self, h, k, d):
        return self.lib.khash_ref_get(h, int(k), int(d))

    def khash_int2int_set(self, h, k, v):
        return self.lib.khash_ref_set(h, int(k), int(v))

    def khash_int2int_destroy(self, h):
        self.lib.khash_ref_destroy(h)


def install_shims(khash='dict'):
    def njit(*a, **k):
        if len(a) == 1 and isinstance(a[0], types.FunctionType) and not k:
            return a[0]
        return lambda f: f

    class _Ty:
        def __getitem__(self, item):
            return self

        def __call__(self, *a, **k):
            return self

    nb = types.ModuleType('numba')
    nb.njit = njit
    nb.int64 = _Ty()
    nb.cffi_support = types.SimpleNamespace(register_module=lambda m: None)
    sys.modules['numba'] = nb
    kh = types.ModuleType('_khash_ffi')
    if khash == 'dict':
        kh.lib = _DictKhash()
    else:
        here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        kh.lib = _CtypesKhash(os.path.join(here, 'oracle', '_ref', 'libkhash_ref.so'))
    sys.modules['_khash_ffi'] = kh
    if REF not in sys.path:
        sys.path.insert(0, REF)


def import_reference(khash='dict'):
    install_shims(khash)
    import models  # noqa: F401  (reference package)
    from models.bilateralNN import BilateralConvFlex, sparse_sum
    from models.bnn_flow import BilateralCorrelationFlex
    from models.HPLFlowNet import HPLFlowNet
    from models.HPLFlowNet_shallow import HPLFlowNetShallow
    import transforms.transforms as T
    return types.SimpleNamespace(BilateralConvFlex=BilateralConvFlex, sparse_sum=sparse_sum,
                                 BilateralCorrelationFlex=BilateralCorrelationFlex,
                                 HPLFlowNet=HPLFlowNet, HPLFlowNetShallow=HPLFlowNetShallow, T=T)
