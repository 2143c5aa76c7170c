"""ctypes driver for the C lattice oracle (oracle/lattice_oracle.c).

TEST INFRASTRUCTURE ONLY -- see the header of lattice_oracle.c.  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.

`generate_data` restates the multi-scale driver
`GenerateDataUnsymmetric.__call__` (/root/reference/transforms/transforms.py:358-485)
over the C functions and returns the same `generated_data` schema
 as numpy arrays.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

_i64p = np.ctypeslib.ndpointer(dtype=np.int64, flags='C_CONTIGUOUS')
_f32p = np.ctypeslib.ndpointer(dtype=np.float32, flags='C_CONTIGUOUS')


def build(force=False):
    """Compile the oracle (and oracle/_ref when /root/reference is present)."""
    so = os.path.join(_HERE, 'liblattice_oracle.so')
    src = os.path.join(_HERE, 'lattice_oracle.c')
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(['make', '-C', _HERE, 'all'], stdout=subprocess.DEVNULL)
    elif os.path.isdir('/root/reference/models') and not os.path.exists(
            os.path.join(_HERE, '_ref', 'libkhash_ref.so')):
        subprocess.check_call(['make', '-C', _HERE, 'ref'], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = ctypes.CDLL(build())
        L.hpl_i2i_init.restype = ctypes.c_void_p
        L.hpl_i2i_destroy.argtypes = [ctypes.c_void_p]
        L.hpl_i2i_get.restype = ctypes.c_int64
        L.hpl_i2i_get.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64]
        L.hpl_i2i_set.restype = ctypes.c_int
        L.hpl_i2i_set.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64]
        L.hpl_i2i_size.restype = ctypes.c_int64
        L.hpl_i2i_size.argtypes = [ctypes.c_void_p]
        L.hpl_key2int.restype = ctypes.c_int64
        L.hpl_key2int.argtypes = [_i64p, _i64p, _i64p]
        L.hpl_int2key.argtypes = [ctypes.c_int64, _i64p, _i64p, _i64p]
        L.hpl_neighbor_offsets.restype = ctypes.c_int
        L.hpl_neighbor_offsets.argtypes = [ctypes.c_int, _i64p]
        L.hpl_elevate_matrix.argtypes = [_f32p]
        L.hpl_expected_std.restype = ctypes.c_double
        L.hpl_keys_and_barycentric.argtypes = [_f32p, ctypes.c_int64, _i64p, _f32p, _f32p]
        L.hpl_key_minmax.argtypes = [_i64p, ctypes.c_int64, _i64p, _i64p]
        L.hpl_count_unique_keys.restype = ctypes.c_int64
        L.hpl_count_unique_keys.argtypes = [_i64p, ctypes.c_int64]
        L.hpl_build_unsymmetric.argtypes = [
            ctypes.c_int64, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int,
            _i64p, _i64p, _i64p, _i64p, _i64p, _i64p, _i64p,
            _i64p, ctypes.c_int64, _i64p, ctypes.c_int64,
            _i64p, _i64p, _i64p, _i64p, _f32p, _f32p, ctypes.c_int,
            ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int64)]
        L.hpl_next_level_points.argtypes = [_f32p, ctypes.c_int64, ctypes.c_double, _f32p]
        _LIB = L
    return _LIB


def filter_size(radius, d1=4):
    return (radius + 1) ** d1 - radius ** d1


def neighbor_offsets(radius):
    out = np.zeros((filter_size(radius), 4), dtype=np.int64)
    n = lib().hpl_neighbor_offsets(radius, out)
    assert n == out.shape[0]
    return out


def elevate_matrix():
    E = np.zeros((4, 3), dtype=np.float32)
    lib().hpl_elevate_matrix(E)
    return E


def keys_and_barycentric(pc):
    """pc (3, N) float32 (already scaled) -> keys (4,N,4) i64, bary (4,N), emg (4,N)."""
    pc = np.ascontiguousarray(pc, dtype=np.float32)
    n = pc.shape[1]
    keys = np.empty((4, n, 4), dtype=np.int64)
    bary = np.empty((4, n), dtype=np.float32)
    emg = np.empty((4, n), dtype=np.float32)
    lib().hpl_keys_and_barycentric(pc, n, keys, bary, emg)
    return keys, bary, emg


def build_level(keys1, keys2, radii, assign_last):
    """One call of build_unsymmetric (+ min/max, unique counts). Returns a dict."""
    L = lib()
    n1, n2 = keys1.shape[1], keys2.shape[1]
    mn1, mx1, mn2, mx2 = (np.empty(4, np.int64) for _ in range(4))
    L.hpl_key_minmax(keys1, n1, mn1, mx1)
    L.hpl_key_minmax(keys2, n2, mn2, mx2)
    key_mins, key_maxs = np.minimum(mn1, mn2), np.maximum(mx1, mx2)
    h1 = int(L.hpl_count_unique_keys(keys1, n1))
    h2 = int(L.hpl_count_unique_keys(keys2, n2))
    bcn_r, cf_r, cc_r = radii
    off1 = np.empty((4, n1), np.int64)
    off2 = np.empty((4, n2), np.int64)
    dummy = np.zeros((1, 1), np.int64)
    if bcn_r != -1:
        bfs = filter_size(bcn_r)
        blur1 = np.full((bfs, h1), -1, np.int64)
        blur2 = np.full((bfs, h2), -1, np.int64)
        boff = neighbor_offsets(bcn_r)
    else:
        bfs, blur1, blur2, boff = -1, dummy.copy(), dummy.copy(), dummy
    if cf_r != -1:
        cfs, ccs = filter_size(cf_r), filter_size(cc_r)
        corr1 = np.full((ccs, h1), -1, np.int64)
        corr2 = np.full((cfs, ccs, h1), -1, np.int64)
        cfo, cco = neighbor_offsets(cf_r), neighbor_offsets(cc_r)
    else:
        cfs = ccs = -1
        corr1, corr2, cfo, cco = dummy.copy(), dummy.copy(), dummy, dummy
    last1 = np.zeros((4, h1), np.float32) if assign_last else np.zeros((1, 1), np.float32)
    last2 = np.zeros((4, h2), np.float32) if assign_last else np.zeros((1, 1), np.float32)
    c1, c2 = ctypes.c_int64(0), ctypes.c_int64(0)
    L.hpl_build_unsymmetric(n1, n2, bfs, cfs, ccs, keys1, keys2, key_maxs, key_mins,
                            off1, off2, boff, blur1, h1, blur2, h2, cfo, cco, corr1, corr2,
                            last1, last2, int(assign_last), ctypes.byref(c1), ctypes.byref(c2))
    assert c1.value == h1 and c2.value == h2
    return dict(off1=off1, off2=off2, blur1=blur1 if bfs != -1 else None,
                blur2=blur2 if bfs != -1 else None,
                corr1=corr1 if cfs != -1 else None, corr2=corr2 if cfs != -1 else None,
                last1=last1, last2=last2, h1=h1, h2=h2, key_mins=key_mins, key_maxs=key_maxs)


def generate_data(pc1, pc2, scales_filter_map):
    """pc1, pc2: (N, 3) float32.  Returns list of per-level dicts (numpy), schema of
    transforms.py:471-483; placeholders for absent tables are zeros(1) int64 (:450-459)."""
    L = lib()
    last1 = np.ascontiguousarray(pc1.T, dtype=np.float32).copy()
    last2 = np.ascontiguousarray(pc2.T, dtype=np.float32).copy()
    out = []
    nlev = len(scales_filter_map)
    for idx, (scale, bcn_r, cf_r, cc_r) in enumerate(scales_filter_map):
        last1 = last1 * np.float32(scale)                    # transforms.py:377-378
        last2 = last2 * np.float32(scale)
        k1, b1, e1 = keys_and_barycentric(last1)
        k2, b2, e2 = keys_and_barycentric(last2)
        assign_last = idx != nlev - 1
        lev = build_level(k1, k2, (bcn_r, cf_r, cc_r), assign_last)
        ph = np.zeros(1, np.int64)
        out.append({'pc1_barycentric': b1, 'pc2_barycentric': b2,
                    'pc1_el_minus_gr': e1, 'pc2_el_minus_gr': e2,
                    'pc1_lattice_offset': lev['off1'], 'pc2_lattice_offset': lev['off2'],
                    'pc1_blur_neighbors': lev['blur1'] if lev['blur1'] is not None else ph,
                    'pc2_blur_neighbors': lev['blur2'] if lev['blur2'] is not None else ph,
                    'pc1_corr_indices': lev['corr1'] if lev['corr1'] is not None else ph,
                    'pc2_corr_indices': lev['corr2'] if lev['corr2'] is not None else ph,
                    'pc1_hash_cnt': lev['h1'], 'pc2_hash_cnt': lev['h2']})
        if assign_last:                                      # transforms.py:461-469
            n1 = np.empty((3, lev['h1']), np.float32)
            n2 = np.empty((3, lev['h2']), np.float32)
            L.hpl_next_level_points(lev['last1'], lev['h1'], float(scale), n1)
            L.hpl_next_level_points(lev['last2'], lev['h2'], float(scale), n2)
            last1, last2 = n1, n2
    return out
