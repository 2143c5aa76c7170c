"""ctypes binding of libhplbcl.so (include/hpl_bcl.h).

torch is used for device memory and streams only: every call passes raw device
pointers (tensor.data_ptr()) and the current HIP stream.  There is NO fallback: if the
shared library is missing or a call fails, an exception is raised.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'libhplbcl.so')

c_i64, c_i32, c_f32, c_vp = ctypes.c_int64, ctypes.c_int32, ctypes.c_float, ctypes.c_void_p


class HplError(RuntimeError):
    pass


class RelayoutJob(ctypes.Structure):
    """Mirror of `hpl_relayout_job` (include/hpl_bcl.h)."""
    _fields_ = [('W', ctypes.c_void_p), ('base', ctypes.c_int64), ('sr', ctypes.c_int64), ('sq', ctypes.c_int64),
                ('sf', ctypes.c_int64), ('R', ctypes.c_int32), ('Q', ctypes.c_int32), ('F', ctypes.c_int32),
                ('mirror', ctypes.c_int32), ('ldw', ctypes.c_int64)]


class Split3Job(ctypes.Structure):
    """Mirror of `hpl_split3_job`."""
    _fields_ = [('Wt', ctypes.c_void_p), ('dst', ctypes.c_void_p), ('k_rows', ctypes.c_int64), ('ldw', ctypes.c_int64),
                ('plane_stride', ctypes.c_int64), ('amax', ctypes.c_void_p), ('planes', ctypes.c_int32), ('pad_', ctypes.c_int32)]


class GConvDesc(ctypes.Structure):
    """Mirror of `struct hpl_gconv_desc` (include/hpl_bcl.h)."""
    _fields_ = [('A', c_vp), ('lda', c_i64), ('rows_a', c_i64),
                ('nbr', c_vp), ('nbr_stride', c_i64), ('reg_stride', c_i64),
                ('M', c_i64), ('C', c_i32), ('F', c_i32),
                ('Wt', c_vp), ('ldw', c_i64), ('N', c_i32), ('act', c_i32), ('slope', c_f32),
                ('bias', c_vp), ('res', c_vp), ('ldres', c_i64), ('res_mod', c_i64),
                ('Y', c_vp), ('ldy', c_i64),
                ('scat', c_vp), ('scat_stride', c_i64), ('scat_c', c_i32), ('w_rows', c_i32),
                ('row_perm', c_vp), ('ws', c_vp), ('ws_bytes', c_i64),
                ('tile_idx', c_vp), ('tile_mask', c_vp), ('tile_bm', c_i32), ('clock_probe', c_vp),
                ('Y2', c_vp), ('ldy2', c_i64), ('rows2', c_i64), ('Wt3', c_vp), ('wt3_plane_stride', c_i64),
                ('wt3_planes', c_i32), ('a_amax', c_vp), ('w_amax', c_vp), ('y_amax', c_vp),
                ('a_guard', c_vp), ('y_guard', c_vp), ('guard_trips', c_vp)]


class Ref(ctypes.Structure):
    """Mirror of `hpl_ref`."""
    _fields_ = [('buf', c_i32), ('row_off_sym', c_i32), ('rows_sym', c_i32), ('col_off', c_i32), ('cols', c_i32)]


class Buf(ctypes.Structure):
    """Mirror of `hpl_buf`."""
    _fields_ = [('rows_sym', c_i32), ('cols', c_i32)]


class Weight(ctypes.Structure):
    """Mirror of `hpl_weight`."""
    _fields_ = [('Wt', c_vp), ('ldw', c_i64), ('rows', c_i64), ('Wt3', c_vp), ('wt3_plane_stride', c_i64),
                ('w_amax', c_vp), ('wt3_planes', c_i32), ('pad_', c_i32)]


class Op(ctypes.Structure):
    """Mirror of `hpl_op`."""
    _fields_ = [('kind', c_i32), ('tag', c_i32), ('a', Ref), ('out', Ref), ('res', Ref), ('m_sym', c_i32),
                ('res_mod_sym', c_i32), ('level', c_i32), ('table', c_i32), ('order', c_i32), ('F', c_i32), ('C', c_i32),
                ('N', c_i32), ('weight', c_i32), ('bias', c_i32), ('act', c_i32), ('slope', c_f32), ('use_norm', c_i32),
                ('reg_stride_sym', c_i32), ('ext', c_i32), ('cond', c_i32), ('cond_level', c_i32), ('out2', Ref),
                ('rows2_sym', c_i32), ('b', Ref), ('flags', c_i32), ('aux', c_i32)]


class LevelTables(ctypes.Structure):
    """Mirror of `hpl_level_tables`."""
    _fields_ = [('n0', c_i64), ('n1', c_i64), ('H0', c_i64), ('H1', c_i64), ('emg_pair', c_vp),
                ('csr_ptr', c_vp), ('csr_pt', c_vp), ('csr_w', c_vp), ('csr_norm', c_vp), ('bary0', c_vp), ('off0', c_vp),
                ('blur', c_vp), ('blur_stride', c_i64), ('blur_perm', c_vp), ('up_perm', c_vp), ('n_up_groups', c_i32),
                ('up_group_cut', c_i32 * 5), ('up_group_perm', c_vp * 4), ('corr1', c_vp), ('corr1_stride', c_i64),
                ('corr1_perm', c_vp), ('corr2', c_vp), ('tile_bm', c_i32), ('group_tile_bm', c_i32),
                ('blur_perm_tidx', c_vp), ('blur_perm_tmask', c_vp), ('up_perm_tidx', c_vp), ('up_perm_tmask', c_vp),
                ('up_group_tidx', c_vp * 4), ('up_group_tmask', c_vp * 4), ('corr1_perm_tidx', c_vp),
                ('corr1_perm_tmask', c_vp), ('bary1', c_vp), ('off1', c_vp), ('up_tap_m', c_vp), ('up_tap_row', c_vp),
                ('up_tap_ptr', c_vp), ('up_tap_max', c_i64)]


class LatticeSpec(ctypes.Structure):
    """Mirror of `hpl_lattice_spec`."""
    _fields_ = [('n_levels', c_i32), ('scale', c_f32 * 8), ('bcn_radius', c_i32 * 8), ('corr_filter_radius', c_i32 * 8),
                ('corr_corr_radius', c_i32 * 8), ('next_divisor', c_f32 * 8), ('wide_up', c_i32 * 8), ('n_groups', c_i32),
                ('group_cut', c_i32 * 5), ('groups_min_sparsity', c_f32), ('perm_min_rows', c_i64), ('groups_min_rows', c_i64),
                ('group_tile_bm', c_i32),
                ('fused', c_i32)]


_SIGNATURES = {
    'hpl_version': (ctypes.c_int, []),
    'hpl_last_error': (ctypes.c_char_p, []),
    'hpl_device_info': (ctypes.c_int, [ctypes.c_int, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int),
                                       ctypes.c_char_p, ctypes.c_int]),
    'hpl_index_narrow': (ctypes.c_int, [c_vp, c_vp, c_i64, c_vp]),
    'hpl_corr2_permute': (ctypes.c_int, [c_vp, c_vp, ctypes.c_int, ctypes.c_int, c_i64, c_vp]),
    'hpl_corr2_permute32': (ctypes.c_int, [c_vp, c_vp, ctypes.c_int, ctypes.c_int, c_i64, c_vp]),
    'hpl_csr_build': (ctypes.c_int, [c_vp, c_vp, c_i64, c_i64, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    'hpl_csr_build_pair': (ctypes.c_int, [c_vp, c_vp, c_i64, c_i64, c_vp, c_vp, c_i64, c_i64, c_vp, c_vp, c_vp, c_vp,
                                          c_vp, c_vp]),
    'hpl_splat': (ctypes.c_int, [c_vp, c_i64, ctypes.c_int, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp, c_i64, c_vp]),
    'hpl_slice': (ctypes.c_int, [c_vp, c_i64, ctypes.c_int, c_vp, c_vp, c_i64, c_vp, c_vp, c_vp, c_i64, c_vp]),
    'hpl_splat_add': (ctypes.c_int, [c_vp, c_i64, ctypes.c_int, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp, c_i64, c_vp]),
    'hpl_slice_add': (ctypes.c_int, [c_vp, c_i64, ctypes.c_int, c_vp, c_vp, c_i64, c_vp, c_vp, c_vp, c_i64, c_vp]),
    'hpl_weight_unlayout_batch': (ctypes.c_int, [c_vp, ctypes.c_int, c_vp, c_i64, c_vp, c_vp]),
    'hpl_psum': (ctypes.c_int, [c_vp, c_i64, c_i64, c_i64, ctypes.c_int, c_vp, c_i64, ctypes.c_int, c_vp]),
    'hpl_regroup': (ctypes.c_int, [c_vp, c_i64, c_i64, ctypes.c_int, ctypes.c_int, c_vp, c_i64, ctypes.c_int, c_vp]),
    'hpl_epe3d': (ctypes.c_int, [c_vp, c_vp, c_i64, c_vp, c_vp, c_vp]),
    'hpl_adam_flat': (ctypes.c_int, [c_vp, c_vp, c_vp, c_vp, c_i64, ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_double,
                                     c_i64, c_vp]),
    'hpl_gather_sum': (ctypes.c_int, [c_vp, c_i64, c_vp, c_i64, c_i64, ctypes.c_int, ctypes.c_int, ctypes.c_int, c_vp, c_vp, c_i64, c_i64,
                                      ctypes.c_int, c_f32, c_vp, c_i64, c_vp]),
    'hpl_table_invert': (ctypes.c_int, [c_vp, c_i64, ctypes.c_int, c_i64, ctypes.c_int, c_i64, c_vp, c_vp]),
    'hpl_weight_relayout': (ctypes.c_int, [c_vp, c_i64, ctypes.c_int, ctypes.c_int, ctypes.c_int, c_i64, c_i64,
                                           c_i64, c_vp, c_vp, c_i64, c_i64, c_vp]),
    'hpl_weight_relayout_batch': (ctypes.c_int, [c_vp, ctypes.c_int, c_vp, c_i64, c_vp, c_vp]),
    'hpl_weight_split3': (ctypes.c_int, [c_vp, c_i64, c_i64, c_vp, c_i64, c_vp]),
    'hpl_weight_split3_batch': (ctypes.c_int, [c_vp, ctypes.c_int, c_i64, ctypes.c_int, c_vp]),
    'hpl_weight_split2h': (ctypes.c_int, [c_vp, c_i64, c_i64, c_vp, c_i64, c_vp, c_vp]),
    'hpl_amax': (ctypes.c_int, [c_vp, c_i64, c_i64, c_i32, c_vp, c_vp]),
    'hpl_amax_rows': (ctypes.c_int, [c_vp, c_i64, c_i64, c_i32, c_vp, c_vp, c_vp]),
    'hpl_gconv_wgrad_scaled': (ctypes.c_int, [c_vp, c_i64, c_i64, c_vp, c_i64, c_i64, c_i64, ctypes.c_int, ctypes.c_int,
                                              c_vp, c_i64, ctypes.c_int, c_vp, c_i64, c_vp, c_vp, c_vp, c_i64, c_vp, c_vp, c_vp, c_vp]),
    'hpl_weight_unlayout': (ctypes.c_int, [c_vp, c_i64, ctypes.c_int, ctypes.c_int, ctypes.c_int, c_vp, c_i64,
                                           c_i64, c_i64, c_i64, ctypes.c_int, c_vp]),
    'hpl_tap_order_scratch_ints': (c_i64, [c_i64]),
    'hpl_tap_order': (ctypes.c_int, [c_vp, c_i64, ctypes.c_int, c_i64, c_vp, c_vp, c_vp]),
    'hpl_tile_index': (ctypes.c_int, [c_vp, c_i64, ctypes.c_int, c_i64, c_vp, ctypes.c_int, c_vp, c_vp, c_vp]),
    'hpl_gconv_forward': (ctypes.c_int, [ctypes.POINTER(GConvDesc), c_vp]),
    'hpl_gconv_forward_naive': (ctypes.c_int, [ctypes.POINTER(GConvDesc), c_vp]),
    'hpl_gconv_wgrad': (ctypes.c_int, [c_vp, c_i64, c_i64, c_vp, c_i64, c_i64, c_i64, ctypes.c_int, ctypes.c_int,
                                       c_vp, c_i64, ctypes.c_int, c_vp, c_i64, c_vp, c_vp, c_vp, c_i64, c_vp, c_vp]),
    'hpl_tap_lists': (ctypes.c_int, [c_vp, c_i64, ctypes.c_int, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp]),
    'hpl_colsum': (ctypes.c_int, [c_vp, c_i64, c_i64, ctypes.c_int, c_vp, c_vp]),
    'hpl_leaky_bwd': (ctypes.c_int, [c_vp, c_i64, c_vp, c_i64, c_f32, c_vp, c_i64, c_i64, ctypes.c_int, c_vp]),
    'hpl_leaky_bwd_amax': (ctypes.c_int, [c_vp, c_i64, c_vp, c_i64, c_f32, c_vp, c_i64, c_i64, ctypes.c_int, c_vp, c_vp]),
    'hpl_transpose': (ctypes.c_int, [c_vp, c_i64, c_vp, c_i64, c_i64, c_i64, c_vp]),
    'hpl_table_symmetric': (ctypes.c_int, [c_vp, c_i64, ctypes.c_int, c_i64, c_vp, c_vp]),
    'hpl_lattice_keys': (ctypes.c_int, [c_vp, c_i64, c_f32, c_vp, c_vp, c_vp, c_i64, c_vp]),
    'hpl_lattice_keys_pair': (ctypes.c_int, [c_vp, c_vp, c_vp, c_vp, c_i64, c_i64, c_f32, c_i64, c_i64, c_f32, c_vp, c_vp,
                                             c_vp, c_vp, c_vp, c_vp, c_i64, c_vp]),
    'hpl_lattice_workspace_bytes': (c_i64, [c_i64, c_i64]),
    'hpl_lattice_hash': (ctypes.c_int, [c_vp, c_i64, c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp]),
    'hpl_lattice_neighbors': (ctypes.c_int, [c_vp, c_i64, c_i64, c_vp, c_vp, c_i64, c_i64, ctypes.c_int,
                                             ctypes.c_int, ctypes.c_int, c_vp, c_vp, c_i64, c_i64, c_vp, c_vp, c_vp]),
    'hpl_lattice_next_points': (ctypes.c_int, [c_vp, c_i64, c_i64, c_f32, c_vp, c_vp]),
    'hpl_lattice_create': (c_vp, [ctypes.POINTER(LatticeSpec)]),
    'hpl_lattice_destroy': (None, [c_vp]),
    'hpl_lattice_arena_bytes': (c_i64, [c_vp, c_i64, c_i64]),
    'hpl_lattice_set_bounds': (ctypes.c_int, [c_vp, ctypes.POINTER(c_i64)]),
    'hpl_lattice_stats': (ctypes.c_int, [c_vp, ctypes.POINTER(c_i32)]),
    'hpl_lattice_begin': (ctypes.c_int, [c_vp, c_vp, c_vp, c_i64, c_i64, c_vp, c_i64, c_vp]),
    'hpl_lattice_ready': (ctypes.c_int, [c_vp]),
    'hpl_lattice_advance': (ctypes.c_int, [c_vp, ctypes.POINTER(ctypes.c_int)]),
    'hpl_lattice_tables': (ctypes.POINTER(LevelTables), [c_vp]),
    'hpl_lattice_extras': (ctypes.c_int, [c_vp, ctypes.POINTER(c_vp), ctypes.POINTER(c_i64)]),
    'hpl_plan_create': (c_vp, [ctypes.POINTER(Op), ctypes.c_int, ctypes.POINTER(Buf), ctypes.c_int,
                               ctypes.POINTER(Weight), ctypes.c_int, ctypes.POINTER(c_vp), ctypes.c_int]),
    'hpl_plan_destroy': (None, [c_vp]),
    'hpl_plan_workspace_bytes': (c_i64, [c_vp, ctypes.POINTER(LevelTables), ctypes.c_int]),
    'hpl_plan_run': (ctypes.c_int, [c_vp, ctypes.POINTER(LevelTables), ctypes.c_int, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp]),
    'hpl_plan_run_range': (ctypes.c_int, [c_vp, ctypes.POINTER(LevelTables), ctypes.c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64,
                                          c_vp, c_vp, ctypes.c_int, ctypes.c_int, ctypes.c_int]),
    'hpl_plan_set_unlayout': (ctypes.c_int, [c_vp, c_vp, c_vp, c_vp, ctypes.POINTER(c_i32), ctypes.POINTER(c_i64), ctypes.c_int]),
    'hpl_plan_profile': (ctypes.c_int, [c_vp, ctypes.c_int]),
    'hpl_plan_guard_trips': (ctypes.c_int, [c_vp, ctypes.POINTER(c_i64)]),
    'hpl_plan_clock_probe': (ctypes.c_int, [c_vp, c_vp]),
    'hpl_plan_profile_read': (ctypes.c_int, [c_vp, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(c_f32)]),
}

#: every symbol include/hpl_bcl.h declares (checked by tests/test_abi.py against the header text)
EXPORTS = tuple(_SIGNATURES)

_lib = None


def load():
    """Load libhplbcl.so; raises HplError if it has not been built (no CPU fallback exists)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise HplError('%s is missing: run `python -c "import __graft_entry__ as g; g.build()"` '
                           '(hipcc --offload-arch=gfx950). There is no CPU fallback.' % LIB_PATH)
        # HPL_LIB: load an experimental build of the same ABI instead (kernel A/B tests on one box)
        lib = ctypes.CDLL(os.environ.get('HPL_LIB') or LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(lib, name)      # AttributeError here = ABI/header mismatch
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


DIAG_PATH = os.path.join(_HERE, 'libhplbcl_diag.so')
_diag = None


def load_diag():
    """libhplbcl_diag.so (include/hpl_diag.h): the MFMA rate probes of bench.py / tools -- not part of the product library."""
    global _diag
    if _diag is None:
        if not os.path.exists(DIAG_PATH):
            raise HplError('%s is missing: run `python -c "import __graft_entry__ as g; g.build()"`' % DIAG_PATH)
        lib = ctypes.CDLL(DIAG_PATH)
        lib.hpl_mfma_probe.restype = ctypes.c_int
        lib.hpl_mfma_probe.argtypes = [c_vp, ctypes.c_int, ctypes.c_int, c_vp]
        lib.hpl_mfma_probe_data.restype = ctypes.c_int
        lib.hpl_mfma_probe_data.argtypes = [c_vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, c_vp, c_vp]
        lib.hpl_diag_splat_atomic.restype = ctypes.c_int
        lib.hpl_diag_splat_atomic.argtypes = [c_vp, ctypes.c_int64, ctypes.c_int, c_vp, c_vp, ctypes.c_int64, c_vp, ctypes.c_int64, c_vp,
                                              ctypes.c_int64, ctypes.c_int, c_vp]
        lib.hpl_diag_chain.restype = ctypes.c_int
        lib.hpl_diag_chain.argtypes = [c_vp, c_vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, c_vp, ctypes.c_int, c_vp]
        _diag = lib
    return _diag


def check(rc, what):
    if rc != 0:
        raise HplError('%s failed (%d): %s' % (what, rc, load().hpl_last_error().decode()))


_RAW_STREAM = getattr(torch._C, '_cuda_getCurrentRawStream', None)
_GET_DEVICE = getattr(torch._C, '_cuda_getDevice', None)


def stream():
    """hipStream_t of torch's current stream on the current device, as an integer.  The raw C
    accessors cost ~0.3 us; torch.cuda.current_stream() builds a Stream object (~6 us, 130 calls per
    forward)."""
    if _RAW_STREAM is not None and _GET_DEVICE is not None:
        return _RAW_STREAM(_GET_DEVICE())
    return torch.cuda.current_stream().cuda_stream


def ptr(t):
    """Device pointer of a tensor (or None)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise HplError('expected a device tensor, got %s (the HIP path has no CPU fallback)' % t.device)
    return t.data_ptr()
