"""CPU: the C lattice oracle against the reference-generated golden vectors (bit-exact),
against the reference's own khash (oracle/_ref), plus the invariants of SURVEY.md §4."""
import ctypes
import os

import numpy as np
import pytest

from common import GOLD, ROOT, load_golden_lattice, oracle_lattice, sha
from hplflownet_amd.synthetic import SCALES_FILTER_MAP, synthetic_pair
from oracle import lattice_oracle as LO


def test_constants_F1():
    z = np.load(os.path.join(GOLD, 'constants.npz'))
    assert np.array_equal(LO.neighbor_offsets(1), z['offsets_r1'])
    assert np.array_equal(LO.neighbor_offsets(2), z['offsets_r2'])
    assert np.array_equal(LO.elevate_matrix(), z['elevate'])          # bitwise float32
    assert LO.lib().hpl_expected_std() == float(z['expected_std'])
    off = LO.neighbor_offsets(1)
    assert np.array_equal(off[15 - np.arange(1, 15)], -off[1:15])      # off[15-f] = -off[f]
    assert LO.filter_size(1) == 15 and LO.filter_size(2) == 65


def test_keys_and_barycentric_F2():
    z = np.load(os.path.join(GOLD, 'keys_n1024.npz'))
    pc1, pc2, _ = synthetic_pair(1024, 0)
    for s in (3, 1):
        for nm, pc in (('pc1', pc1), ('pc2', pc2)):
            p = np.ascontiguousarray(pc.T) * np.float32(s)
            keys, bary, emg = LO.keys_and_barycentric(p)
            tag = '%s_s%d' % (nm, s)
            assert np.array_equal(keys, z[tag + '_keys'])
            assert np.array_equal(bary, z[tag + '_bary'])                # bitwise
            assert np.array_equal(emg, z[tag + '_emg'])
            assert np.all(keys.sum(axis=0) == 0)                         # lattice keys sum to zero
            assert np.all(bary >= -1e-6) and np.allclose(bary.sum(0), 1, atol=1e-5)


@pytest.mark.parametrize('n', [256, 1024])
def test_generated_data_F3(n):
    gold, digest = load_golden_lattice(n)
    _, _, _, gd = oracle_lattice(n)
    for l, d in enumerate(gd):                       # every array of all 7 levels by sha256
        for k, v in d.items():
            v = np.asarray(v)
            v = v.astype(np.int64) if v.dtype.kind == 'i' else v
            assert sha(v) == digest['L%d_%s' % (l, k)], (l, k)
    for l, g in enumerate(gold):                     # levels stored in full, element-wise
        for k, v in g.items():
            assert np.array_equal(np.asarray(gd[l][k]), np.asarray(v)), (l, k)


def test_invariants_F7():
    _, _, _, gd = oracle_lattice(256)
    for l, d in enumerate(gd):
        H1 = d['pc1_hash_cnt']
        off = d['pc1_lattice_offset']
        assert off.min() == 0 and off.max() == H1 - 1
        # vertex ids appear in first-appearance order (points outer, remainder inner)
        flat = off.T.reshape(-1)
        first = np.full(H1, -1)
        seen = 0
        for v in flat:
            if first[v] < 0:
                assert v == seen
                first[v] = 1
                seen += 1
        nb = d['pc1_blur_neighbors']
        assert np.array_equal(nb[0], np.arange(H1))             # offset 0 is the vertex itself
        for f in range(1, 15):                                  # symmetry nbr[f,h]=g => nbr[15-f,g]=h
            h = np.nonzero(nb[f] >= 0)[0]
            assert np.array_equal(nb[15 - f][nb[f][h]], h)
        if SCALES_FILTER_MAP[l][2] != -1:
            assert np.array_equal(d['pc1_corr_indices'], nb)    # equal radii => same table
            assert d['pc2_corr_indices'].shape == (15, 15, H1)
            assert d['pc2_corr_indices'].max() < d['pc2_hash_cnt']


def test_key_packing_roundtrip():
    rng = np.random.RandomState(1)
    mins = np.array([-40, -33, -7, -90], np.int64)
    maxs = np.array([55, 20, 31, 64], np.int64)
    for _ in range(500):
        k = np.array([rng.randint(mins[i], maxs[i] + 1) for i in range(4)], np.int64)
        packed = LO.lib().hpl_key2int(k, maxs, mins)
        back = np.zeros(4, np.int64)
        LO.lib().hpl_int2key(packed, maxs, mins, back)
        assert np.array_equal(back, k)


def test_map_against_reference_khash():
    """oracle/_ref/libkhash_ref.so is the reference's own khash (models/khash_int2int.h)."""
    path = os.path.join(ROOT, 'oracle', '_ref', 'libkhash_ref.so')
    if not os.path.exists(path):
        pytest.skip('oracle/_ref not built (reference tree absent and no prebuilt file)')
    ref = ctypes.CDLL(path)
    ref.khash_ref_init.restype = ctypes.c_void_p
    ref.khash_ref_get.restype = ctypes.c_int64
    ref.khash_ref_get.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64]
    ref.khash_ref_set.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64]
    ref.khash_ref_destroy.argtypes = [ctypes.c_void_p]
    L = LO.lib()
    rng = np.random.RandomState(7)
    keys = np.concatenate([rng.randint(-2 ** 62, 2 ** 62, 20000), rng.randint(-50, 50, 20000)]).astype(np.int64)
    vals = rng.randint(-2 ** 40, 2 ** 40, keys.size).astype(np.int64)
    a, b = ref.khash_ref_init(), L.hpl_i2i_init()
    for k, v in zip(keys.tolist(), vals.tolist()):
        if rng.rand() < 0.7:
            ref.khash_ref_set(a, k, v)
            L.hpl_i2i_set(b, k, v)
        probe = int(keys[rng.randint(keys.size)])
        assert ref.khash_ref_get(a, probe, -1) == L.hpl_i2i_get(b, probe, -1)
    assert L.hpl_i2i_set(None, 1, 2) == -1
    ref.khash_ref_destroy(a)
    L.hpl_i2i_destroy(b)
    L.hpl_i2i_destroy(None)                      # NULL-safe like khash_int2int.h:12-15


def test_khash_names_drive_the_reference_build_unsymmetric():
    """The oracle library exports the four functions under the names the reference's CFFI cdef binds
    (/root/reference/models/build_khash_cffi.py:15-22): the same get / set / destroy contract through
    `khash_int2int_*` as through `hpl_i2i_*`."""
    so = ctypes.CDLL(os.path.join(ROOT, 'oracle', 'liblattice_oracle.so'))
    so.khash_int2int_init.restype = ctypes.c_void_p
    so.khash_int2int_destroy.argtypes = [ctypes.c_void_p]
    so.khash_int2int_get.restype = ctypes.c_int64
    so.khash_int2int_get.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64]
    so.khash_int2int_set.restype = ctypes.c_int
    so.khash_int2int_set.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64]
    h = so.khash_int2int_init()
    rng = np.random.RandomState(3)
    truth = {}
    for k, v in zip(rng.randint(-2 ** 50, 2 ** 50, 5000).tolist(), rng.randint(0, 2 ** 31, 5000).tolist()):
        assert so.khash_int2int_set(h, k, v) >= 0
        truth[k] = v
    assert so.khash_int2int_set(h, 7, 1) >= 0 and so.khash_int2int_set(h, 7, 2) >= 0      # overwrite
    truth[7] = 2
    for k, v in truth.items():
        assert so.khash_int2int_get(h, k, -1) == v
    assert so.khash_int2int_get(h, 2 ** 55 + 1, -123) == -123
    assert so.khash_int2int_set(None, 1, 2) == -1                                          # khash_int2int.h:29
    so.khash_int2int_destroy(h)
    so.khash_int2int_destroy(None)                                                         # :12-15


def test_edge_cases():
    # one point, duplicate points, and a cloud that is a single repeated point
    pc = np.array([[0.1, -0.2, 5.0]], np.float32)
    gd = LO.generate_data(pc, pc.copy(), SCALES_FILTER_MAP)
    assert gd[0]['pc1_hash_cnt'] == 4 and gd[0]['pc2_hash_cnt'] == 4
    pc = np.repeat(pc, 7, axis=0)
    gd = LO.generate_data(pc, pc.copy(), SCALES_FILTER_MAP)
    assert gd[0]['pc1_hash_cnt'] == 4
    assert np.array_equal(gd[0]['pc1_lattice_offset'], np.tile(np.arange(4)[:, None], (1, 7)))
    # ragged: the two clouds may have different sizes (NO_CORR sampling, transforms.py:519-523)
    p1, p2, _ = synthetic_pair(64, 5)
    gd = LO.generate_data(p1, p2[:40], SCALES_FILTER_MAP)
    assert gd[0]['pc1_barycentric'].shape[1] == 64 and gd[0]['pc2_barycentric'].shape[1] == 40
