"""Shared helpers for the test-suite (CPU and GPU)."""
import hashlib
import json
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, 'tests', 'golden')

from hplflownet_amd.synthetic import SCALES_FILTER_MAP, synthetic_pair  # noqa: E402

INT_KEYS = ('pc1_lattice_offset', 'pc2_lattice_offset', 'pc1_blur_neighbors', 'pc2_blur_neighbors',
            'pc1_corr_indices', 'pc2_corr_indices')


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def load_golden_lattice(n):
    """-> (list of per-level dicts holding the levels stored in full, digest dict)."""
    z = np.load(os.path.join(GOLD, 'lattice_n%d.npz' % n))
    digest = json.loads(bytes(z['sha256_json']).decode())
    levels = {}
    for k in z.files:
        if k == 'sha256_json':
            continue
        l, name = k.split('_', 1)
        v = z[k]
        if v.dtype == np.int32:
            v = v.astype(np.int64)
        levels.setdefault(int(l[1:]), {})[name] = v
    out = []
    for l in sorted(levels):
        d = levels[l]
        d['pc1_hash_cnt'] = int(d['pc1_hash_cnt'])
        d['pc2_hash_cnt'] = int(d['pc2_hash_cnt'])
        out.append(d)
    return out, digest


_ORACLE_CACHE = {}


def oracle_lattice(n, seed=0, nscales=7):
    """generated_data from the C oracle (bit-identical to the reference: test_oracle_lattice)."""
    from oracle import lattice_oracle as LO
    key = (n, seed, nscales)
    if key not in _ORACLE_CACHE:
        pc1, pc2, sf = synthetic_pair(n, seed)
        _ORACLE_CACHE[key] = (pc1, pc2, sf, LO.generate_data(pc1, pc2, SCALES_FILTER_MAP[:nscales]))
    return _ORACLE_CACHE[key]


def rel_err(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))
