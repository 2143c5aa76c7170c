"""Synthetic FlyingThings3D-like point-cloud pairs and closed-form parameter fills.

No dataset or trained checkpoint exists in the build or GPU containers, so every
test, fixture and benchmark draws its inputs from the generators below.  The
frustum uses the FT3D intrinsics of the reference (`utils/geometry.py:61`,
f=1050, cx=479.5, cy=269.5) and the 35 m depth cut of
`configs/train_ours.yaml:41` (DEPTH_THRESHOLD), as specified in SURVEY.md §8(d1).
"""
import math

import numpy as np

#: `configs/train_ours.yaml:5-34` — [scale, bcn radius, corr filter radius, corr corr radius]
SCALES_FILTER_MAP = [[3., 1, -1, -1],
                     [2., 1, -1, -1],
                     [1., 1, 1, 1],
                     [0.5, 1, 1, 1],
                     [0.25, 1, 1, 1],
                     [0.125, 1, 1, 1],
                     [0.0625, 1, 1, 1]]


def synthetic_pair(num_points, seed=0):
    """Return (pc1, pc2, sf) float32 arrays of shape (N, 3).

    Draw order is part of the contract (fixtures depend on it): u, v, z (N draws
    each), then the (N, 3) Gaussian displacement.
    """
    rng = np.random.RandomState(seed)
    u = rng.uniform(0., 960., num_points)
    v = rng.uniform(0., 540., num_points)
    z = rng.uniform(1.5, 35., num_points)
    pc1 = np.stack([(u - 479.5) * z / 1050., (v - 269.5) * z / 1050., z], axis=1).astype(np.float32)
    pc2 = (pc1 + rng.normal(0., 0.3, (num_points, 3))).astype(np.float32)
    sf = pc2 - pc1
    return pc1, pc2, sf


def surface_pair(num_points, seed=0):
    """A surface-like pair: points on 8 smooth patches (what depth-map data such as FlyingThings3D looks like
    to the lattice: ~0.5 vertices per point at level 0 instead of the 3.2 of the uniform frustum above), cloud 2 =
    cloud 1 moved rigidly + 2 cm noise.  Used by `bench.py --data surface` and tests/stress/surface_check.py to
    show the sensitivity of the numbers to the data; not part of any fixture."""
    rng = np.random.RandomState(seed)
    pts = []
    per = (num_points + 7) // 8
    for _ in range(8):
        c = rng.uniform([-8, -3, 5], [8, 3, 30])
        u, v = rng.uniform(-3, 3, per), rng.uniform(-2, 2, per)
        a, b = rng.uniform(-0.5, 0.5, 2)
        z = c[2] + a * u + b * v + 0.3 * np.sin(u) + rng.normal(0, 0.01, per)
        pts.append(np.stack([c[0] + u, c[1] + v, z], 1))
    pc1 = np.concatenate(pts)[:num_points].astype(np.float32)
    pc2 = (pc1 + np.array([0.3, 0.0, 0.2], np.float32) + rng.normal(0, 0.02, pc1.shape)).astype(np.float32)
    return pc1, pc2, pc2 - pc1


def closed_form_fill(name, shape):
    """Deterministic parameter values so fixtures need not ship weight blobs.

    value[i] = amp * sin(0.37 * i + len(name)); amp = 1/sqrt(fan_in) for weights
    (ndim >= 2, fan_in = prod(shape[1:])) and 0.05 for vectors (biases).
    """
    n = int(np.prod(shape))
    i = np.arange(n, dtype=np.float64)
    if len(shape) >= 2:
        amp = 1.0 / math.sqrt(float(np.prod(shape[1:])))
    else:
        amp = 0.05
    return (amp * np.sin(0.37 * i + len(name))).astype(np.float32).reshape(shape)


def hash_fill(name, shape):
    """Exactly reproducible pseudo-random parameter values (integer hash, no libm):
    u = splitmix64(i + crc32(name) * 2^32) mapped to U(-a, a), a = sqrt(6 / fan_in) for weights
    (He-uniform: keeps activations O(1) through the ~25 layers of the models) and 0.05 for
    vectors.  Used by the whole-model fixtures: the sin fill above gives rank-2 weight
    matrices, whose massive cancellations make whole-model gradients hypersensitive to
    single LeakyReLU sign flips (measured; see DESIGN_HISTORY.md)."""
    import zlib
    n = int(np.prod(shape))
    x = np.arange(n, dtype=np.uint64) + (np.uint64(zlib.crc32(name.encode())) << np.uint64(32))
    with np.errstate(over='ignore'):
        x = x + np.uint64(0x9E3779B97F4A7C15)
        x ^= x >> np.uint64(30)
        x *= np.uint64(0xBF58476D1CE4E5B9)
        x ^= x >> np.uint64(27)
        x *= np.uint64(0x94D049BB133111EB)
        x ^= x >> np.uint64(31)
    u = (x >> np.uint64(11)).astype(np.float64) / float(1 << 53)          # [0, 1)
    a = math.sqrt(6.0 / float(np.prod(shape[1:]))) if len(shape) >= 2 else 0.05
    return ((2.0 * u - 1.0) * a).astype(np.float32).reshape(shape)


#: weight gain of the (legacy) sin fill for whole models; the model fixtures use hash_fill
MODEL_GAIN = 5.0

#: gradient tensors larger than this are stored strided in the fixtures
GRAD_SUBSAMPLE_MIN, GRAD_SUBSAMPLE_STRIDE = 16384, 5


def subsample(a):
    """Fixture storage rule for big gradient tensors: flat stride-5 subsample."""
    a = a.reshape(-1)
    return a[::GRAD_SUBSAMPLE_STRIDE] if a.size > GRAD_SUBSAMPLE_MIN else a


def fill_module_(module, gain=1.0, kind='sin'):
    """In-place deterministic fill of every float parameter of a torch module
    (weights, i.e. ndim >= 2, additionally scaled by `gain`); kind 'sin' or 'hash'."""
    import torch
    fn = closed_form_fill if kind == 'sin' else hash_fill
    with torch.no_grad():
        for name, p in module.named_parameters():
            v = fn(name, tuple(p.shape))
            if p.dim() >= 2:
                v = v * np.float32(gain)
            p.copy_(torch.from_numpy(v))
    return module
