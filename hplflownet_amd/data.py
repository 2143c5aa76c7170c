"""Data path in front of the lattice build.

Own restatement of what feeds the hot path in the reference:

* `ProcessData` (/root/reference/transforms/transforms.py:494-548): scene flow `sf = pc2 - pc1`,
  depth cut (both clouds closer than DEPTH_THRESHOLD, :509-512), sampling of `num_points` indices
  without replacement -- the same indices for both clouds unless NO_CORR (:519-525) -- and the
  `allow_less_points` fallback (:526-532).  Differences: the RNG is an explicit, seedable
  `numpy.random.RandomState` instead of the global one, and failure returns `(None, None, None)`
  without printing.
* `Augmentation` (/root/reference/transforms/transforms.py:551-640), the training-time transform:
  both clouds get one random per-axis scale x rotation about y, one shift and a clipped per-point
  jitter (:565-590); cloud 2 then gets its own rotation about y and shift (:592-606), `sf` is
  taken BEFORE cloud 2's own jitter (:607-613); depth cut and sampling as in ProcessData.  The
  draws are made in the reference's order from the seedable RandomState, so `seed=s` reproduces
  the reference under `np.random.seed(s)` bit for bit (tests/golden/transforms.npz).  Unlike the
  reference the caller's arrays are not modified in place.
* `FlyingThings3DSubset` (/root/reference/datasets/flyingthings3d_subset.py:22-101): leaf
  directories below `<root>/FlyingThings3D_subset_processed_35m/{train,val}` holding `pc1.npy` /
  `pc2.npy`; x and z are negated on load (:96-99); every 4th sample unless `full` (:79-82).
* `KITTI` (/root/reference/datasets/kitti.py:22-107): leaf directories below
  `<root>/KITTI_processed_occ_final`; points with y < -1.4 in BOTH clouds are ground and removed
  (:100-105); frames whose line in the mapping file is empty are skipped (:76-83; the mapping file
  ships with the reference's dataset code, pass its path if you have it).

The reference asserts the canonical sample counts (19 640 / 3 824 / 200) and exits; here a
mismatch is reported by `check_counts()` and left to the caller.  No dataset is available in the
build environment: the readers are exercised on synthetic directory trees of the same layout
(tests/test_data_cpu.py).  Samples are returned as device tensors `(3, N)` ready for
`GenerateDataUnsymmetric.build` -- the lattice itself is built on the GPU by the consumer
(engine.Trainer pipelines it on a second stream), not in DataLoader workers.
"""
import os

import numpy as np
import torch

__all__ = ['ProcessData', 'Augmentation', 'FlyingThings3DSubset', 'KITTI']


class ProcessData(object):
    def __init__(self, data_process_args, num_points, allow_less_points, seed=None):
        self.DEPTH_THRESHOLD = data_process_args['DEPTH_THRESHOLD']
        self.no_corr = data_process_args['NO_CORR']
        self.num_points = num_points
        self.allow_less_points = allow_less_points
        self.rng = np.random.RandomState(seed)

    def __call__(self, data):
        pc1, pc2 = data
        if pc1 is None:
            return None, None, None
        return self._select(pc1, pc2, pc2[:, :3] - pc1[:, :3])

    def _select(self, pc1, pc2, sf):
        """Depth cut + sampling; `sf` rows follow cloud 1's draw."""
        if self.DEPTH_THRESHOLD > 0:
            near = (pc1[:, 2] < self.DEPTH_THRESHOLD) & (pc2[:, 2] < self.DEPTH_THRESHOLD)
        else:
            near = np.ones(pc1.shape[0], dtype=bool)
        idx = np.nonzero(near)[0]
        if idx.size == 0:
            return None, None, None
        i1 = i2 = idx
        if self.num_points > 0:
            if idx.size >= self.num_points:
                i1 = self.rng.choice(idx, size=self.num_points, replace=False)
                i2 = self.rng.choice(idx, size=self.num_points, replace=False) if self.no_corr else i1
            elif not self.allow_less_points:
                return None, None, None
        return pc1[i1], pc2[i2], sf[i1]

    def __repr__(self):
        return ('%s\n(data_process_args: \n\tDEPTH_THRESHOLD: %s\n\tNO_CORR: %s\n\tallow_less_points: %s\n'
                '\tnum_points: %s\n)' % (self.__class__.__name__, self.DEPTH_THRESHOLD, self.no_corr,
                                         self.allow_less_points, self.num_points))


def _rot_y(angle, dtype):
    c, s = np.cos(angle), np.sin(angle)
    return np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]], dtype=dtype)


class Augmentation(object):
    def __init__(self, aug_together_args, aug_pc2_args, data_process_args, num_points, allow_less_points=False,
                 seed=None):
        self.together_args = aug_together_args
        self.pc2_args = aug_pc2_args
        self.sampler = ProcessData(data_process_args, num_points, allow_less_points)
        self.rng = self.sampler.rng = np.random.RandomState(seed)
        self.no_corr = self.sampler.no_corr

    def _jitter(self, a, n):
        return np.clip(a['jitter_sigma'] * self.rng.randn(n, 3), -a['jitter_clip'], a['jitter_clip']).astype(np.float32)

    def __call__(self, data):
        pc1, pc2 = data
        if pc1 is None:
            return None, None, None
        tg, p2, rng = self.together_args, self.pc2_args, self.rng
        n = pc1.shape[0]
        # common motion of the scene: per-axis scale, rotation about y, shift, per-point jitter (in this draw order)
        scale = np.diag(rng.uniform(tg['scale_low'], tg['scale_high'], 3).astype(np.float32))
        m = scale.dot(_rot_y(rng.uniform(-tg['degree_range'], tg['degree_range']), np.float32).T)
        shift = rng.uniform(-tg['shift_range'], tg['shift_range'], (1, 3)).astype(np.float32)
        bias = shift + self._jitter(tg, n)
        a = pc1.copy()
        b = pc2.copy()
        a[:, :3] = pc1[:, :3].dot(m) + bias
        b[:, :3] = pc2[:, :3].dot(m) + bias
        # extra motion of cloud 2: rotation about y, shift; the flow is read off before its jitter
        m2 = _rot_y(rng.uniform(-p2['degree_range'], p2['degree_range']), pc1.dtype)
        shift2 = rng.uniform(-p2['shift_range'], p2['shift_range'], (1, 3)).astype(np.float32)
        b[:, :3] = b[:, :3].dot(m2.T) + shift2
        sf = b[:, :3] - a[:, :3]
        if not self.no_corr:
            b[:, :3] += self._jitter(p2, n)
        return self.sampler._select(a, b, sf)

    def __repr__(self):
        fmt = lambda d: ''.join('\t%-10s %s\n' % (k, d[k]) for k in sorted(d))      # noqa: E731
        return ('%s\n(together_args: \n%s\npc2_args: \n%s\ndata_process_args: \n\tDEPTH_THRESHOLD: %s\n'
                '\tNO_CORR: %s\n\tallow_less_points: %s\n\tnum_points: %s\n)' % (
                    self.__class__.__name__, fmt(self.together_args), fmt(self.pc2_args), self.sampler.DEPTH_THRESHOLD,
                    self.no_corr, self.sampler.allow_less_points, self.sampler.num_points))


def _leaf_dirs(root):
    """Sorted directories below `root` that contain no sub-directory (the reference's `useful_paths`)."""
    root = os.path.realpath(os.path.expanduser(root))
    return sorted(d for d, sub, _ in os.walk(root) if len(sub) == 0)


class _PairFolder(object):
    """Common part: list of sample directories, transform, tensors on `device`."""

    canonical = None        # expected number of leaf directories, for check_counts()

    def __init__(self, transform, device='cuda'):
        self.transform = transform
        self.device = device
        self.samples = []

    def __len__(self):
        return len(self.samples)

    def check_counts(self):
        """None if the tree has the canonical number of samples, else a message."""
        if self.canonical is not None and self._found != self.canonical:
            return '%s: found %d sample directories, the published split has %d' % (
                self.__class__.__name__, self._found, self.canonical)
        return None

    def load(self, path):
        raise NotImplementedError

    def __getitem__(self, index):
        """-> (pc1, pc2, sf) float32 device tensors (3, N); falls on to the next sample if the
        transform rejects this one (the reference draws a random replacement, :44-47)."""
        for k in range(len(self.samples)):
            path = self.samples[(index + k) % len(self.samples)]
            out = self.transform(self.load(path)) if self.transform is not None else None
            if out is None:
                pc1, pc2 = self.load(path)
                out = (pc1, pc2, pc2 - pc1)
            if out[0] is not None:
                return tuple(torch.from_numpy(np.ascontiguousarray(a[:, :3].T, dtype=np.float32)).to(self.device)
                             for a in out)
        raise RuntimeError('no usable sample under %s' % self.root)


class FlyingThings3DSubset(_PairFolder):
    def __init__(self, train, transform, data_root, full=False, device='cuda'):
        super(FlyingThings3DSubset, self).__init__(transform, device)
        self.train = train
        self.root = os.path.join(data_root, 'FlyingThings3D_subset_processed_35m', 'train' if train else 'val')
        self.canonical = 19640 if train else 3824
        dirs = _leaf_dirs(self.root)
        self._found = len(dirs)
        self.samples = dirs if full else dirs[::4]
        if not self.samples:
            raise RuntimeError('Found 0 files in subfolders of: ' + self.root)

    def load(self, path):
        pc1 = np.load(os.path.join(path, 'pc1.npy'))
        pc2 = np.load(os.path.join(path, 'pc2.npy'))
        for pc in (pc1, pc2):           # the subset stores x and z with the opposite sign
            pc[..., 0] *= -1
            pc[..., -1] *= -1
        return pc1, pc2


class KITTI(_PairFolder):
    canonical = 200

    def __init__(self, transform, data_root, remove_ground=True, mapping_file=None, device='cuda'):
        super(KITTI, self).__init__(transform, device)
        self.root = os.path.join(data_root, 'KITTI_processed_occ_final')
        self.remove_ground = remove_ground
        dirs = _leaf_dirs(self.root)
        self._found = len(dirs)
        if mapping_file is not None:
            with open(mapping_file) as fd:
                lines = [ln.strip() for ln in fd.readlines()]
            dirs = [d for d in dirs if lines[int(os.path.split(d)[-1])] != '']
        self.samples = dirs
        if not self.samples:
            raise RuntimeError('Found 0 files in subfolders of: ' + self.root)

    def load(self, path):
        pc1 = np.load(os.path.join(path, 'pc1.npy'))
        pc2 = np.load(os.path.join(path, 'pc2.npy'))
        if self.remove_ground:
            keep = ~((pc1[:, 1] < -1.4) & (pc2[:, 1] < -1.4))
            pc1, pc2 = pc1[keep], pc2[keep]
        return pc1, pc2
