#!/usr/bin/env python
"""Golden vectors for the dataset readers: run the REFERENCE's own
`datasets.FlyingThings3DSubset` and `datasets.KITTI` (datasets/flyingthings3d_subset.py:62-101,
datasets/kitti.py:66-107) on a synthetic directory tree laid out like the published datasets and store what
they return: the sample lists (relative paths), `pc_loader` outputs and one `__getitem__` through the reference's
`ProcessData` under `np.random.seed(s)`.  Build container only (needs /root/reference through tools/ref_import.py).
The file written, tests/golden/datasets.npz, holds outputs only: the tree is regenerated from seeds by `make_tree`
below, which tests/test_data_cpu.py imports.  The KITTI split comes from the reference's own data file
datasets/KITTI_mapping.txt, read in place by its reader; the fixture keeps WHICH of the 200 frames it maps (a boolean
mask), from which the test writes a stand-in mapping file for the reader under test.

The reference's FlyingThings3D reader calls sys.exit(1) when the tree does not hold the published 19 640 / 3 824
samples (flyingthings3d_subset.py:71-77); the generator runs it on a small tree with that module's `sys.exit`
replaced by a no-op -- the sample selection that follows (every 4th directory unless `full`, :79-82) is the
reference's code, unmodified."""
import os
import sys
import tempfile
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
GOLD = os.path.join(ROOT, 'tests', 'golden')

FT3D = 'FlyingThings3D_subset_processed_35m'
KITTI_DIR = 'KITTI_processed_occ_final'
N_TRAIN, N_VAL, N_KITTI = 37, 9, 200
PD_ARGS = dict(dp=dict(DEPTH_THRESHOLD=35., NO_CORR=True), n=48, less=False)        # configs/test_ours_*.yaml shape
SEED = 11


def _pair(rng, n, ground=False):
    pc1 = rng.uniform(-6, 6, (n, 3)).astype(np.float32)
    pc1[:, 2] = rng.uniform(2, 40, n)
    if ground:
        pc1[:, 1] = rng.uniform(-1.8, 1.0, n)            # some points under the -1.4 m ground cut (kitti.py:100)
    pc2 = (pc1 + rng.normal(0, 0.25, (n, 3))).astype(np.float32)
    return pc1, pc2


def make_tree(root):
    """The synthetic datasets, deterministic.  FT3D stores x and z with the opposite sign (the reader flips them back)."""
    rng = np.random.RandomState(4242)
    for split, n in (('train', N_TRAIN), ('val', N_VAL)):
        for i in range(n):
            d = os.path.join(root, FT3D, split, '%07d' % i)
            os.makedirs(d)
            pc1, pc2 = _pair(rng, 64 + (i % 5) * 7)
            flip = np.array([-1, 1, -1], np.float32)
            np.save(os.path.join(d, 'pc1.npy'), pc1 * flip)
            np.save(os.path.join(d, 'pc2.npy'), pc2 * flip)
    for i in range(N_KITTI):
        d = os.path.join(root, KITTI_DIR, '%06d' % i)
        os.makedirs(d)
        pc1, pc2 = _pair(rng, 60 + (i % 3) * 11, ground=True)
        np.save(os.path.join(d, 'pc1.npy'), pc1)
        np.save(os.path.join(d, 'pc2.npy'), pc2)


def main():
    sys.path.insert(0, HERE)
    from ref_import import import_reference, REF
    T = import_reference().T
    import datasets.flyingthings3d_subset as ft_mod
    import datasets.kitti as kitti_mod
    ft_mod.sys = types.SimpleNamespace(exit=lambda code=0: None)        # see the module docstring
    out = {}
    with tempfile.TemporaryDirectory() as root:
        make_tree(root)
        ident = lambda x: (x[0], x[1], x[2], None)                      # gen_func: the lattice build is tested elsewhere
        for split, train in (('train', True), ('val', False)):
            for full in (False, True):
                args = types.SimpleNamespace(data_root=root, num_points=PD_ARGS['n'], full=full)
                tr = T.ProcessData(PD_ARGS['dp'], PD_ARGS['n'], PD_ARGS['less'])
                ds = ft_mod.FlyingThings3DSubset(train, tr, ident, args)
                rel = [os.path.relpath(p, os.path.realpath(root)) for p in ds.samples]
                out['ft3d_%s_%s_samples' % (split, 'full' if full else 'quarter')] = np.array(rel)
                if not full:
                    for k in (0, len(ds) - 1):
                        a, b = ds.pc_loader(ds.samples[k])
                        out['ft3d_%s_load%d_pc1' % (split, k)], out['ft3d_%s_load%d_pc2' % (split, k)] = a, b
                    np.random.seed(SEED)
                    p1, p2, sf, _, path = ds[1]
                    out['ft3d_%s_item1_pc1' % split], out['ft3d_%s_item1_pc2' % split] = p1, p2
                    out['ft3d_%s_item1_sf' % split] = sf
                    out['ft3d_%s_item1_path' % split] = np.array(os.path.relpath(path, os.path.realpath(root)))
        for rg in (True, False):
            args = types.SimpleNamespace(data_root=root, num_points=PD_ARGS['n'], remove_ground=rg)
            tr = T.ProcessData(PD_ARGS['dp'], PD_ARGS['n'], PD_ARGS['less'])
            ds = kitti_mod.KITTI(False, tr, ident, args)
            tag = 'kitti_%s' % ('noground' if rg else 'all')
            rel = [os.path.relpath(p, os.path.realpath(root)) for p in ds.samples]
            out[tag + '_samples'] = np.array(rel)
            for k in (0, 77):
                a, b = ds.pc_loader(ds.samples[k])
                out['%s_load%d_pc1' % (tag, k)], out['%s_load%d_pc2' % (tag, k)] = a, b
            np.random.seed(SEED)
            p1, p2, sf, _, path = ds[5]
            out[tag + '_item5_pc1'], out[tag + '_item5_pc2'], out[tag + '_item5_sf'] = p1, p2, sf
        lines = [ln.strip() for ln in open(os.path.join(REF, 'datasets', 'KITTI_mapping.txt'))]
        out['kitti_mapped'] = np.array([ln != '' for ln in lines])
    np.savez_compressed(os.path.join(GOLD, 'datasets.npz'), **out)
    print({k: getattr(v, 'shape', None) for k, v in out.items() if 'samples' in k or 'mapped' in k})


if __name__ == '__main__':
    main()
