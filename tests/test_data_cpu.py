"""CPU: the data path in front of the lattice build on synthetic directory
trees laid out like the published datasets; expectations restate the reference's rules
(transforms.py:494-548, flyingthings3d_subset.py:62-101, kitti.py:62-107)."""
import os

import numpy as np

from hplflownet_amd.data import KITTI, FlyingThings3DSubset, ProcessData


def _write(root, rel, pc1, pc2):
    d = os.path.join(root, rel)
    os.makedirs(d)
    np.save(os.path.join(d, 'pc1.npy'), pc1)
    np.save(os.path.join(d, 'pc2.npy'), pc2)
    return d


def test_process_data_rules():
    rng = np.random.RandomState(0)
    pc1 = rng.uniform(-5, 5, (300, 3)).astype(np.float32)
    pc1[:, 2] = rng.uniform(1, 60, 300)
    pc2 = pc1 + rng.normal(0, 0.2, (300, 3)).astype(np.float32)
    near = (pc1[:, 2] < 35.0) & (pc2[:, 2] < 35.0)
    t = ProcessData({'DEPTH_THRESHOLD': 35.0, 'NO_CORR': False}, 64, False, seed=1)
    a, b, sf = t([pc1, pc2])
    assert a.shape == b.shape == sf.shape == (64, 3)
    assert (a[:, 2] < 35).all() and (b[:, 2] < 35).all()
    assert np.array_equal(sf, b - a)                                  # same indices for both clouds (:525)
    rows = [np.nonzero((pc1 == r).all(1))[0][0] for r in a]
    assert len(set(rows)) == 64 and near[rows].all()                  # without replacement, inside the mask
    a2, _, _ = ProcessData({'DEPTH_THRESHOLD': 35.0, 'NO_CORR': False}, 64, False, seed=1)([pc1, pc2])
    assert np.array_equal(a, a2)                                      # seedable
    # NO_CORR: independent draws for cloud 2; sf still belongs to cloud 1's draw (:519-523,541-543)
    a, b, sf = ProcessData({'DEPTH_THRESHOLD': 35.0, 'NO_CORR': True}, 64, False, seed=2)([pc1, pc2])
    rows = [np.nonzero((pc1 == r).all(1))[0][0] for r in a]
    assert np.array_equal(sf, (pc2 - pc1)[rows]) and not np.array_equal(sf, b - a)
    # too few points: rejected, or everything inside the mask with allow_less_points (:526-532)
    assert ProcessData({'DEPTH_THRESHOLD': 35.0, 'NO_CORR': False}, 1000, False)([pc1, pc2]) == (None, None, None)
    a, b, sf = ProcessData({'DEPTH_THRESHOLD': 35.0, 'NO_CORR': False}, 1000, True)([pc1, pc2])
    assert len(a) == int(near.sum()) and np.array_equal(sf, b - a)
    # no threshold, no sampling: identity
    a, b, sf = ProcessData({'DEPTH_THRESHOLD': -1, 'NO_CORR': False}, -1, False)([pc1, pc2])
    assert np.array_equal(a, pc1) and np.array_equal(b, pc2)
    assert ProcessData({'DEPTH_THRESHOLD': 0.5, 'NO_CORR': False}, 8, True)([pc1, pc2]) == (None, None, None)
    assert ProcessData({'DEPTH_THRESHOLD': 35.0, 'NO_CORR': False}, 8, True)([None, None]) == (None, None, None)


def test_flyingthings_subset_reader(tmp_path):
    rng = np.random.RandomState(1)
    root = str(tmp_path)
    base = os.path.join(root, 'FlyingThings3D_subset_processed_35m')
    clouds = {}
    for i in range(9):
        pc1 = rng.uniform(1, 20, (50, 3)).astype(np.float32)
        pc2 = (pc1 + 0.1).astype(np.float32)
        clouds[_write(base, 'val/%07d' % i, pc1, pc2)] = (pc1, pc2)
    ds = FlyingThings3DSubset(False, None, root, device='cpu')
    assert len(ds) == 3 and ds.samples == sorted(clouds)[::4]         # every 4th unless full (:79-82)
    assert len(FlyingThings3DSubset(False, None, root, full=True, device='cpu')) == 9
    assert 'found 9' in ds.check_counts()                             # canonical 3824: reported, not fatal
    p1, p2, sf = ds[1]
    want1, want2 = clouds[ds.samples[1]]
    flip = np.array([-1, 1, -1], np.float32)
    assert np.array_equal(p1.numpy().T, want1 * flip) and np.array_equal(p2.numpy().T, want2 * flip)   # (:96-99)
    assert np.allclose(sf.numpy(), (p2 - p1).numpy())
    t = ProcessData({'DEPTH_THRESHOLD': 35.0, 'NO_CORR': False}, 16, False, seed=0)
    p1, p2, sf = FlyingThings3DSubset(False, t, root, device='cpu')[0]
    assert tuple(p1.shape) == (3, 16)                                  # z was negated -> all "near"


def test_kitti_reader(tmp_path):
    rng = np.random.RandomState(2)
    root = str(tmp_path)
    base = os.path.join(root, 'KITTI_processed_occ_final')
    for i in range(4):
        pc1 = rng.uniform(-3, 3, (80, 3)).astype(np.float32)
        pc2 = (pc1 + rng.normal(0, 0.3, (80, 3))).astype(np.float32)
        _write(base, '%06d' % i, pc1, pc2)
    mapping = os.path.join(root, 'map.txt')
    with open(mapping, 'w') as f:
        f.write('a\n\nb\nc\n')                                       # frame 1 unmapped -> skipped (:76-83)
    ds = KITTI(None, root, remove_ground=True, mapping_file=mapping, device='cpu')
    assert [os.path.basename(s) for s in ds.samples] == ['000000', '000002', '000003']
    raw1 = np.load(os.path.join(ds.samples[0], 'pc1.npy'))
    raw2 = np.load(os.path.join(ds.samples[0], 'pc2.npy'))
    keep = ~((raw1[:, 1] < -1.4) & (raw2[:, 1] < -1.4))               # ground only if below in BOTH (:100-105)
    p1, p2, _ = ds[0]
    assert np.array_equal(p1.numpy().T, raw1[keep]) and np.array_equal(p2.numpy().T, raw2[keep])
    assert 0 < keep.sum() < 80
    assert len(KITTI(None, root, remove_ground=False, device='cpu')) == 4


def test_transforms_match_reference_vectors():
    """ProcessData / Augmentation with seed=s reproduce the reference run under np.random.seed(s), bit for bit
    (fixture: tools/make_transform_fixture.py -> tests/golden/transforms.npz), two consecutive calls each."""
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.join(os.path.dirname(here), 'tools'))
    import make_transform_fixture as F
    from hplflownet_amd import data
    gold = np.load(os.path.join(here, 'golden', 'transforms.npz'))
    for tag, kind, kw, seed in F.CASES:
        pc1, pc2 = F.cloud_pair(seed)
        keep1, keep2 = pc1.copy(), pc2.copy()
        t = F.make(data, kind, kw, seed=seed)
        for suffix in ('', '_b'):
            a, b, sf = t([pc1, pc2])
            for name, got in (('pc1', a), ('pc2', b), ('sf', sf)):
                want = gold['%s_%s%s' % (tag, name, suffix)]
                assert got.dtype == want.dtype and np.array_equal(got, want), (tag, name, suffix)
        assert np.array_equal(pc1, keep1) and np.array_equal(pc2, keep2)      # inputs are left alone
    assert 'together_args' in repr(F.make(data, 'Augmentation', F.CASES[3][3 - 1]))


def test_readers_match_reference_readers(tmp_path):
    """Row f3 pinned to the reference: tests/golden/datasets.npz holds what the reference's own FlyingThings3DSubset /
    KITTI classes returned on the synthetic tree of tools/make_dataset_fixture.make_tree (sample lists, pc_loader
    outputs, one __getitem__ through the reference's ProcessData under np.random.seed); the readers here give the same
    on the same tree, bit for bit."""
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.join(os.path.dirname(here), 'tools'))
    import make_dataset_fixture as F
    gold = np.load(os.path.join(here, 'golden', 'datasets.npz'))
    root = str(tmp_path)
    F.make_tree(root)
    real = os.path.realpath(root)
    rel = lambda ds: [os.path.relpath(p, real) for p in ds.samples]
    pd = lambda seed=None: ProcessData(F.PD_ARGS['dp'], F.PD_ARGS['n'], F.PD_ARGS['less'], seed=seed)
    for split, train in (('train', True), ('val', False)):
        for full in (False, True):
            ds = FlyingThings3DSubset(train, None, root, full=full, device='cpu')
            assert rel(ds) == list(gold['ft3d_%s_%s_samples' % (split, 'full' if full else 'quarter')])
        ds = FlyingThings3DSubset(train, None, root, device='cpu')
        for k in (0, len(ds) - 1):
            a, b = ds.load(ds.samples[k])
            assert np.array_equal(a, gold['ft3d_%s_load%d_pc1' % (split, k)]) and np.array_equal(b, gold['ft3d_%s_load%d_pc2' % (split, k)])
        ds = FlyingThings3DSubset(train, pd(F.SEED), root, device='cpu')
        assert os.path.relpath(ds.samples[1], real) == str(gold['ft3d_%s_item1_path' % split])
        p1, p2, sf = ds[1]
        for got, name in ((p1, 'pc1'), (p2, 'pc2'), (sf, 'sf')):
            assert np.array_equal(got.numpy().T, gold['ft3d_%s_item1_%s' % (split, name)]), (split, name)
    # KITTI: the 142-frame split of the reference's mapping file (here: a stand-in with the same empty lines)
    mapping = os.path.join(root, 'mapping.txt')
    with open(mapping, 'w') as f:
        f.write(''.join('x\n' if m else '\n' for m in gold['kitti_mapped']))
    assert int(gold['kitti_mapped'].sum()) == 142
    for rg, tag in ((True, 'kitti_noground'), (False, 'kitti_all')):
        ds = KITTI(None, root, remove_ground=rg, mapping_file=mapping, device='cpu')
        assert rel(ds) == list(gold[tag + '_samples']) and len(ds) == 142
        for k in (0, 77):
            a, b = ds.load(ds.samples[k])
            assert np.array_equal(a, gold['%s_load%d_pc1' % (tag, k)]) and np.array_equal(b, gold['%s_load%d_pc2' % (tag, k)])
        p1, p2, sf = KITTI(pd(F.SEED), root, remove_ground=rg, mapping_file=mapping, device='cpu')[5]
        for got, name in ((p1, 'pc1'), (p2, 'pc2'), (sf, 'sf')):
            assert np.array_equal(got.numpy().T, gold['%s_item5_%s' % (tag, name)]), (tag, name)
