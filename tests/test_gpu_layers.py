"""GPU: the drop-in layers, the models and the device lattice against the reference-generated
golden vectors (tests/golden) and the CPU oracle.  These tests read like the reference would
test itself: same constructors, same forward calls, (1, C, N) tensors, int64 index tensors."""
import os
import types

import numpy as np
import pytest
import torch

from common import GOLD, oracle_lattice, rel_err
from hplflownet_amd.synthetic import (SCALES_FILTER_MAP, closed_form_fill, fill_module_, subsample,
                                      synthetic_pair)

pytestmark = pytest.mark.gpu
DEV = 'cuda'
TOL = 2e-5        # forward, relative to max-abs (reference itself is fp32 with another summation order)
GTOL = 5e-4       # gradients that reduce over all vertices (see tests/test_oracle_layers.py RTOL)


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def model_args(n, evaluate=True):
    return types.SimpleNamespace(dim=3, scales_filter_map=SCALES_FILTER_MAP[:n], evaluate=evaluate, use_leaky=True,
                                 bcn_use_bias=True, bcn_use_norm=True, last_relu=False, DEVICE='cuda')


def gd_batched_device(gd):
    """generated_data as a DataLoader would deliver it: B=1 tensors, counts as 1-element tensors."""
    out = []
    for d in gd:
        out.append({k: (T(v)[None] if isinstance(v, np.ndarray) else torch.tensor([v])) for k, v in d.items()})
    return out


BCL_CASES = [  # tag, n, level, cin, couts, do_splat, do_slice, last_relu, use_norm
    ('cfg1', 1024, 0, 68, [64, 64], True, True, False, True),
    ('down0', 256, 0, 68, [64, 64], True, False, False, True),
    ('down2', 256, 2, 68, [64, 64], True, False, False, True),
    ('up2', 256, 2, 36, [32, 32], False, True, False, True),
    ('down1_single', 256, 1, 68, [64], True, False, False, True),
    ('up1_single_relu', 256, 1, 20, [32], False, True, True, True),
    ('cfg_nonorm', 256, 0, 12, [16, 16], True, True, False, False),
]


@pytest.mark.parametrize('case', BCL_CASES, ids=[c[0] for c in BCL_CASES])
def test_bilateral_conv_golden_F4(case):
    import hplflownet_amd as H
    tag, n, lvl, cin, couts, do_splat, do_slice, last_relu, use_norm = case
    z = np.load(os.path.join(GOLD, 'layers.npz'))
    _, _, _, gd = oracle_lattice(n)
    g = gd[lvl]
    m = H.BilateralConvFlex(3, 1, cin, couts, 'cuda', True, True, use_norm, do_splat, do_slice, last_relu,
                            chunk_size=-1)
    fill_module_(m)
    if do_slice:
        with torch.no_grad():
            m.bias.copy_(torch.from_numpy(closed_form_fill('slice_bias', (couts[-1],))))
    m = m.to(DEV)
    Hc = g['pc1_hash_cnt']
    nfeat = g['pc1_barycentric'].shape[1] if do_splat else Hc
    x = T(closed_form_fill(tag + '_x', (1, cin, nfeat)) * np.float32(np.sqrt(cin))).requires_grad_(True)
    y = m(x,
          T(g['pc1_barycentric'])[None] if do_splat else None,
          T(g['pc1_lattice_offset'])[None] if do_splat else None,
          T(g['pc1_blur_neighbors'])[None],
          T(g['pc1_barycentric'])[None] if do_slice else None,
          T(g['pc1_lattice_offset'])[None] if do_slice else None)
    assert tuple(y.shape) == (1, couts[-1], g['pc1_barycentric'].shape[1] if do_slice else Hc)
    assert rel_err(subsample(y.detach().cpu().numpy()[0]), z[tag + '_y']) < TOL
    go = T(closed_form_fill(tag + '_g', tuple(y.shape)) * np.float32(np.sqrt(y.shape[1])))
    (y * go).sum().backward()
    assert rel_err(subsample(x.grad.cpu().numpy()[0]), z[tag + '_gx']) < 5 * TOL
    for name, p in m.named_parameters():
        assert rel_err(subsample(p.grad.cpu().numpy()), z[tag + '_g_' + name].reshape(-1)) < GTOL, name


@pytest.mark.parametrize('tag,lvl,prev_dim,corr_outs,outs', [
    ('corr_noprev', 2, 0, [32, 32], [64, 64]),
    ('corr_prev', 3, 64, [32, 32], [64, 64]),
    ('corr_shallow', 4, 64, [32], [32])])
def test_bilateral_corr_golden_F4(tag, lvl, prev_dim, corr_outs, outs):
    import hplflownet_amd as H
    z = np.load(os.path.join(GOLD, 'layers.npz'))
    _, _, _, gd = oracle_lattice(256)
    g = gd[lvl]
    H1, H2 = g['pc1_hash_cnt'], g['pc2_hash_cnt']
    m = H.BilateralCorrelationFlex(3, 1, 1, 64, corr_outs, outs, 'cuda', True, True, True, prev_dim, False,
                                   chunk_size=-1)
    fill_module_(m)
    m = m.to(DEV)
    f1 = T(closed_form_fill(tag + '_f1', (1, 64, H1)) * 8).requires_grad_(True)
    f2 = T(closed_form_fill(tag + '_f2', (1, 64, H2)) * 8).requires_grad_(True)
    prev = None
    if prev_dim:
        prev = T(closed_form_fill(tag + '_prev', (1, prev_dim, g['pc1_barycentric'].shape[1])) * 8)
        prev.requires_grad_(True)
    y = m(f1, f2, prev,
          T(g['pc1_barycentric'])[None] if prev_dim else None,
          T(g['pc1_lattice_offset'])[None] if prev_dim else None,
          T(g['pc1_corr_indices'])[None], T(g['pc2_corr_indices'])[None], H1, H2)
    assert tuple(y.shape) == (1, outs[-1], H1)
    assert rel_err(subsample(y.detach().cpu().numpy()[0]), z[tag + '_y']) < TOL
    go = T(closed_form_fill(tag + '_g', tuple(y.shape)) * 8)
    (y * go).sum().backward()
    assert rel_err(subsample(f1.grad.cpu().numpy()[0]), z[tag + '_gf1']) < GTOL
    assert rel_err(subsample(f2.grad.cpu().numpy()[0]), z[tag + '_gf2']) < GTOL
    if prev is not None:
        assert rel_err(subsample(prev.grad.cpu().numpy()[0]), z[tag + '_gprev']) < GTOL
    for name, p in m.named_parameters():
        assert rel_err(subsample(p.grad.cpu().numpy()), z[tag + '_g_' + name].reshape(-1)) < GTOL, name


def test_sparse_sum_golden_F4():
    import hplflownet_amd as H
    z = np.load(os.path.join(GOLD, 'layers.npz'))
    _, _, _, gd = oracle_lattice(256)
    idx = T((gd[0]['pc1_lattice_offset'] + 1).reshape(1, -1))
    vals = T(closed_form_fill('ss_vals', (idx.shape[1], 5))).requires_grad_(True)
    ss = H.sparse_sum(idx, vals, torch.Size([gd[0]['pc1_hash_cnt'] + 1, 5]), True)
    (ss * ss).sum().backward()
    assert rel_err(ss.detach().cpu().numpy(), z['ss_y']) < TOL
    assert rel_err(vals.grad.cpu().numpy(), z['ss_gvals']) < TOL


@pytest.mark.parametrize('tag,cls,n,nsc', [('shallow_n256', 'HPLFlowNetShallow', 256, 5),
                                           ('shallow_n1024', 'HPLFlowNetShallow', 1024, 5),
                                           ('full_n256', 'HPLFlowNet', 256, 7)])
def test_whole_model_golden_F5(tag, cls, n, nsc):
    import hplflownet_amd as H
    z = np.load(os.path.join(GOLD, 'models.npz'))
    pc1, pc2, sf, gd = oracle_lattice(n)
    m = getattr(H, cls)(model_args(nsc))
    fill_module_(m, 1.0, 'hash')
    m = m.to(DEV)
    p1, p2 = T(pc1.T)[None], T(pc2.T)[None]
    y = m(p1, p2, gd_batched_device(gd[:nsc]))
    assert tuple(y.shape) == (1, 3, n)
    loss = torch.norm(y - T(sf.T)[None], p=2, dim=1).mean()
    ref = z[tag + '_flow']
    # north-star bar: EPE3D delta < 1e-4 on fixed inputs
    assert abs(float(loss.item()) - float(z[tag + '_loss'])) < 1e-4
    assert np.abs(y.detach().cpu().numpy()[0] - ref).max() < 2e-4 * max(1.0, np.abs(ref).max())
    loss.backward()
    names = bytes(z[tag + '_gradnames']).decode().split('\n')
    want = dict(zip(names, z[tag + '_gradnorm']))
    got = {k: float(p.grad.norm()) for k, p in m.named_parameters()}
    assert set(got) == set(want)
    scale = max(want.values())
    for k in want:
        assert abs(got[k] - want[k]) < 2e-3 * max(want[k], 1e-3 * scale), (k, got[k], want[k])
    # the device-built lattice takes the pair-batched Down path under autograd: same bar against the reference
    gen = H.GenerateDataUnsymmetric(model_args(nsc), device=DEV)
    lat = gen.build(p1[0], p2[0]).prepare(for_training=True)
    m.zero_grad(set_to_none=True)
    lossp = torch.norm(m(p1, p2, lat) - T(sf.T)[None], p=2, dim=1).mean()
    assert abs(float(lossp.item()) - float(z[tag + '_loss'])) < 1e-4
    lossp.backward()
    for k, p in m.named_parameters():
        assert abs(float(p.grad.norm()) - want[k]) < 2e-3 * max(want[k], 1e-3 * scale), (k, 'pair path')
    # inference path (no autograd, in-place channel-block writes) gives the same flow
    with torch.no_grad():
        y2 = m(p1, p2, gd_batched_device(gd[:nsc]))
    # (it also runs the last 1x1 conv of the Up layers after the slice, see bcl.py: rounding differs)
    assert float((y2 - y.detach()).abs().max()) < 2e-5 * max(1.0, float(y.detach().abs().max()))


@pytest.mark.parametrize('n,seed', [(256, 0), (1024, 0), (1024, 3), (8192, 0), (32768, 1)])
def test_device_lattice_bit_exact(n, seed):
    """GPU lattice == C oracle (== reference, tests/test_oracle_lattice.py) on every table."""
    import hplflownet_amd as H
    pc1, pc2, sf, gd = oracle_lattice(n, seed)
    gen = H.GenerateDataUnsymmetric(types.SimpleNamespace(dim=3, scales_filter_map=SCALES_FILTER_MAP), device=DEV)
    t1, t2, tsf, lat = gen([pc1, pc2, sf])
    assert tuple(t1.shape) == (3, n) and np.array_equal(tsf.cpu().numpy(), sf.T)
    ref = H.to_reference_format(lat)
    for l, (a, b) in enumerate(zip(ref, gd)):
        for k in b:
            va = a[k].cpu().numpy() if torch.is_tensor(a[k]) else a[k]
            assert np.array_equal(np.asarray(va), np.asarray(b[k])), (l, k)


def test_device_lattice_edge_cases():
    import hplflownet_amd as H
    from oracle import lattice_oracle as LO
    gen = H.GenerateDataUnsymmetric(types.SimpleNamespace(dim=3, scales_filter_map=SCALES_FILTER_MAP), device=DEV)
    one = np.array([[0.1, -0.2, 5.0]], np.float32)
    p1, p2, _ = synthetic_pair(64, 5)
    for a, b in ((one, one.copy()), (np.repeat(one, 7, 0), one.copy()), (p1, p2[:40])):
        _, _, _, lat = gen([a, b, np.zeros_like(a)])
        gd = LO.generate_data(a, b, SCALES_FILTER_MAP)
        for l, (x, y) in enumerate(zip(H.to_reference_format(lat), gd)):
            for k in y:
                vx = x[k].cpu().numpy() if torch.is_tensor(x[k]) else x[k]
                assert np.array_equal(np.asarray(vx), np.asarray(y[k])), (l, k)
    assert gen([None, None, None]) == (None, None, None, None)       # transforms.py:360-361


def test_config2_shallow_n4096_vs_oracle():
    """BASELINE config 2: HPLFlowNetShallow forward, N=4096, device lattice + HIP layers vs oracle."""
    import hplflownet_amd as H
    from oracle import bcl_oracle as BO
    n = 4096
    pc1, pc2, sf, gd = oracle_lattice(n, 0, nscales=5)
    args = model_args(5)
    gen = H.GenerateDataUnsymmetric(args, device=DEV)
    t1, t2, tsf, lat = gen([pc1, pc2, sf])
    m = H.HPLFlowNetShallow(args)
    fill_module_(m, 1.0, 'hash')
    sd = {k: v.numpy() for k, v in m.state_dict().items()}
    m = m.to(DEV)
    with torch.no_grad():
        y = m(t1[None], t2[None], lat)
    ref = BO.hplflownet_forward(sd, pc1.T, pc2.T, gd, shallow=True)
    got = y.cpu().numpy()[0]
    assert abs(BO.epe3d(got, sf.T) - BO.epe3d(ref, sf.T)) < 1e-4
    assert np.abs(got - ref).max() < 2e-4 * max(1.0, np.abs(ref).max())


def test_full_n8192_mfma_vs_naive_kernel_and_determinism():
    """Full HPLFlowNet forward at N=8192: the MFMA gather-GEMM against the one-thread-per-output HIP kernel
    (same C ABI, no matrix cores, k-ordered fmaf chain), bit-determinism across two runs, finiteness.  A
    self-consistency check of the kernel; parity with the reference / the oracle at this size is
    tests/test_gpu_bench_size.py."""
    import hplflownet_amd as H
    from hplflownet_amd import ops
    n = 8192
    pc1, pc2, sf = synthetic_pair(n, 0)
    args = model_args(7)
    gen = H.GenerateDataUnsymmetric(args, device=DEV)
    t1, t2, tsf, lat = gen([pc1, pc2, sf])
    assert [lv.H[0] for lv in lat.levels] == [25841, 34631, 9433, 1787, 426, 132, 53]
    m = H.HPLFlowNet(args)
    fill_module_(m, 1.0, 'hash')
    m = m.to(DEV)
    with torch.no_grad():
        y1 = m(t1[None], t2[None], lat)
        y2 = m(t1[None], t2[None], lat)
        assert torch.equal(y1, y2)
        orig = ops.gconv_raw

        def naive(*a, **k):
            k['naive'] = True
            return orig(*a, **k)
        ops.gconv_raw = naive
        try:
            y3 = m(t1[None], t2[None], lat)
        finally:
            ops.gconv_raw = orig
    assert torch.isfinite(y1).all()
    epe = lambda y: float(torch.norm(y - tsf[None], p=2, dim=1).mean())
    assert abs(epe(y1) - epe(y3)) < 1e-4
    assert float((y1 - y3).abs().max()) < 2e-4 * max(1.0, float(y3.abs().max()))


@pytest.mark.parametrize('cls,nsc,n1,n2', [('HPLFlowNet', 7, 1024, 1024), ('HPLFlowNetShallow', 5, 2048, 1500)])
def test_pair_batched_down_path_equals_per_cloud(cls, nsc, n1, n2):
    """Inference runs conv1 + the Down BCLs once per PAIR (stacked clouds, pair CSR, pair blur table);
    the per-cloud path (training, reference-format lattices) must give the same flow.  Rows of a
    GEMM are independent and a vertex belongs to one cloud, so only the split-K decisions differ."""
    import hplflownet_amd as H
    from hplflownet_amd import ops
    pc1, pc2, sf = synthetic_pair(max(n1, n2), 2)
    pc1, pc2 = pc1[:n1], pc2[:n2]
    args = model_args(nsc)
    gen = H.GenerateDataUnsymmetric(args, device=DEV)
    t1, t2, _, lat = gen([pc1, pc2, pc1])
    # pair CSR == per-cloud CSRs laid end to end
    for lv in lat.levels:
        fresh = [ops.CloudTables(c.bary, c.off, c.H) for c in lv.clouds]          # independent per-cloud builds
        p, q0, q1 = lv.pair.csr(), fresh[0].csr(), fresh[1].csr()
        for c, q in zip(lv.clouds, (q0, q1)):                                      # derived from the pair CSR
            assert all(torch.equal(x, y) for x, y in zip(c.csr(), q))
        assert torch.equal(p[0][:lv.H[0] + 1], q0[0]) and torch.equal(p[0][lv.H[0]:] - p[0][lv.H[0]], q1[0])
        assert torch.equal(p[1], torch.cat([q0[1], q1[1] + lv.clouds[0].N]))
        assert torch.equal(p[2], torch.cat([q0[2], q1[2]])) and torch.equal(p[3], torch.cat([q0[3], q1[3]]))
    m = getattr(H, cls)(args)
    fill_module_(m, 1.0, 'hash')
    m = m.to(DEV).eval()
    with torch.no_grad():
        y_pair = m(t1[None], t2[None], lat)
        m.pair_batched = False
        y_sep = m(t1[None], t2[None], lat)
        y_ref = m(t1[None], t2[None], H.to_reference_format(lat))      # per-cloud, reference wire format
    assert torch.equal(y_sep, y_ref)
    scale = max(1.0, float(y_sep.abs().max()))
    assert float((y_pair - y_sep).abs().max()) < 1e-5 * scale
    # training: the same stacked Down path under autograd -- loss and every parameter gradient agree
    sf = torch.from_numpy(np.ascontiguousarray((pc2[:n1] - pc1[:n1]).T if n2 >= n1 else pc1.T * 0.1)).to(DEV)
    grads = []
    for pair_mode in (True, False):
        m.pair_batched = pair_mode
        m.zero_grad(set_to_none=True)
        lat_t = gen.build(t1, t2).prepare(for_training=True)
        loss = torch.norm(m(t1[None], t2[None], lat_t) - sf[None], p=2, dim=1).mean()
        loss.backward()
        grads.append((float(loss.detach()), {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None}))
    (la, ga), (lb, gb) = grads
    assert abs(la - lb) < 1e-5 * max(1.0, abs(lb)) and set(ga) == set(gb)
    scale = max(float(g.norm()) for g in gb.values())
    # Same sums in another order + LeakyReLU kinks deep in the chain.  Noise floor measured on the per-cloud path
    # alone by scaling all weights by (1 + 1e-7): worst entry moves by 4.1e-3 of the largest entry of its tensor
    # (bcn3_ weight, few vertices per entry), worst norm by 5.4e-4 -- the pair path differs by exactly as much.
    for k in gb:
        assert abs(float(ga[k].norm()) - float(gb[k].norm())) <= 2e-3 * max(float(gb[k].norm()), 1e-3 * scale), k
        assert float((ga[k] - gb[k]).abs().max()) <= 1e-2 * float(gb[k].abs().max()) + 1e-7, k


def test_device_lattice_fuzz_vs_oracle():
    """Randomised clouds (sizes 1..300, ragged pairs, duplicated points, points on a line / plane, huge and
    tiny coordinates): every table of the GPU lattice equals the C oracle bit for bit."""
    import hplflownet_amd as H
    from hypothesis import given, settings, strategies as st
    from oracle import lattice_oracle as LO
    gen = H.GenerateDataUnsymmetric(types.SimpleNamespace(dim=3, scales_filter_map=SCALES_FILTER_MAP), device=DEV)

    @settings(max_examples=25, deadline=None, derandomize=True)
    @given(st.integers(1, 300), st.integers(1, 300), st.integers(0, 2 ** 31 - 1),
           st.sampled_from(['cloud', 'dup', 'line', 'plane', 'far', 'tiny']))
    def check(n1, n2, seed, kind):
        rng = np.random.RandomState(seed)
        p1 = rng.uniform(-8, 8, (n1, 3)).astype(np.float32)
        p1[:, 2] = rng.uniform(1.5, 35, n1)
        p2 = rng.uniform(-8, 8, (n2, 3)).astype(np.float32)
        p2[:, 2] = rng.uniform(1.5, 35, n2)
        if kind == 'dup':
            p1[:] = p1[rng.randint(0, max(1, n1 // 4), n1)]
            p2[: min(n1, n2)] = p1[: min(n1, n2)]
        elif kind == 'line':
            p1[:, :2] = 0.0
            p2[:, 1:] = p2[0, 1:]
        elif kind == 'plane':
            p1[:, 2] = 10.0
            p2[:, 0] = -1.0
        elif kind == 'far':
            p1 *= 40.0
            p2 *= 40.0
        elif kind == 'tiny':
            p1 *= 1e-3
            p2 *= 1e-3
        _, _, _, lat = gen([p1, p2, np.zeros_like(p1)])
        gd = LO.generate_data(p1, p2, SCALES_FILTER_MAP)
        for l, (x, y) in enumerate(zip(H.to_reference_format(lat), gd)):
            for k in y:
                vx = x[k].cpu().numpy() if torch.is_tensor(x[k]) else x[k]
                assert np.array_equal(np.asarray(vx), np.asarray(y[k])), (n1, n2, seed, kind, l, k)

    check()


def test_single_cloud_lattice_entries_match_pair_entry():
    """hpl_lattice_keys / hpl_lattice_next_points (one cloud, (3, N) points materialised) give the same
    keys / barycentric / el_minus_gr as hpl_lattice_keys_pair, which the builder uses (both clouds per
    launch, next level's points computed from the vertex keys inside the kernel)."""
    from hplflownet_amd import _lib
    from hplflownet_amd._lib import check, ptr, stream
    L = _lib.load()
    rng = np.random.RandomState(4)
    n = (700, 513)
    pts = [torch.from_numpy(rng.uniform(-6, 6, (3, m)).astype(np.float32)).to(DEV) for m in n]
    scale = 1.5

    def alloc(m):
        return (torch.empty((4, m, 4), dtype=torch.int32, device=DEV), torch.empty((4, m), device=DEV),
                torch.empty((m, 4), device=DEV))
    single = [alloc(m) for m in n]
    for c in (0, 1):
        check(L.hpl_lattice_keys(ptr(pts[c]), n[c], scale, ptr(single[c][0]), ptr(single[c][1]), ptr(single[c][2]), 4,
                                 stream()), 'keys')
    pair = [alloc(m) for m in n]
    check(L.hpl_lattice_keys_pair(ptr(pts[0]), ptr(pts[1]), None, None, 0, 0, 1.0, n[0], n[1], scale, ptr(pair[0][0]),
                                  ptr(pair[1][0]), ptr(pair[0][1]), ptr(pair[1][1]), ptr(pair[0][2]), ptr(pair[1][2]), 4,
                                  stream()), 'keys_pair')
    for c in (0, 1):
        for a, b in zip(single[c], pair[c]):
            assert torch.equal(a, b)
    # vertex-key mode == next_points followed by the point mode
    vk = [torch.from_numpy(rng.randint(-40, 40, (4, 4 * m)).astype(np.int32)).to(DEV) for m in n]
    Hn = (300, 250)
    div = 3.25
    nxt = [torch.empty((3, h), device=DEV) for h in Hn]
    for c in (0, 1):
        check(L.hpl_lattice_next_points(ptr(vk[c]), 4 * n[c], Hn[c], div, ptr(nxt[c]), stream()), 'next_points')
    ref = [alloc(h) for h in Hn]
    check(L.hpl_lattice_keys_pair(ptr(nxt[0]), ptr(nxt[1]), None, None, 0, 0, 1.0, Hn[0], Hn[1], scale, ptr(ref[0][0]),
                                  ptr(ref[1][0]), ptr(ref[0][1]), ptr(ref[1][1]), ptr(ref[0][2]), ptr(ref[1][2]), 4,
                                  stream()), 'keys_pair')
    got = [alloc(h) for h in Hn]
    check(L.hpl_lattice_keys_pair(None, None, ptr(vk[0]), ptr(vk[1]), 4 * n[0], 4 * n[1], div, Hn[0], Hn[1], scale,
                                  ptr(got[0][0]), ptr(got[1][0]), ptr(got[0][1]), ptr(got[1][1]), ptr(got[0][2]),
                                  ptr(got[1][2]), 4, stream()), 'keys_pair')
    for c in (0, 1):
        for a, b in zip(ref[c], got[c]):
            assert torch.equal(a, b)


@pytest.mark.parametrize('depth,training', [(1, False), (4, False), (3, True)])
def test_lattice_pipeline_equals_blocking_build(depth, training):
    """Several pairs under construction at once on a side stream (asynchronous read-backs of the vertex
    counts) give, pair by pair and in order, exactly the tables of the blocking build; each pair is
    fetched once; symmetry verdicts are resolved when the lattice is built for training."""
    import hplflownet_amd as H
    from hplflownet_amd.lattice import LatticeBuild, LatticePipeline
    gen = H.GenerateDataUnsymmetric(types.SimpleNamespace(dim=3, scales_filter_map=SCALES_FILTER_MAP), device=DEV)
    sizes = [300, 64, 1000, 17, 512, 128]
    pairs, fetched = [], []
    for s, n in enumerate(sizes):
        p1, p2, _ = synthetic_pair(n, 20 + s)
        pairs.append((torch.from_numpy(p1.T.copy()).to(DEV), torch.from_numpy(p2[: max(1, n - s)].T.copy()).to(DEV)))

    def source(i):
        fetched.append(i)
        return pairs[i]
    side = torch.cuda.Stream()
    pipe = LatticePipeline(gen, source, 1, 5, depth=depth, stream=side, for_training=training)
    for want in range(1, 6):
        (i, item), lat, ev = pipe.get()
        assert i == want and item is pairs[i]
        ev.synchronize()
        ref = H.to_reference_format(gen.build(*pairs[i]))
        for l, (x, y) in enumerate(zip(H.to_reference_format(lat), ref)):
            for k in y:
                same = torch.equal(x[k], y[k]) if torch.is_tensor(y[k]) else x[k] == y[k]
                assert same, (i, l, k)
        if training:
            assert all(lv.blur.pair._sym is not None for lv in lat.levels)
    assert fetched == [1, 2, 3, 4, 5]
    with pytest.raises(StopIteration):
        pipe.get()
    # a single build driven by hand: ready() / advance() / finish()
    b = LatticeBuild(gen, pairs[0][0], pairs[0][1], stream=side, prepare=False)
    steps = 0
    while not b.advance():
        steps += 1
    assert steps == len(SCALES_FILTER_MAP) and b.ready() and b.finish() is b.result


def test_radius2_layers_vs_oracle():
    """neighborhood_size / corr radii 2 (65 taps; the reference supports any radius, its configs use 1): the
    contraction runs as ceil(65 / 15) accumulating passes of the same kernel.  BilateralConvFlex forward and
    every gradient, BilateralCorrelationFlex forward, against the numpy oracle on a device-built lattice."""
    import hplflownet_amd as H
    from oracle import bcl_oracle as BO
    sfm = [[2.0, 2, -1, -1], [1.0, 2, 2, 2]]
    pc1, pc2, _ = synthetic_pair(160, 11)
    gen = H.GenerateDataUnsymmetric(types.SimpleNamespace(dim=3, scales_filter_map=sfm), device=DEV)
    _, _, _, lat = gen([pc1, pc2[:150], pc1])
    gd = [{k: (v.cpu().numpy() if torch.is_tensor(v) else v) for k, v in d.items()} for d in H.to_reference_format(lat)]
    g = gd[0]
    assert g['pc1_blur_neighbors'].shape[0] == 65
    rng = np.random.RandomState(0)
    for cin, couts, do_splat, do_slice in ((20, [24, 16], True, True), (33, [40], False, False)):
        m = H.BilateralConvFlex(3, 2, cin, couts, 'cuda', True, True, True, do_splat, do_slice, False, chunk_size=-1).to(DEV)
        with torch.no_grad():
            for p in m.parameters():
                p.copy_(torch.from_numpy(rng.randn(*p.shape).astype(np.float32)) * (0.5 if p.dim() == 1 else 0.05))
        n_in = g['pc1_barycentric'].shape[1] if do_splat else g['pc1_hash_cnt']
        x = T(rng.randn(1, cin, n_in).astype(np.float32)).requires_grad_(True)
        y = m(x, T(g['pc1_barycentric'])[None] if do_splat else None, T(g['pc1_lattice_offset'])[None] if do_splat else None,
              T(g['pc1_blur_neighbors'])[None], T(g['pc1_barycentric'])[None] if do_slice else None,
              T(g['pc1_lattice_offset'])[None] if do_slice else None)
        convs = []
        for mod in m.blur_conv:
            conv = mod.conv if hasattr(mod, 'conv') else mod
            W = conv.weight.detach().cpu().numpy()
            convs.append((W.reshape(W.shape[0], W.shape[1], -1), conv.bias.detach().cpu().numpy()))
        bias = m.bias.detach().cpu().numpy() if do_slice else None
        yo, cache = BO.bilateral_conv_forward(x.detach().cpu().numpy()[0], convs, bias, g['pc1_barycentric'],
                                              g['pc1_lattice_offset'], g['pc1_blur_neighbors'], g['pc1_barycentric'],
                                              g['pc1_lattice_offset'], do_splat, do_slice)
        assert rel_err(y.detach().cpu().numpy()[0], yo) < 1e-5
        go = T(rng.randn(*y.shape).astype(np.float32))
        (y * go).sum().backward()
        gr = BO.bilateral_conv_backward(go.cpu().numpy()[0], cache, x.detach().cpu().numpy()[0], convs, bias,
                                        g['pc1_barycentric'], g['pc1_lattice_offset'], g['pc1_blur_neighbors'],
                                        g['pc1_barycentric'], g['pc1_lattice_offset'], do_splat, do_slice)
        assert rel_err(x.grad.cpu().numpy()[0], gr['features']) < 1e-4
        for mod, (gW, gb) in zip(m.blur_conv, gr['convs']):
            conv = mod.conv if hasattr(mod, 'conv') else mod
            assert rel_err(conv.weight.grad.cpu().numpy().reshape(gW.shape), gW) < 1e-4
            assert rel_err(conv.bias.grad.cpu().numpy(), gb) < 1e-4
    # correlation layer with 65 x 65 patch taps on the second level
    g = gd[1]
    H1, H2 = g['pc1_hash_cnt'], g['pc2_hash_cnt']
    assert g['pc2_corr_indices'].shape[:2] == (65, 65)
    mc = H.BilateralCorrelationFlex(3, 2, 2, 8, [8, 8], [16, 16], 'cuda', True, True, True, 0, False, chunk_size=-1).to(DEV)
    with torch.no_grad():
        for p in mc.parameters():
            p.copy_(torch.from_numpy(rng.randn(*p.shape).astype(np.float32)) * (0.5 if p.dim() == 1 else 0.05))
    f1, f2 = T(rng.randn(1, 8, H1).astype(np.float32)), T(rng.randn(1, 8, H2).astype(np.float32))
    with torch.no_grad():
        yc = mc(f1, f2, None, None, None, T(g['pc1_corr_indices'])[None], T(g['pc2_corr_indices'])[None], H1, H2)
    cc = []
    for mod in mc.corr_conv:
        W = mod.conv.weight.detach().cpu().numpy()
        cc.append((W.reshape(W.shape[0], W.shape[1], -1), mod.conv.bias.detach().cpu().numpy()))
    bc = []
    for mod in mc.blur_conv:
        conv = mod.conv if hasattr(mod, 'conv') else mod
        W = conv.weight.detach().cpu().numpy()
        bc.append((W.reshape(W.shape[0], W.shape[1], -1), conv.bias.detach().cpu().numpy()))
    yco = BO.bilateral_corr_forward(f1.cpu().numpy()[0], f2.cpu().numpy()[0], None, g['pc1_barycentric'],
                                    g['pc1_lattice_offset'], g['pc1_corr_indices'], g['pc2_corr_indices'], cc, bc)
    assert rel_err(yc.cpu().numpy()[0], yco) < 1e-5
