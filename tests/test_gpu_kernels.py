"""GPU: every HIP kernel through the C ABI against numpy/oracle on seeded inputs.
Integer outputs bit-exact; fp32 outputs within 1e-5 (relative to the tensor's max-abs) of a
float64 evaluation unless a looser bound is stated next to the assert."""
import numpy as np
import pytest
import torch

from common import oracle_lattice, rel_err

pytestmark = pytest.mark.gpu

DEV = 'cuda'


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


@pytest.fixture(scope='module')
def ops():
    from hplflownet_amd import ops as o
    return o


def test_device_is_gfx950():
    import ctypes
    from hplflownet_amd import _lib
    cu, ws = ctypes.c_int(0), ctypes.c_int(0)
    arch = ctypes.create_string_buffer(64)
    _lib.check(_lib.load().hpl_device_info(0, ctypes.byref(cu), ctypes.byref(ws), arch, 64), 'hpl_device_info')
    assert arch.value.decode().startswith('gfx950'), arch.value
    assert ws.value == 64 and cu.value == 256


def test_narrow_and_permute(ops):
    rng = np.random.RandomState(0)
    a = rng.randint(-1, 5000, size=(15, 777)).astype(np.int64)
    assert np.array_equal(ops.narrow(dev(a)).cpu().numpy(), a.astype(np.int32))
    c2 = rng.randint(-1, 900, size=(15, 15, 333)).astype(np.int64)
    want = c2.transpose(1, 0, 2).reshape(15, 15 * 333).astype(np.int32)
    assert np.array_equal(ops.corr2_permute(dev(c2)).cpu().numpy(), want)
    assert np.array_equal(ops.corr2_permute(dev(c2.astype(np.int32))).cpu().numpy(), want)


@pytest.mark.parametrize('n,level', [(256, 0), (256, 2), (1024, 1)])
def test_csr_build(ops, n, level):
    _, _, _, gd = oracle_lattice(n)
    g = gd[level]
    H = g['pc1_hash_cnt']
    off, bary = g['pc1_lattice_offset'], g['pc1_barycentric']
    ct = ops.CloudTables(dev(bary), dev(off), H)
    csr_ptr, csr_pt, csr_w, norm = [t.cpu().numpy() for t in ct.csr()]
    N = off.shape[1]
    flat = off.reshape(-1)
    order = np.argsort(flat, kind='stable')                 # ascending vertex, then ascending entry
    cnt = np.bincount(flat, minlength=H)
    assert np.array_equal(csr_ptr, np.concatenate([[0], np.cumsum(cnt)]).astype(np.int32))
    assert np.array_equal(csr_pt, (order % N).astype(np.int32))
    assert np.array_equal(csr_w, bary.reshape(-1)[order])
    w = np.zeros(H, np.float64)
    np.add.at(w, flat, bary.reshape(-1).astype(np.float64))
    assert rel_err(norm, 1.0 / (w + 1e-5)) < 1e-6


def test_csr_build_beyond_one_scan_chunk(ops):
    """More than 2^20 vertices: the exclusive scan runs in chunks that continue from each other's totals."""
    rng = np.random.RandomState(3)
    N, H = 450000, (1 << 20) + 300001
    off = rng.randint(0, H, (4, N)).astype(np.int32)
    off[0, :5] = [0, H - 1, (1 << 20) - 1, 1 << 20, (1 << 20) + 1]          # both sides of the chunk seam
    bary = rng.uniform(0.05, 1.0, (4, N)).astype(np.float32)
    csr_ptr, csr_pt, csr_w, _ = [t.cpu().numpy() for t in ops.CloudTables(dev(bary), dev(off), H).csr()]
    flat = off.reshape(-1)
    order = np.argsort(flat, kind='stable')
    assert np.array_equal(csr_ptr, np.concatenate([[0], np.cumsum(np.bincount(flat, minlength=H))]).astype(np.int32))
    assert np.array_equal(csr_pt, (order % N).astype(np.int32)) and np.array_equal(csr_w, bary.reshape(-1)[order])


@pytest.mark.parametrize('C', [68, 64, 4, 12, 5, 33, 128, 1024])
def test_splat_and_slice(ops, C):
    from oracle import bcl_oracle as BO
    _, _, _, gd = oracle_lattice(256)
    g = gd[1]
    H, off, bary = g['pc1_hash_cnt'], g['pc1_lattice_offset'], g['pc1_barycentric']
    N = off.shape[1]
    rng = np.random.RandomState(C)
    feat = rng.randn(N, C).astype(np.float32)
    ct = ops.CloudTables(dev(bary), dev(off), H)
    for use_norm in (True, False):
        S = ops.splat_raw(dev(feat), ct.csr(), H, use_norm).cpu().numpy()
        want = BO.splat(feat.T.astype(np.float64), bary.astype(np.float64), off, H, use_norm)[:, 1:].T
        assert rel_err(S, want) < 1e-5, (C, use_norm)
    Y = rng.randn(H, C).astype(np.float32)
    bias = rng.randn(C).astype(np.float32)
    out = ops.slice_raw(dev(Y), ct.bary, ct.off, N, bias=dev(bias)).cpu().numpy()
    want = (bary[:, :, None].astype(np.float64) * Y[off].astype(np.float64)).sum(0) + bias[None]
    assert rel_err(out, want) < 1e-5
    vs = rng.rand(H).astype(np.float32)
    out = ops.slice_raw(dev(Y), ct.bary, ct.off, N, vscale=dev(vs)).cpu().numpy()
    want = ((bary * vs[off])[:, :, None].astype(np.float64) * Y[off]).sum(0)
    assert rel_err(out, want) < 1e-5
    # strided views: read a column slice of a wider buffer, write into another
    wide = torch.zeros(N, C + 8, device=DEV)
    wide[:, 4:4 + C] = dev(feat)
    dst = torch.full((H, C + 4), 7.0, device=DEV)
    ops.splat_raw(wide[:, 4:4 + C], ct.csr(), H, True, out=dst[:, :C])
    assert np.array_equal(dst[:, :C].cpu().numpy(), ops.splat_raw(dev(feat), ct.csr(), H, True).cpu().numpy())
    assert float(dst[:, C:].min()) == 7.0


def _gconv_ref(A, nbr, M, C, F, W, bias, res, res_mod, act, slope=0.1):
    """float64 reference: W is (O, C, F)."""
    Ap = np.concatenate([A[:, :C].astype(np.float64), np.zeros((1, C))], axis=0)    # row -1 -> zeros
    if nbr is None:
        X = Ap[:M][None]
    else:
        X = Ap[nbr]                                                                 # (F, M, C)
    y = np.einsum('fmc,ocf->mo', X, W.astype(np.float64))
    if bias is not None:
        y = y + bias[None].astype(np.float64)
    if res is not None:
        y = y + res[np.arange(M) % res_mod].astype(np.float64)
    if act:
        y = np.where(y > 0, y, slope * y)
    return y


GCONV_CASES = [
    # M, rowsA, C, F, O, table, bias, res, act
    (300, 300, 68, 15, 64, True, True, False, True),       # Down blur conv, 64x64 tile
    (1000, 1000, 64, 1, 64, False, True, False, False),    # 1x1 conv, dense
    (700, 650, 36, 15, 128, True, True, False, True),      # 128x128 tile, table into a different row set
    (513, 513, 580, 15, 200, True, False, False, False),   # wide C, K tail (8700 % 32 != 0), N tail
    (2000, 2000, 3, 1, 32, False, True, False, True),      # conv1 first layer: C=3 scalar path
    (900, 900, 32, 1, 3, False, True, False, False),       # conv4: N=3
    (15 * 97, 120, 64, 15, 32, True, True, True, True),    # corr B-term: virtual vertices + broadcast residual
    (97, 15 * 97, 32, 15, 64, True, True, False, True),    # displacement filter through a regular table
    (70000, 70000, 68, 15, 64, True, True, False, True),   # enough tiles for the 128-row configs
    (70000, 500, 64, 15, 32, True, False, False, False),   # 128x32 config
    (33, 40, 8, 15, 16, True, True, False, True),          # tiny
]


@pytest.mark.parametrize('case', GCONV_CASES, ids=[str(i) for i in range(len(GCONV_CASES))])
def test_gconv_forward(ops, case):
    M, rows, C, F, O, table, has_bias, has_res, act = case
    rng = np.random.RandomState(M + C + O)
    A = rng.randn(rows, C).astype(np.float32)
    W = (rng.randn(O, C, F) / np.sqrt(C * F)).astype(np.float32)
    bias = rng.randn(O).astype(np.float32) if has_bias else None
    nbr = None
    if table:
        nbr = rng.randint(-1, rows, size=(F, M)).astype(np.int32)
        nbr[rng.rand(F, M) < 0.3] = -1
    res_mod = 97 if has_res else 0
    res = rng.randn(res_mod, O).astype(np.float32) if has_res else None
    Wd = dev(W)
    Wt = ops.weight_relayout(Wd, C, O, F, F, C * F, 1)
    args = dict(bias=dev(bias) if has_bias else None, act=1 if act else 0,
                res=dev(res) if has_res else None, res_mod=res_mod)
    y = ops.gconv_raw(dev(A), dev(nbr) if table else None, M, C, F, Wt, O, **args).cpu().numpy()
    yn = ops.gconv_raw(dev(A), dev(nbr) if table else None, M, C, F, Wt, O, naive=True, **args).cpu().numpy()
    if M * C * F * O < 4e9:
        want = _gconv_ref(A, nbr, M, C, F, W, bias, res, res_mod, act)
        assert rel_err(yn, want) < 1e-5
        assert rel_err(y, want) < 1e-5
    # the MFMA is a k-ordered fmaf chain like the naive kernel: expect (near) bit equality -- unless the
    # slice list was split over workgroups (small problems), which re-associates the sum
    y1 = ops.gconv_raw(dev(A), dev(nbr) if table else None, M, C, F, Wt, O, split_k=False, **args).cpu().numpy()
    assert rel_err(y1, yn) < 1e-6
    assert rel_err(y, yn) < 1e-5
    # determinism: same launch twice, bit-identical
    y2 = ops.gconv_raw(dev(A), dev(nbr) if table else None, M, C, F, Wt, O, **args).cpu().numpy()
    assert np.array_equal(y, y2)


def test_gconv_strided_io(ops):
    rng = np.random.RandomState(3)
    M, C, F, O = 500, 64, 15, 64
    wide = torch.zeros(M, 4 + C + 12, device=DEV)
    A = rng.randn(M, C).astype(np.float32)
    wide[:, 4:4 + C] = dev(A)
    nbr = rng.randint(-1, M, size=(F, M)).astype(np.int32)
    W = (rng.randn(O, C, F) / 30).astype(np.float32)
    Wt = ops.weight_relayout(dev(W), C, O, F, F, C * F, 1)
    dst = torch.full((M, O + 8), -3.0, device=DEV)
    ops.gconv_raw(wide[:, 4:4 + C], dev(nbr), M, C, F, Wt, O, out=dst[:, 4:4 + O])
    want = ops.gconv_raw(dev(A), dev(nbr), M, C, F, Wt, O)
    assert torch.equal(dst[:, 4:4 + O], want)
    assert float(dst[:, :4].max()) == -3.0 and float(dst[:, 4 + O:].max()) == -3.0


def test_weight_relayout_roundtrip(ops):
    from hplflownet_amd import _lib
    rng = np.random.RandomState(5)
    O, Ctot, F, c0, C = 24, 40, 15, 8, 20
    W = rng.randn(O, Ctot, F).astype(np.float32)
    Wt = ops.weight_relayout(dev(W), C, O, F, F, Ctot * F, 1, base=c0 * F).cpu().numpy()
    want = np.zeros_like(Wt)
    want[:F * C, :O] = W[:, c0:c0 + C, :].transpose(2, 1, 0).reshape(F * C, O)
    assert np.array_equal(Wt, want)
    fmap = ((F - np.arange(F)) % F).astype(np.int32)
    WtT = ops.weight_relayout(dev(W), O, C, F, Ctot * F, F, 1, base=c0 * F, fmap=dev(fmap)).cpu().numpy()
    want = np.zeros_like(WtT)
    for f in range(F):
        want[fmap[f] * O:(fmap[f] + 1) * O, :C] = W[:, c0:c0 + C, f]
    assert np.array_equal(WtT, want)
    back = torch.zeros(O, Ctot, F, device=DEV)
    Wt_d = ops.weight_relayout(dev(W), C, O, F, F, Ctot * F, 1, base=c0 * F)
    _lib.check(_lib.load().hpl_weight_unlayout(Wt_d.data_ptr(), Wt_d.shape[1], C, O, F, back.data_ptr(), c0 * F, F,
                                               Ctot * F, 1, 0, _lib.stream()), 'unlayout')
    want = np.zeros_like(W)
    want[:, c0:c0 + C] = W[:, c0:c0 + C]
    assert np.array_equal(back.cpu().numpy(), want)


@pytest.mark.parametrize('O,Ctot,F,c0,C', [(24, 40, 15, 8, 20), (130, 68, 15, 0, 68), (64, 3, 1, 0, 3), (513, 260, 1, 4, 250),
                                           (33, 70, 15, 1, 69), (100, 100, 7, 0, 100), (5, 9, 27, 2, 6)])
def test_weight_bank_and_unlayout_every_layout(ops, O, Ctot, F, c0, C):
    """The LDS-staged batch re-layout (forward image, mirrored data-gradient image, any strides) against the per-element
    kernel of hpl_weight_relayout, and the staged un-layout (plain + accumulate) against numpy -- bit for bit."""
    from hplflownet_amd import _lib
    rng = np.random.RandomState(O * 31 + C)
    W = dev(rng.randn(O, Ctot, F).astype(np.float32))
    W2 = dev(rng.randn(O + 3, Ctot, F).astype(np.float32))
    bank = ops.WeightBank()
    reqs = [(W, C, O, F, F, Ctot * F, 1, c0 * F, False), (W, O, C, F, Ctot * F, F, 1, c0 * F, True),
            (W2, C, O + 3, F, F, Ctot * F, 1, c0 * F, False), (W2, O + 3, C, F, Ctot * F, F, 1, c0 * F, False),
            (W, C, O, F, F, Ctot * F, 1, c0 * F, True)]
    if F > 1:
        reqs.append((W, O, F, C, Ctot * F, 1, F, c0 * F, False))          # neither layout: the generic path
    for w, R, Q, Fk, sr, sq, sf, base, mirror in reqs:
        bank.get(w, R, Q, Fk, sr, sq, sf, base, mirror)
    bank.refresh()
    for w, R, Q, Fk, sr, sq, sf, base, mirror in reqs:
        img = bank.get(w, R, Q, Fk, sr, sq, sf, base, mirror)
        assert getattr(img, '_hpl_bank_job', None) is not None          # served from the bank
        fmap = dev(((Fk - np.arange(Fk)) % Fk).astype(np.int32)) if mirror else None
        want = ops.weight_relayout(w, R, Q, Fk, sr, sq, sf, base=base, fmap=fmap)
        assert torch.equal(img, want), (R, Q, Fk, sr, sq, sf, mirror)
    Wt = ops.weight_relayout(W, C, O, F, F, Ctot * F, 1, base=c0 * F)
    back = torch.full((O, Ctot, F), 2.0, device=DEV)
    for acc in (0, 1):
        _lib.check(_lib.load().hpl_weight_unlayout(Wt.data_ptr(), Wt.shape[1], C, O, F, back.data_ptr(), c0 * F, F,
                                                   Ctot * F, 1, acc, _lib.stream()), 'unlayout')
        want = np.full((O, Ctot, F), 2.0, np.float32)
        want[:, c0:c0 + C] = W.cpu().numpy()[:, c0:c0 + C] * (1 + acc)
        assert np.array_equal(back.cpu().numpy(), want)
    # the data-gradient layout as destination (q contiguous in W)
    WtT = ops.weight_relayout(W, O, C, F, Ctot * F, F, 1, base=c0 * F)
    back = torch.zeros(O, Ctot, F, device=DEV)
    _lib.check(_lib.load().hpl_weight_unlayout(WtT.data_ptr(), WtT.shape[1], O, C, F, back.data_ptr(), c0 * F, Ctot * F, F, 1, 0,
                                               _lib.stream()), 'unlayout')
    want = np.zeros((O, Ctot, F), np.float32)
    want[:, c0:c0 + C] = W.cpu().numpy()[:, c0:c0 + C]
    assert np.array_equal(back.cpu().numpy(), want)


@pytest.mark.parametrize('M,rows,C,F,O', [(3000, 3000, 68, 15, 64), (5000, 5000, 32, 1, 200), (900, 900, 3, 1, 32),
                                          (40000, 40000, 36, 15, 32)])
def test_wgrad_colsum_leaky(ops, M, rows, C, F, O):
    rng = np.random.RandomState(M)
    A = rng.randn(rows, C).astype(np.float32)
    nbr = rng.randint(-1, rows, size=(F, M)).astype(np.int32) if F > 1 else None
    dY = rng.randn(M, O).astype(np.float32)
    dWt, gb = ops.wgrad_raw(dev(A), dev(nbr) if nbr is not None else None, M, C, F, dev(dY), O, want_bias=True)
    assert rel_err(gb.cpu().numpy(), dY.astype(np.float64).sum(0)) < 2e-5      # bias gradient from the same launch
    dWt = dWt.cpu().numpy()
    Ap = np.concatenate([A.astype(np.float64), np.zeros((1, C))], 0)
    X = Ap[nbr] if nbr is not None else Ap[:M][None]                 # (F, M, C)
    want = np.einsum('fmc,mo->fco', X, dY.astype(np.float64)).reshape(F * C, O)
    assert rel_err(dWt[:F * C, :O], want) < 2e-5                     # fp32 atomics over vertex slabs
    assert not dWt[F * C:].any() and not dWt[:, O:].any()
    assert rel_err(ops.colsum(dev(dY)).cpu().numpy(), dY.astype(np.float64).sum(0)) < 2e-5
    Y = rng.randn(M, O).astype(np.float32)
    got = ops.leaky_bwd(dev(dY), dev(Y)).cpu().numpy()
    assert np.array_equal(got, dY * np.where(Y > 0, np.float32(1), np.float32(0.1)))


@pytest.mark.parametrize('M,rows,C,F,O,density', [(6000, 6000, 260, 15, 128, 0.4), (5000, 4000, 128, 15, 40, 0.7),
                                                  (2500, 2500, 580, 15, 96, 0.1)])
def test_tap_lists_and_wgrad_tap_mode(ops, M, rows, C, F, O, density):
    """Per-tap lists of present vertices (hpl_tap_lists) and the weight gradient summed over them
    (tap-aligned k tiles): same result as the full vertex loop and as float64."""
    rng = np.random.RandomState(C + M)
    A = rng.randn(rows, C).astype(np.float32)
    nbr = np.where(rng.rand(F, M) < density, rng.randint(0, rows, size=(F, M)), -1).astype(np.int32)
    nbr[0] = np.arange(M) % rows                                     # centre tap always present
    if F * M > 3:
        nbr[3, :] = -1                                               # an empty tap
    dY = rng.randn(M, O).astype(np.float32)
    dn = dev(nbr)
    lst, lrow, tp = ops.tap_lists(dn)
    tp_h = tp.cpu().numpy()
    cnt = (nbr >= 0).sum(1)
    assert np.array_equal(tp_h, np.concatenate([[0], np.cumsum(cnt)]))
    lst_h, lrow_h = lst.cpu().numpy(), lrow.cpu().numpy()
    for f in range(F):
        present = np.nonzero(nbr[f] >= 0)[0]
        assert np.array_equal(lst_h[tp_h[f]:tp_h[f + 1]], present)
        assert np.array_equal(lrow_h[tp_h[f]:tp_h[f + 1]], nbr[f][present])
    got, gb = ops.wgrad_raw(dev(A), dn, M, C, F, dev(dY), O, taps=(lst, lrow, tp), want_bias=True)
    assert rel_err(gb.cpu().numpy(), dY.astype(np.float64).sum(0)) < 2e-5
    got = got.cpu().numpy()
    full = ops.wgrad_raw(dev(A), dn, M, C, F, dev(dY), O).cpu().numpy()
    Ap = np.concatenate([A.astype(np.float64), np.zeros((1, C))], 0)
    want = np.einsum('fmc,mo->fco', Ap[nbr], dY.astype(np.float64)).reshape(F * C, O)
    assert rel_err(got[:F * C, :O], want) < 2e-5
    assert rel_err(got, full) < 2e-5
    assert not got[F * C:].any() and not got[:, O:].any()


def test_gconv_scatter_epilogue(ops):
    rng = np.random.RandomState(11)
    M, rows, O, F, C = 400, 300, 32, 15, 16
    g = rng.randn(M, O).astype(np.float32)
    W = rng.randn(O, F * C).astype(np.float32) / 6
    nbr = rng.randint(-1, rows, size=(F, M)).astype(np.int32)
    Wt = ops.weight_relayout(dev(W), O, F * C, 1, F * C, 1, 1)
    out = torch.zeros(rows, C, device=DEV)
    ops.gconv_raw(dev(g), None, M, O, 1, Wt, F * C, out=out, scat=dev(nbr), scat_c=C)
    G = (g.astype(np.float64) @ W.astype(np.float64)).reshape(M, F, C)
    want = np.zeros((rows, C))
    for f in range(F):
        ok = nbr[f] >= 0
        np.add.at(want, nbr[f][ok], G[ok, f])
    assert rel_err(out.cpu().numpy(), want) < 2e-5


def test_full_size_properties(ops):
    """BASELINE size (N=8192 level-0 tables from the oracle): size-independent properties."""
    _, _, _, gd = oracle_lattice(8192, nscales=1)
    g = gd[0]
    H, off, bary = g['pc1_hash_cnt'], g['pc1_lattice_offset'], g['pc1_barycentric']
    N = 8192
    ct = ops.CloudTables(dev(bary), dev(off), H)
    rng = np.random.RandomState(0)
    x = dev(rng.randn(N, 68).astype(np.float32))
    y = dev(rng.randn(N, 68).astype(np.float32))
    Sx, Sy = ops.splat_raw(x, ct.csr(), H), ops.splat_raw(y, ct.csr(), H)
    Sxy = ops.splat_raw(x + 2 * y, ct.csr(), H)
    assert float((Sxy - (Sx + 2 * Sy)).abs().max()) < 1e-4                       # linearity
    ones = torch.ones(N, 4, device=DEV)
    s1 = ops.splat_raw(ones, ct.csr(), H, use_norm=False)                       # total mass = sum of weights
    assert abs(float(s1[:, 0].sum()) - float(bary.astype(np.float64).sum())) < 1e-2
    sl = ops.slice_raw(torch.ones(H, 4, device=DEV), ct.bary, ct.off, N)        # partition of unity
    assert float((sl - 1).abs().max()) < 1e-5
    # <splat x, z> == <x, slice_norm z>  (splat and its backward are adjoint)
    z = dev(rng.randn(H, 68).astype(np.float32))
    lhs = float((Sx.double() * z.double()).sum())
    rhs = float((x.double() * ops.slice_raw(z, ct.bary, ct.off, N, vscale=ct.csr()[3]).double()).sum())
    assert abs(lhs - rhs) < 1e-3 * max(1.0, abs(lhs))
    assert torch.equal(Sx, ops.splat_raw(x, ct.csr(), H))                       # deterministic


def test_tile_index_tables_and_gconv_with_them(ops):
    """hpl_tile_index: per-tile gather indices / tap masks of a row-ordered launch == what the kernel's prologue
    computes itself; a launch that is handed the tables gives bit-identical results (single order and tap groups)."""
    from hplflownet_amd.bcl import NbrTable
    rng = np.random.RandomState(11)
    M, F, C, O = 20000, 15, 96, 160
    nbr_np = rng.randint(0, M, (F, M)).astype(np.int32)
    nbr_np[rng.rand(F, M) < 0.55] = -1
    nbr_np[0] = np.arange(M)
    nbr = torch.from_numpy(nbr_np).to(DEV)
    tbl = NbrTable(nbr)
    tbl.vertices_per_point = 3.0
    perm = tbl.perm
    idx, mask = ops.tile_index(nbr, perm)
    tiles = (M + 63) // 64
    assert tuple(idx.shape) == (tiles, F, 64) and tuple(mask.shape) == (tiles, 8)
    pm = torch.full((tiles * 64,), -1, dtype=torch.long, device=DEV)
    pm[:M] = perm.long()
    want = torch.where(pm[None, :] >= 0, nbr[:, pm.clamp(min=0)], torch.full_like(nbr[:, :1], -1)).view(F, tiles, 64)
    assert torch.equal(idx, want.permute(1, 0, 2).contiguous())
    present = (want >= 0)                                           # (F, tiles, 64)
    bits = (2 ** torch.arange(F, device=DEV))[:, None]
    assert torch.equal(mask[:, 0].long(), (present.any(2).long() * bits).sum(0))
    assert torch.equal(mask[:, 3].long(), (present[:, :, 32:].any(2).long() * bits).sum(0))
    order = mask[:, 6].long()                                        # schedule: a permutation of the tiles, most taps first
    assert torch.equal(torch.sort(order)[0], torch.arange(tiles, device=DEV))
    pop = present.any(2).long().sum(0)[order]
    assert bool((pop[:-1] >= pop[1:]).all())
    A = torch.from_numpy(rng.randn(M, C).astype(np.float32)).to(DEV)
    W = torch.from_numpy((rng.randn(O, C, F) / np.sqrt(C * F)).astype(np.float32)).to(DEV)
    Wt = ops.weight_relayout(W, C, O, F, F, C * F, 1)
    bias = torch.from_numpy(rng.randn(O).astype(np.float32)).to(DEV)
    y0 = ops.gconv_raw(A, nbr, M, C, F, Wt, O, bias=bias, act=ops.ACT_LEAKY, row_perm=perm)
    y1 = ops.gconv_raw(A, nbr, M, C, F, Wt, O, bias=bias, act=ops.ACT_LEAKY, row_perm=perm, tiles=(idx, mask))
    assert torch.equal(y0, y1)
    groups, gt = tbl.groups(), tbl.group_tiles()
    assert groups is not None and len(gt) == len(groups)
    z0 = ops.gconv_passes(A, nbr, M, C, F, Wt, O, groups, bias=bias, act=ops.ACT_LEAKY)
    z1 = ops.gconv_passes(A, nbr, M, C, F, Wt, O, groups, bias=bias, act=ops.ACT_LEAKY, tiles=gt)
    assert torch.equal(z0, z1)
    # N <= 32 picks 128-row tiles: the 64-row tables are ignored, not misread
    Wt8 = ops.weight_relayout(W[:24].contiguous(), C, 24, F, F, C * F, 1)
    assert torch.equal(ops.gconv_raw(A, nbr, M, C, F, Wt8, 24, row_perm=perm),
                       ops.gconv_raw(A, nbr, M, C, F, Wt8, 24, row_perm=perm, tiles=(idx, mask)))


@pytest.mark.gpu
@pytest.mark.parametrize('M,C,O,F,density', [(53, 64, 32, 15, 0.5), (426, 260, 128, 15, 0.4), (1787, 64, 64, 1, 1.0),
                                              (4324, 580, 1024, 15, 0.95), (9433, 388, 256, 15, 0.45)])
def test_split_k_row_order_independent_and_second_destination(ops, M, C, O, F, density):
    """Split-K launches (small M, and mid-size launches split for load balance) cut the contraction at slice indices:
    the result is bit-identical with and without a row order (whose tiles keep different slice lists), equal to the
    unsplit launch within fp32 reassociation; rows < rows2 of the result also land in a second matrix (a column view
    of a wider buffer) on every path."""
    rng = np.random.RandomState(M)
    if F > 1:
        nbr_np = rng.randint(0, M, (F, M)).astype(np.int32)
        nbr_np[rng.rand(F, M) > density] = -1
        nbr_np[0] = np.arange(M)
        nbr = torch.from_numpy(nbr_np).to(DEV)
        perm = ops.tap_order(nbr)
        tiles = ops.tile_index(nbr, perm)
    else:
        nbr = perm = tiles = None
    A = torch.from_numpy(rng.randn(M, C).astype(np.float32)).to(DEV)
    W = torch.from_numpy((rng.randn(O, C, F) / np.sqrt(C * F)).astype(np.float32)).to(DEV)
    Wt = ops.weight_relayout(W, C, O, F, F, C * F, 1)
    bias = torch.from_numpy(rng.randn(O).astype(np.float32)).to(DEV)
    res = torch.from_numpy(rng.randn(M, O).astype(np.float32)).to(DEV)
    rows2 = M // 2 + 1
    kw = dict(bias=bias, act=ops.ACT_LEAKY, res=res)
    outs, seconds = [], []
    for order in ((perm, tiles), (None, None)):
        wide = torch.full((M, O + 8), 7.0, device=DEV)
        outs.append(ops.gconv_raw(A, nbr, M, C, F, Wt, O, row_perm=order[0], tiles=order[1], out2=wide[:, 4:4 + O],
                                  rows2=rows2, **kw))
        seconds.append(wide)
    for o, w in zip(outs, seconds):
        assert torch.equal(o, outs[0])                           # the row order is irrelevant
        assert torch.equal(w[:rows2, 4:4 + O], o[:rows2])
        assert bool((w[rows2:] == 7.0).all()) and bool((w[:, :4] == 7.0).all()) and bool((w[:, 4 + O:] == 7.0).all())
    wide = torch.full((M, O + 8), 7.0, device=DEV)
    unsplit = ops.gconv_raw(A, nbr, M, C, F, Wt, O, split_k=False, out2=wide[:, 4:4 + O], rows2=rows2, **kw)
    assert torch.equal(wide[:rows2, 4:4 + O], unsplit[:rows2]) and bool((wide[rows2:] == 7.0).all())
    assert rel_err(outs[0].cpu().numpy(), unsplit.cpu().numpy()) < 1e-5        # fp32 reassociation over K = F*C
    naive = torch.full((M, O + 8), 7.0, device=DEV)
    y_naive = ops.gconv_raw(A, nbr, M, C, F, Wt, O, naive=True, out2=naive[:, 4:4 + O], rows2=rows2, **kw)
    assert torch.equal(naive[:rows2, 4:4 + O], y_naive[:rows2])
    assert rel_err(outs[0].cpu().numpy(), y_naive.cpu().numpy()) < 1e-5


def test_fast_and_generic_epilogues_are_bit_identical():
    """hpl_gconv_forward's epilogue has a fast form (32-bit buffer addressing, residual loads batched per block) and the
    generic form it falls back to for operands of 2 GB and more; HPL_GCONV_EPILOGUE=0 forces the generic one.  Same bits on:
    a wrapped residual (the correlation layer: row m adds residual row m % res_mod), a plain residual + second destination,
    a split-K launch (partial tiles), with bias and LeakyReLU, ragged M / N."""
    import os
    import subprocess
    import sys
    import tempfile
    code = (
        "import sys, torch\n"
        "sys.path.insert(0, %r)\n"
        "from hplflownet_amd import ops\n"
        "torch.manual_seed(11)\n"
        "outs = []\n"
        "for M, rows, C, F, N, res_mod, rows2 in ((15 * 613, 700, 64, 15, 32, 613, 0), (5003, 5003, 68, 15, 64, 5003, 3001),\n"
        "                                         (301, 400, 260, 15, 128, 0, 0), (20000, 20000, 32, 1, 40, 0, 0)):\n"
        "    g = torch.Generator(device='cpu').manual_seed(M)\n"
        "    A = torch.randn(rows, C, device='cuda')\n"
        "    Wt = torch.zeros(ops.round_up(F * C, 32), ops.round_up(N, 4), device='cuda')\n"
        "    Wt[:F * C, :N] = torch.randn(F * C, N, device='cuda') / (F * C) ** 0.5\n"
        "    nbr = None\n"
        "    if F > 1:\n"
        "        nbr = torch.randint(0, rows, (F, M), generator=g, dtype=torch.int32)\n"
        "        nbr[torch.rand((F, M), generator=g) > 0.6] = -1\n"
        "        nbr = nbr.cuda()\n"
        "    res = torch.randn(res_mod, N, device='cuda') if res_mod else None\n"
        "    out2 = torch.zeros(rows2, N, device='cuda') if rows2 else None\n"
        "    y = ops.gconv_raw(A, nbr, M, C, F, Wt, N, bias=torch.randn(N, device='cuda'), act=ops.ACT_LEAKY, res=res,\n"
        "                      res_mod=res_mod, out2=out2, rows2=rows2)\n"
        "    outs.append(y.cpu())\n"
        "    if out2 is not None:\n"
        "        outs.append(out2.cpu())\n"
        "torch.save(outs, sys.argv[1])\n" % os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    got = []
    with tempfile.TemporaryDirectory() as d:
        for mode in ('1', '0'):
            f = os.path.join(d, 'y%s.pt' % mode)
            r = subprocess.run([sys.executable, '-c', code, f], env=dict(os.environ, HPL_GCONV_EPILOGUE=mode), capture_output=True,
                               text=True, timeout=600)
            assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
            got.append(torch.load(f))
    assert len(got[0]) == len(got[1]) == 5
    for a, b in zip(*got):
        assert torch.isfinite(a).all() and torch.equal(a, b)


@pytest.mark.parametrize('kind', ['gconv_f32_stencil', 'gconv_f32_dense_splitk', 'gconv3_groups', 'gconv3_dense', 'gconv3_groups_bf16x3',
                                  'gconv3_dense_bf16x3', 'gconv3_stencil_splitk', 'gconv3_stencil_splitk_bf16x3', 'wgrad_f32', 'splat', 'slice'])
def test_results_do_not_depend_on_what_runs_beside_them(kind):
    """Every kernel with hand-counted load waits (buffer loads the compiler's s_waitcnt bookkeeping does not see) on a side stream
    while memory-bound kernels of another stream saturate HBM: the same result as alone.  (This disturbance exposed the
    split-operand weight gradient in round 5, tests/test_gpu_wgrad3.py::test_concurrent_streams; the others passed it.)"""
    import torch
    from hplflownet_amd import ops
    g = torch.Generator().manual_seed(2)
    dev = 'cuda'
    # round 6 (ADVICE): both operand forms of the split kernel (fp16 pairs / bf16 triples: the LDS stages and products differ, the
    # hand-counted waits do not), its split-K form (mid-size stencil, partial tiles + finish), and a disturbance that is not only
    # elementwise reads: copies that saturate HBM in both directions and a stream of same-address atomics beside it
    planes = 3 if kind.endswith('_bf16x3') else None
    kind = kind.replace('_bf16x3', '')
    if kind in ('gconv_f32_stencil', 'wgrad_f32'):
        M, C, N, F = 70000, 68, 64, 15
    elif kind == 'gconv_f32_dense_splitk':
        M, C, N, F = 1787, 260, 128, 1
    elif kind == 'gconv3_groups':
        M, C, N, F = 26000, 580, 1024, 8
    elif kind == 'gconv3_dense':
        M, C, N, F = 8192, 1024, 1024, 1
    elif kind == 'gconv3_stencil_splitk':
        M, C, N, F = 9433, 388, 256, 15               # bcn3_: 74 row tiles x 1 column tile, three workgroups per tile over K
    else:
        M, C, N, F = 52000, 68, 68, 4
    A = torch.randn(M, C, generator=g).to(dev)
    nbr = None
    if F > 1:
        nbr = torch.randint(0, M, (F, M), generator=g).int()
        nbr[torch.rand(F, M, generator=g) < 0.4] = -1
        nbr = nbr.to(dev)
    if kind.startswith('gconv'):
        W = (torch.randn(N, C, F, generator=g) * 0.03).to(dev)
        Wt = ops.weight_relayout(W, C, N, F, F, C * F, 1)
        kw = {}
        if kind.startswith('gconv3'):
            kw['Wt3'] = ops.weight_split3(Wt, planes=planes) if planes else ops.weight_split3(Wt)
            if F > 1:
                perm = ops.tap_order(nbr)
                kw.update(row_perm=perm, tiles=ops.tile_index(nbr, perm, BM=128))
        fn = lambda: ops.gconv_raw(A, nbr, M, C, F, Wt, N, **kw)
    elif kind == 'wgrad_f32':
        dY = torch.randn(M, N, generator=g).to(dev)
        fn = lambda: ops.wgrad_raw(A, nbr, M, C, F, dY, N)
    else:
        H = 17000
        off = torch.randint(0, H, (4, M), generator=g).int().to(dev)
        bary = torch.rand(4, M, generator=g).to(dev)
        cl = ops.CloudTables(bary, off, H)
        Y = torch.randn(H, C, generator=g).to(dev)
        fn = (lambda: ops.splat_raw(A, cl.csr(), H, True)) if kind == 'splat' else (lambda: ops.slice_raw(Y, bary, off, M))
    ref = fn().clone()
    torch.cuda.synchronize()
    X = torch.randn(30000, 1024, device=dev)
    X2 = torch.empty_like(X)
    cnt = torch.zeros(64, device=dev)
    hot = torch.zeros(1 << 20, dtype=torch.int64, device=dev)          # index_add_ onto 64 words: a stream of contended atomics
    ones = torch.ones(1 << 20, device=dev)
    side = torch.cuda.Stream(priority=-1)
    third = torch.cuda.Stream()
    for it in range(8):
        ev = torch.cuda.Event()
        ev.record()
        with torch.cuda.stream(side):
            side.wait_event(ev)
            out = fn()
        with torch.cuda.stream(third):
            third.wait_event(ev)
            for _ in range(3):
                X2.copy_(X)
                cnt.index_add_(0, hot, ones)
        for _ in range(4):
            Z = torch.relu(X) + 1
        torch.cuda.synchronize()
        if kind == 'wgrad_f32':
            assert float((out - ref).abs().max()) < 1e-5 * float(ref.abs().max())
        else:
            assert torch.equal(out, ref)


@pytest.mark.parametrize('C,H', [(68, 97), (68, 700), (64, 300), (5, 50), (33, 211), (12, 1), (1024, 64)])
def test_splat_long_segments(ops, C, H):
    """Vertices with many contributors (csr_ptr[H] >= 6 H: the coarse levels) take the lane-group form of k_splat -- several
    lane groups per vertex, partial sums added in group order: same sums as the float64 restatement of
    models/bilateralNN.py:151-186, deterministic, and the accumulating variant adds the same numbers."""
    rng = np.random.RandomState(C * 1000 + H)
    N = 4096
    off = rng.randint(0, H, size=(4, N)).astype(np.int64)
    off[:, :8] = (H - 1)                                   # one vertex with a long list of its own
    bary = rng.rand(4, N).astype(np.float32)
    feat = rng.randn(N, C).astype(np.float32)
    ct = ops.CloudTables(dev(bary), dev(off), H)
    assert int(ct.csr()[0][-1]) == 4 * N >= 6 * H
    for use_norm in (True, False):
        S = ops.splat_raw(dev(feat), ct.csr(), H, use_norm)
        want = np.zeros((H, C))
        np.add.at(want, off.ravel(), (bary.ravel()[:, None].astype(np.float64) * np.tile(feat.astype(np.float64), (4, 1))))
        if use_norm:
            w = np.zeros(H)
            np.add.at(w, off.ravel(), bary.ravel().astype(np.float64))
            assert rel_err(ct.csr()[3].cpu().numpy(), 1.0 / (w + 1e-5)) < 1e-5
            want = want * ct.csr()[3].cpu().numpy()[:, None].astype(np.float64)
        assert rel_err(S.cpu().numpy(), want) < 1e-5, (C, H, use_norm)
        assert torch.equal(S, ops.splat_raw(dev(feat), ct.csr(), H, use_norm))
    from hplflownet_amd import _lib
    cp, cpt, cw, cn = ct.csr()
    acc = torch.ones(H, C, device='cuda')
    x = dev(feat)
    _lib.check(_lib.load().hpl_splat_add(_lib.ptr(x), C, C, _lib.ptr(cp), _lib.ptr(cpt), _lib.ptr(cw), _lib.ptr(cn), H, _lib.ptr(acc), C,
                                         _lib.stream()), 'hpl_splat_add')
    assert torch.equal(acc, 1.0 + ops.splat_raw(x, ct.csr(), H, True))
