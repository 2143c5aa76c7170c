"""GPU: autograd of the HIP ops against a float64 torch restatement of the same op
(torch.index_select / einsum / index_add on the device), random shapes and tables."""
import os

import numpy as np
import pytest
import torch

from common import oracle_lattice, rel_err

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def ref_gconv(A, W, bias, nbr, M, c0, C, F, act, res, res_mod, slope=0.1):
    """float64 torch reference.  W (O, Ctot, F)."""
    O = W.shape[0]
    Ap = torch.cat([A[:, :C].double(), torch.zeros(1, C, dtype=torch.float64, device=A.device)], 0)
    if nbr is None:
        X = Ap[:M][None]
    else:
        idx = nbr.long()
        idx = torch.where(idx < 0, torch.full_like(idx, A.shape[0]), idx)
        X = Ap[idx]                                                    # (F, M, C)
    Wd = W.double().view(O, -1, F)[:, c0:c0 + C, :]
    y = torch.einsum('fmc,ocf->mo', X, Wd)
    if bias is not None:
        y = y + bias.double()[None]
    if res is not None:
        y = y + res.double()[torch.arange(M, device=A.device) % res_mod]
    if act:
        y = torch.where(y > 0, y, slope * y)
    return y


CASES = [
    # M, rows, Ctot, c0, C, F, O, mode, bias, res, act
    (256, 256, 1024, 0, 1024, 1, 512, 'dense', True, False, True),
    (256, 256, 512, 0, 512, 1, 3, 'dense', True, False, False),
    (300, 300, 3, 0, 3, 1, 32, 'dense', True, False, True),
    (500, 500, 68, 0, 68, 15, 64, 'scatter', True, False, True),
    (500, 420, 192, 64, 64, 15, 32, 'scatter', False, False, False),
    (15 * 60, 77, 192, 128, 64, 15, 32, 'scatter', True, True, True),
    (60, 15 * 60, 32, 0, 32, 15, 64, 'scatter', True, False, True),
]


@pytest.mark.parametrize('case', CASES, ids=[str(i) for i in range(len(CASES))])
def test_gconv_autograd(case):
    from hplflownet_amd import ops
    M, rows, Ctot, c0, C, F, O, mode, has_bias, has_res, act = case
    g = torch.Generator(device='cpu').manual_seed(M + O)
    A = torch.randn(rows, C, generator=g).to(DEV).requires_grad_(True)
    W = (torch.randn(O, Ctot, F, generator=g) / np.sqrt(C * F) * 3).to(DEV).requires_grad_(True)
    bias = torch.randn(O, generator=g).to(DEV).requires_grad_(True) if has_bias else None
    nbr = None
    if F > 1:
        nbr = torch.randint(-1, rows, (F, M), generator=g).to(torch.int32).to(DEV)
    res_mod = 60 if has_res else 0
    res = torch.randn(res_mod, O, generator=g).to(DEV).requires_grad_(True) if has_res else None
    go = torch.randn(M, O, generator=g).to(DEV)
    y = ops.gconv(A, W, bias, nbr, M, F, act=1 if act else 0, c0=c0, C=C, res=res, res_mod=res_mod, bwd_mode=mode)
    (y * go).sum().backward()
    got = [A.grad.clone(), W.grad.clone(), bias.grad.clone() if has_bias else None,
           res.grad.clone() if has_res else None]
    for t in (A, W, bias, res):
        if t is not None:
            t.grad = None
    yr = ref_gconv(A, W, bias, nbr, M, c0, C, F, act, res, res_mod)
    (yr * go.double()).sum().backward()
    assert rel_err(y.detach().cpu().numpy(), yr.detach().cpu().numpy()) < 1e-5
    want = [A.grad, W.grad, bias.grad if has_bias else None, res.grad if has_res else None]
    for nm, a, b in zip(('dA', 'dW', 'db', 'dres'), got, want):
        if b is not None:
            assert rel_err(a.cpu().numpy(), b.cpu().numpy()) < 2e-5, nm


def test_gconv_autograd_mirror():
    """symmetric table (a real blur table): mirror-gather backward == scatter backward == torch."""
    from hplflownet_amd import ops
    from hplflownet_amd.bcl import NbrTable
    _, _, _, gd = oracle_lattice(256)
    nbr = torch.from_numpy(gd[2]['pc1_blur_neighbors']).to(torch.int32).to(DEV)
    H = nbr.shape[1]
    tbl = NbrTable(nbr)
    assert tbl.symmetric and tbl.bwd_mode(H) == 'mirror'
    g = torch.Generator(device='cpu').manual_seed(1)
    C, O, F = 36, 48, 15
    W = (torch.randn(O, C, F, generator=g) / 10).to(DEV).requires_grad_(True)
    go = torch.randn(H, O, generator=g).to(DEV)
    grads = {}
    for mode in ('mirror', 'scatter'):
        A = torch.randn(H, C, generator=torch.Generator().manual_seed(2)).to(DEV).requires_grad_(True)
        y = ops.gconv(A, W, None, nbr, H, F, act=1, bwd_mode=mode)
        (y * go).sum().backward()
        grads[mode] = A.grad.clone()
        W.grad = None
    A = torch.randn(H, C, generator=torch.Generator().manual_seed(2)).to(DEV).requires_grad_(True)
    yr = ref_gconv(A, W, None, nbr, H, 0, C, F, True, None, 0)
    (yr * go.double()).sum().backward()
    assert rel_err(grads['mirror'].cpu().numpy(), A.grad.cpu().numpy()) < 2e-5
    assert rel_err(grads['scatter'].cpu().numpy(), A.grad.cpu().numpy()) < 2e-5
    bad = nbr.clone()
    bad[3, 5] = (bad[3, 5] + 1) % H
    assert not NbrTable(bad).symmetric


def test_splat_slice_autograd():
    from hplflownet_amd import ops
    _, _, _, gd = oracle_lattice(256)
    g = gd[1]
    H, off, bary = g['pc1_hash_cnt'], g['pc1_lattice_offset'], g['pc1_barycentric']
    N = off.shape[1]
    ct = ops.CloudTables(torch.from_numpy(bary).to(DEV), torch.from_numpy(off).to(DEV), H)
    gen = torch.Generator().manual_seed(0)
    x = torch.randn(N, 20, generator=gen).to(DEV).requires_grad_(True)
    bias = torch.randn(20, generator=gen).to(DEV).requires_grad_(True)
    go = torch.randn(N, 20, generator=gen).to(DEV)
    s = ops.SplatFn.apply(x, ct, True)
    out = ops.SliceFn.apply(s, ct, bias)
    (out * go).sum().backward()
    gx, gb = x.grad.clone(), bias.grad.clone()
    x.grad = bias.grad = None
    offl = torch.from_numpy(off).to(DEV).long()
    b = torch.from_numpy(bary).to(DEV).double()
    S = torch.zeros(H, 20, dtype=torch.float64, device=DEV)
    w = torch.zeros(H, dtype=torch.float64, device=DEV)
    for r in range(4):
        S = S.index_add(0, offl[r], b[r][:, None] * x.double())
        w = w.index_add(0, offl[r], b[r])
    S = S / (w + 1e-5)[:, None]
    o = sum(b[r][:, None] * S[offl[r]] for r in range(4)) + bias.double()[None]
    (o * go.double()).sum().backward()
    assert rel_err(out.detach().cpu().numpy(), o.detach().cpu().numpy()) < 1e-5
    assert rel_err(gx.cpu().numpy(), x.grad.cpu().numpy()) < 2e-5
    assert rel_err(gb.cpu().numpy(), bias.grad.cpu().numpy()) < 2e-5


def test_tap_order_and_row_perm_leave_results_unchanged():
    """hpl_tap_order groups rows by tap mask; gconv with row_perm is bit-identical to without."""
    from hplflownet_amd import ops
    _, _, _, gd = oracle_lattice(1024)
    nbr_np = gd[0]['pc1_blur_neighbors'].astype(np.int32)
    nbr = torch.from_numpy(nbr_np).to(DEV)
    F, H = nbr.shape
    perm = ops.tap_order(nbr)
    p = perm.cpu().numpy()
    assert np.array_equal(np.sort(p), np.arange(H))                       # a permutation
    mask = ((nbr_np >= 0) * (1 << np.arange(F))[:, None]).sum(0)
    rank = mask.copy()                                                   # position of a mask in the Gray-code sequence
    for sh in (1, 2, 4, 8):
        rank ^= rank >> sh
    key = rank[p].astype(np.int64) * H + p                                # groups in Gray order, ascending row id inside: deterministic
    assert np.all(np.diff(key) > 0)
    assert np.array_equal(ops.tap_order(nbr).cpu().numpy(), p)           # same bits on every call
    g = torch.Generator().manual_seed(4)
    for C, O in ((68, 64), (132, 128)):
        A = torch.randn(H, C, generator=g).to(DEV)
        W = (torch.randn(O, C, F, generator=g) / 30).to(DEV)
        bias = torch.randn(O, generator=g).to(DEV)
        Wt = ops.weight_relayout(W, C, O, F, F, C * F, 1)
        y0 = ops.gconv_raw(A, nbr, H, C, F, Wt, O, bias=bias, act=1, split_k=False)
        y1 = ops.gconv_raw(A, nbr, H, C, F, Wt, O, bias=bias, act=1, row_perm=perm, split_k=False)
        yn = ops.gconv_raw(A, nbr, H, C, F, Wt, O, bias=bias, act=1, naive=True)
        assert torch.equal(y0, y1)
        assert rel_err(y1.cpu().numpy(), yn.cpu().numpy()) < 1e-6
    # a table with whole taps missing: skipped slices must not change anything
    sparse = nbr.clone()
    sparse[[2, 5, 9, 14]] = -1
    A = torch.randn(H, 68, generator=g).to(DEV)
    W = (torch.randn(64, 68, F, generator=g) / 30).to(DEV)
    Wt = ops.weight_relayout(W, 68, 64, F, F, 68 * F, 1)
    ys = ops.gconv_raw(A, sparse, H, 68, F, Wt, 64, row_perm=ops.tap_order(sparse))
    yn = ops.gconv_raw(A, sparse, H, 68, F, Wt, 64, naive=True)
    assert rel_err(ys.cpu().numpy(), yn.cpu().numpy()) < 1e-6
    empty = torch.full_like(nbr, -1)
    ye = ops.gconv_raw(A, empty, H, 68, F, Wt, 64, bias=torch.ones(64, device=DEV))
    assert float((ye - 1).abs().max()) == 0.0


def test_engine_train_validate_resume(tmp_path):
    """Driver row f1: a few optimiser steps on two tiny pairs lower the training loss, the lattice
    pipeline (side stream) feeds both loops, and a checkpoint round trip reproduces the metrics."""
    from hplflownet_amd import engine
    tr = engine.Trainer('HPLFlowNetShallow', DEV, lr=1e-3)
    data = engine.SyntheticPairs(2, 512, DEV)
    first = tr.train_epoch(data)
    for _ in range(5):
        last = tr.train_epoch(data)
    assert np.isfinite(first) and np.isfinite(last) and last < first, (first, last)
    m1 = tr.validate(data)
    assert set(m1) == {'EPE3D', 'Acc3DS', 'Acc3DR', 'Outliers'} and m1['EPE3D'] < first
    tr.min_loss = m1['EPE3D']
    path = tr.save_checkpoint(str(tmp_path), is_best=True)
    tr2 = engine.Trainer('HPLFlowNetShallow', DEV, lr=1e-3)
    tr2.resume(path)
    assert tr2.epoch == tr.epoch == 6
    m2 = tr2.validate(data)
    assert all(abs(m1[k] - m2[k]) < 1e-6 for k in m1), (m1, m2)
    # the optimiser state travelled too: the next step is identical
    a, b = tr.train_epoch(data), tr2.train_epoch(data)
    assert abs(a - b) < 1e-5 * max(1.0, abs(a)), (a, b)


def test_engine_real_data_layout(tmp_path):
    """Rows f2-f4 end to end: a directory tree laid out like FlyingThings3D_subset_processed_35m goes through
    the reader (axis flips), Augmentation / ProcessData, the on-device lattice pipeline and two epochs of
    training with shuffling; then evaluation of the best checkpoint through the CLI entry."""
    from hplflownet_amd import engine
    from hplflownet_amd.synthetic import synthetic_pair
    for split, count in (('train', 8), ('val', 4)):
        for i in range(count):
            d = tmp_path / 'FlyingThings3D_subset_processed_35m' / split / ('%07d' % i)
            d.mkdir(parents=True)
            pc1, pc2, _ = synthetic_pair(700, 50 + i)
            flip = np.array([-1, 1, -1], np.float32)                  # stored with x and z negated
            np.save(str(d / 'pc1.npy'), pc1 * flip)
            np.save(str(d / 'pc2.npy'), pc2 * flip)
    ck = tmp_path / 'ck'
    logs = []
    import builtins
    real_print = builtins.print
    builtins.print = lambda *a, **k: logs.append(' '.join(str(x) for x in a))
    try:
        best = engine.main(['--arch', 'HPLFlowNetShallow', '--points', '512', '--pairs', '2', '--val-pairs', '1',
                            '--epochs', '2', '--dataset', 'FlyingThings3DSubset', '--data-root', str(tmp_path),
                            '--init', 'xavier', '--ckpt-dir', str(ck)])
        res = engine.main(['--arch', 'HPLFlowNetShallow', '--points', '512', '--pairs', '1', '--evaluate',
                           '--dataset', 'FlyingThings3DSubset', '--data-root', str(tmp_path),
                           '--resume', str(ck / 'model_best.pth.tar')])
    finally:
        builtins.print = real_print
    assert np.isfinite(best) and sorted(os.listdir(str(ck))) == ['checkpoint.pth.tar', 'checkpoint_1.pth.tar',
                                                                  'model_best.pth.tar']
    assert any('published split has 19640' in ln for ln in logs)        # count mismatch is reported, not fatal
    assert abs(res['EPE3D'] - best) < 1e-5, (res, best)


@pytest.mark.parametrize('G', [2, 3, 5])
def test_tap_group_passes_equal_single_pass(G):
    """Inference runs wide stencil convs as one pass per group of consecutive taps (own row order per
    group, passes accumulate into the output): same result as the single pass up to fp32 summation
    order, bias / residual / activation applied once."""
    from hplflownet_amd import ops
    from hplflownet_amd.bcl import NbrTable
    from hplflownet_amd.ops import ACT_LEAKY
    rng = np.random.RandomState(G)
    M, C, O, F = 20000, 260, 96, 15
    nbr = np.where(rng.rand(F, M) < 0.45, rng.randint(0, M, size=(F, M)), -1).astype(np.int32)
    nbr[0] = np.arange(M)
    tbl = NbrTable(torch.from_numpy(nbr).to(DEV))
    tbl.TAP_GROUPS = G
    A = torch.from_numpy(rng.randn(M, C).astype(np.float32)).to(DEV)
    W = torch.from_numpy((rng.randn(O, C, F, 1) / np.sqrt(C * F)).astype(np.float32)).to(DEV)
    b = torch.from_numpy(rng.randn(O).astype(np.float32)).to(DEV)
    res = torch.from_numpy(rng.randn(M, O).astype(np.float32)).to(DEV)
    groups = tbl.groups()
    assert len(groups) == G and groups[0][0] == 0 and groups[-1][1] == F
    for f0, f1, perm in groups:
        assert sorted(perm.cpu().tolist()) == list(range(M))
    with torch.no_grad():
        one = ops.gconv(A, W, b, tbl.t, M, F, act=ACT_LEAKY, res=res, res_mod=M, row_perm=tbl.perm)
        out = torch.empty(M, O + 8, device=DEV)[:, 4:4 + O]                  # strided destination
        many = ops.gconv(A, W, b, tbl.t, M, F, act=ACT_LEAKY, res=res, res_mod=M, row_perm=tbl.perm,
                         tap_groups=groups, out=out)
    assert many.data_ptr() == out.data_ptr()
    assert rel_err(many.cpu().numpy(), one.cpu().numpy()) < 2e-6
    assert float((many - one).abs().max()) < 1e-4


def test_weight_bank_matches_individual_relayouts():
    """Batched re-layout (one launch for all recorded images, forward and mirrored direction): identical
    images; stale entries (parameter changed since the refresh) fall back to the individual launch."""
    from hplflownet_amd import ops
    rng = np.random.RandomState(5)
    bank = ops.WeightBank()
    ws = [torch.from_numpy(rng.randn(o, c, f, 1).astype(np.float32)).to(DEV).requires_grad_(True)
          for o, c, f in ((64, 68, 15), (33, 7, 1), (128, 260, 15), (512, 1024, 1))]
    reqs = []
    for w in ws:
        O, C, F = w.shape[0], w.shape[1], w.shape[2]
        reqs.append((w, (C, O, F, F, C * F, 1), dict(base=0, mirror=False)))
        reqs.append((w, (O, C, F, C * F, F, 1), dict(base=0, mirror=(F > 1))))
    reqs.append((ws[0], (20, 64, 15, 15, 68 * 15, 1), dict(base=5 * 15, mirror=False)))       # channel sub-range
    first = [bank.get(w, *a, **k).clone() for w, a, k in reqs]            # recorded, served individually
    bank.refresh()
    for (w, a, k), ref in zip(reqs, first):
        got = bank.get(w, *a, **k)
        assert got.data_ptr() >= bank.buf.data_ptr() and torch.equal(got, ref)
    with torch.no_grad():
        ws[2].mul_(2.0)                                                    # version bump -> entry is stale
    w, a, k = reqs[4]
    stale = bank.get(w, *a, **k)
    assert not (bank.buf.data_ptr() <= stale.data_ptr() < bank.buf.data_ptr() + 4 * bank.total)
    assert torch.equal(stale, 2.0 * first[4])
    bank.refresh()
    assert torch.equal(bank.get(w, *a, **k), 2.0 * first[4])
