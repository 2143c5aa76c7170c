"""GPU: the small entry points the native training step and the per-tap patch correlation added (include/hpl_bcl.h), each against
a plain torch fp32 / fp64 statement of the same arithmetic: hpl_gather_sum + hpl_table_invert (models/bnn_flow.py:195-202 and its
gradient), hpl_psum, hpl_regroup, hpl_epe3d (models/epe3d_loss.py:9-10), hpl_splat_add / hpl_slice_add, the "taps as column blocks"
weight images and hpl_weight_unlayout_batch, and the workspace layout of the executor (a poisoned workspace must not reach a result)."""
import ctypes
import types

import numpy as np
import pytest
import torch

from hplflownet_amd import _lib, ops
from hplflownet_amd._lib import RelayoutJob, check, ptr, stream

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def _corr_table(K, F, H0, H1, gen, miss=0.3):
    """[K, F*H0] table whose (k, f) blocks are injective maps h -> v (as the lattice's pc2 correlation table)."""
    t = torch.full((K, F * H0), -1, dtype=torch.int32)
    for k in range(K):
        for f in range(F):
            v = torch.randperm(max(H0, H1), generator=gen)[:H0]
            v = torch.where((v < H1) & (torch.rand(H0, generator=gen) > miss), v, torch.full_like(v, -1))
            t[k, f * H0:(f + 1) * H0] = v.int()
    return t.to(DEV)


@pytest.mark.parametrize('H0,H1,N', [(37, 41, 32), (500, 480, 32), (130, 131, 8)])
def test_gather_sum_and_its_gradient_through_the_inverse_table(H0, H1, N):
    g = torch.Generator().manual_seed(3)
    K = F = 15
    tbl = _corr_table(K, F, H0, H1, g)
    M = F * H0
    Z = torch.randn(H1, K * N, generator=g).to(DEV)
    bias, res = torch.randn(N, generator=g).to(DEV), torch.randn(H0, N, generator=g).to(DEV)
    y = ops.gather_sum_raw(Z, tbl, M, K, N, N, bias=bias, res=res, res_mod=H0, act=1, slope=0.1)
    idx = tbl.long()
    ref = torch.zeros(M, N, dtype=torch.float64, device=DEV)
    for k in range(K):
        rows = Z[idx[k].clamp(min=0), k * N:(k + 1) * N].double()
        ref += torch.where((idx[k] >= 0)[:, None], rows, torch.zeros_like(rows))
    ref = ref + bias.double() + res.double().repeat(F, 1)
    ref = torch.where(ref > 0, ref, 0.1 * ref)
    assert float((y.double() - ref).abs().max()) < 1e-5 * float(ref.abs().max())
    # gradient w.r.t. Z: dZ[v, k*N + n] = sum over the rows m with tbl[k][m] == v of G[m, n]
    G = torch.randn(M, N, generator=g).to(DEV)
    inv = ops.table_invert(tbl, H0, F, H1)
    dZ = ops.gather_sum_raw(G, inv, K * H1, F, N, 0).view(H1, K * N)
    want = torch.zeros(H1, K * N, dtype=torch.float64, device=DEV)
    for k in range(K):
        ok = idx[k] >= 0
        want[:, k * N:(k + 1) * N].index_add_(0, idx[k][ok], G[ok].double())
    assert float((dZ.double() - want).abs().max()) < 1e-5 * max(1.0, float(want.abs().max()))
    assert torch.equal(dZ, ops.gather_sum_raw(G, inv, K * H1, F, N, 0).view(H1, K * N))            # no atomics: bit for bit


def test_psum_regroup_epe3d():
    L = _lib.load()
    g = torch.Generator().manual_seed(5)
    mod, per, N = 77, 15, 32
    X = torch.randn(per * mod, N + 4, generator=g).to(DEV)[:, :N]
    out = torch.full((mod, N), 3.0, device=DEV)
    check(L.hpl_psum(ptr(X), X.stride(0), per * mod, mod, N, ptr(out), N, 0, stream()), 'hpl_psum')
    want = X.view(per, mod, N).double().sum(0)
    assert float((out.double() - want).abs().max()) < 1e-5
    check(L.hpl_psum(ptr(X), X.stride(0), per * mod, mod, N, ptr(out), N, 1, stream()), 'hpl_psum')
    assert float((out.double() - 2 * want).abs().max()) < 2e-5
    M, F, C = 91, 15, 32
    Gm = torch.randn(M, F * C, generator=g).to(DEV)
    dst = torch.zeros(F * M, C + 4, device=DEV)
    check(L.hpl_regroup(ptr(Gm), F * C, M, F, C, ptr(dst), C + 4, 0, stream()), 'hpl_regroup')
    assert torch.equal(dst[:, :C], Gm.view(M, F, C).transpose(0, 1).reshape(F * M, C)) and float(dst[:, C:].abs().max()) == 0.0
    n = 3001
    pred = torch.randn(n, 3, generator=g).to(DEV)
    sf = torch.randn(3, n, generator=g).to(DEV)
    pred[7] = sf[:, 7]                                   # a zero-length error vector: gradient 0, not nan
    grad, loss = torch.empty(n, 3, device=DEV), torch.zeros(1, device=DEV)
    check(L.hpl_epe3d(ptr(pred), ptr(sf), n, ptr(grad), ptr(loss), stream()), 'hpl_epe3d')
    p = pred.clone().requires_grad_(True)
    ref = torch.norm(p.t()[None] - sf[None], p=2, dim=1).mean()
    assert abs(float(loss) - float(ref)) < 1e-6 * float(ref)
    d = (pred - sf.t())
    want = d / (n * d.norm(dim=1, keepdim=True).clamp(min=1e-30))
    want[7] = 0
    assert float((grad - want).abs().max()) < 1e-9 + 1e-6 * float(want.abs().max()) and bool(torch.isfinite(grad).all())


def test_splat_add_and_slice_add_accumulate():
    g = torch.Generator().manual_seed(9)
    n, H, C = 600, 230, 68
    off = torch.randint(0, H, (4, n), generator=g).int().to(DEV)
    bary = torch.rand(4, n, generator=g).to(DEV)
    cl = ops.CloudTables(bary, off, H)
    feat = torch.randn(n, C, generator=g).to(DEV)
    base = torch.randn(H, C, generator=g).to(DEV)
    s = ops.splat_raw(feat, cl.csr(), H, use_norm=True)
    acc = base.clone()
    cp, cpt, cw, cn = cl.csr()
    check(_lib.load().hpl_splat_add(ptr(feat), C, C, ptr(cp), ptr(cpt), ptr(cw), ptr(cn), H, ptr(acc), C, stream()), 'hpl_splat_add')
    assert float((acc - (base + s)).abs().max()) < 1e-5
    Y = torch.randn(H, C, generator=g).to(DEV)
    z = ops.slice_raw(Y, bary, off, n)
    base2 = torch.randn(n, C, generator=g).to(DEV)
    acc2 = base2.clone()
    check(_lib.load().hpl_slice_add(ptr(Y), C, C, ptr(bary), ptr(off), n, None, None, ptr(acc2), C, stream()), 'hpl_slice_add')
    assert float((acc2 - (base2 + z)).abs().max()) < 1e-5


def test_cols_images_and_batched_unlayout_round_trip():
    """mirror == 2 images of hpl_weight_relayout_batch ([c][f*O + o] = W[o, c0 + c, f]) and hpl_weight_unlayout_batch (mirror 0 and 2)
    as the inverse of the batch re-layout: a gradient written in image layout comes back in the parameter's layout."""
    g = torch.Generator().manual_seed(11)
    O, Ctot, F, c0, C = 32, 192, 15, 128, 48
    W = torch.randn(O, Ctot, 1, F, 1, generator=g).to(DEV)
    W2 = torch.randn(64, 68, F, 1, generator=g).to(DEV)
    bank = ops.WeightBank()
    j_cols = bank.register(W, C, O, F, F, Ctot * F, 1, c0 * F, 2)
    j_fwd = bank.register(W2, 68, 64, F, F, 68 * F, 1, 0, 0)
    bank.refresh()
    img = bank.buf[j_cols[2]:j_cols[2] + j_cols[3]].view(j_cols[6], j_cols[7])
    want = W.view(O, Ctot, F)[:, c0:c0 + C, :].permute(1, 2, 0).reshape(C, F * O)
    assert torch.equal(img[:C, :F * O], want) and float(img[C:].abs().max()) == 0.0
    # un-layout both images into zeroed gradient tensors
    gW, gW2 = torch.zeros_like(W), torch.zeros_like(W2)
    arr = (RelayoutJob * 2)()
    for a, (w, gw, job, (R, Q, Ff, sr, sq, sf, base, mirror)) in zip(arr, ((W, gW, j_cols, j_cols[1]), (W2, gW2, j_fwd, j_fwd[1]))):
        a.W, a.base, a.sr, a.sq, a.sf, a.R, a.Q, a.F, a.mirror, a.ldw = gw.data_ptr(), base, sr, sq, sf, R, Q, Ff, mirror, job[7]
    src = torch.cat([bank.buf[j_cols[2]:j_cols[2] + j_cols[3]], bank.buf[j_fwd[2]:j_fwd[2] + j_fwd[3]]])
    jobs = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(DEV)
    prefix = torch.tensor([0, j_cols[3], j_cols[3] + j_fwd[3]], dtype=torch.int64, device=DEV)
    check(_lib.load().hpl_weight_unlayout_batch(ptr(jobs), 2, ptr(prefix), int(src.numel()), ptr(src), stream()), 'hpl_weight_unlayout_batch')
    assert torch.equal(gW.view(O, Ctot, F)[:, c0:c0 + C], W.view(O, Ctot, F)[:, c0:c0 + C]) and float(gW.view(O, Ctot, F)[:, :c0].abs().max()) == 0.0
    assert torch.equal(gW2, W2)


@pytest.mark.parametrize('fill', [float('nan'), 1e30])
def test_inference_plan_on_a_poisoned_workspace(fill):
    """The executor lays the activation matrices out by their lifetimes (matrices that are never alive together share memory): a
    forward must not read a cell before the program has written it -- NaN-filled workspaces (1e30-filled: a stale cell inside an
    operand-scale reduction would not make a NaN, it would silently shrink the scale), two different pairs in turn, same flows
    as the launch-by-launch Python path."""
    import hplflownet_amd as H
    from hplflownet_amd.synthetic import SCALES_FILTER_MAP, fill_module_, synthetic_pair
    a = types.SimpleNamespace(dim=3, scales_filter_map=SCALES_FILTER_MAP, evaluate=True, use_leaky=True, bcn_use_bias=True,
                              bcn_use_norm=True, last_relu=False, DEVICE='cuda')
    model = H.HPLFlowNet(a)
    fill_module_(model, 1.0, 'hash')
    model = model.to(DEV).eval()
    gen = H.GenerateDataUnsymmetric(a, device=DEV, wide_up=model.lattice_hint())
    plan = model.forward_plan()
    with torch.no_grad():
        for seed, n in ((0, 2048), (1, 1500), (0, 2048)):
            pc1, pc2, _ = synthetic_pair(n, seed)
            t1, t2 = [torch.from_numpy(np.ascontiguousarray(x.T)).to(DEV) for x in (pc1, pc2)]
            lat = gen.build(t1, t2)
            for ws in plan._ws.values():
                ws.view(torch.float32)[:ws.numel() // 4].fill_(fill)
            y = model(t1[None], t2[None], lat)
            model.native_forward = False
            ref = model(t1[None], t2[None], lat)
            model.native_forward = True
            assert bool(torch.isfinite(y).all()) and torch.equal(y, ref)
