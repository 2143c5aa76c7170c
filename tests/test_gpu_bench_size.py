"""GPU: parity at the sizes the benchmark runs (BASELINE configs 3, 4, 5), pinned to the reference.

* config 3 / 5 -- full HPLFlowNet inference at N=8192 (device-built lattice + HIP layers) against
  (a) vectors produced by the REFERENCE itself at this size (tests/golden/models_large.npz, fixture F8 of
  tools/make_fixtures.py: two frustum seeds and one surface-like pair) and (b) the numpy oracle on the same
  inputs; reference loop: /root/reference/evaluation_bnn.py:50-76.
* config 4 -- one training step of the full model (reference loop /root/reference/main.py:203-217: forward,
  EPE3DLoss mean, backward, Adam) at N=4096 against the reference's own loss / gradient norms (F8) and at
  N=8192 against the differentiable torch oracle (float64: the exact answer), every parameter gradient.
* lattice-rebuild determinism of the inference path, and the gradient all-reduce through RCCL (backend
  "nccl") in a process group of one rank -- hooks, bucket packing, asynchronous all-reduce on hardware.
"""
import os
import socket
import types

import numpy as np
import pytest
import torch

from common import GOLD, oracle_lattice
from hplflownet_amd.synthetic import SCALES_FILTER_MAP, fill_module_, subsample, surface_pair, synthetic_pair

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def model_args(evaluate=True):
    return types.SimpleNamespace(dim=3, scales_filter_map=SCALES_FILTER_MAP, evaluate=evaluate, use_leaky=True,
                                 bcn_use_bias=True, bcn_use_norm=True, last_relu=False, DEVICE='cuda')


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def full_model(evaluate=True):
    import hplflownet_amd as H
    m = H.HPLFlowNet(model_args(evaluate))
    fill_module_(m, 1.0, 'hash')
    sd = {k: v.numpy().copy() for k, v in m.state_dict().items()}
    return m.to(DEV), sd


PAIRS = {'full_n8192_s0': lambda: synthetic_pair(8192, 0), 'full_n8192_s1': lambda: synthetic_pair(8192, 1),
         'surf_n8192_s0': lambda: surface_pair(8192, 0)}


@pytest.mark.parametrize('tag', sorted(PAIRS))
def test_config3_full_n8192_vs_reference_and_oracle(tag):
    import hplflownet_amd as H
    from oracle import bcl_oracle as BO, lattice_oracle as LO
    z = np.load(os.path.join(GOLD, 'models_large.npz'))
    pc1, pc2, sf = PAIRS[tag]()
    m, sd = full_model()
    m.eval()
    gen = H.GenerateDataUnsymmetric(model_args(), device=DEV, wide_up=m.lattice_hint())
    t1, t2, tsf, lat = gen([pc1, pc2, sf])
    assert [lv.H[0] for lv in lat.levels] == z[tag + '_H1'].tolist()          # the reference's own vertex counts
    with torch.no_grad():
        y = m(t1[None], t2[None], lat)
    got = y[0].cpu().numpy()
    # (a) the reference itself, run at this size in the build container
    ref = z[tag + '_flow']
    scale = max(1.0, float(np.abs(ref).max()))
    epe = float(np.sqrt(((got - sf.T) ** 2).sum(0)).mean())
    assert abs(epe - float(z[tag + '_loss'])) < 1e-4                          # north-star bar: EPE3D delta < 1e-4
    assert np.abs(subsample(got) - ref).max() < 2e-4 * scale
    # (b) the CPU oracle on the same inputs: lattice bit-exact, every flow vector
    gd = LO.generate_data(pc1, pc2, SCALES_FILTER_MAP)
    for lv, d in zip(lat.levels, gd):
        assert np.array_equal(lv.clouds[0].off.cpu().numpy(), d['pc1_lattice_offset'])
        assert np.array_equal(lv.blur[1].t.cpu().numpy(), d['pc2_blur_neighbors'])
    flow_o = BO.hplflownet_forward(sd, pc1.T, pc2.T, gd)
    assert abs(epe - BO.epe3d(flow_o, sf.T)) < 1e-4
    assert np.abs(got - flow_o).max() < 2e-4 * scale


def _grads_close(got, want, norm_tol, entry_tol):
    """per-parameter: gradient norm within norm_tol (relative, floor 1e-3 of the largest norm) and every entry
    within entry_tol of the tensor's largest entry."""
    assert set(got) == set(want)
    scale = max(float(np.linalg.norm(v)) for v in want.values())
    worst_n = worst_e = (0.0, '')
    for k in want:
        a, b = got[k].astype(np.float64), np.asarray(want[k], np.float64)
        nb = float(np.linalg.norm(b))
        worst_n = max(worst_n, (abs(float(np.linalg.norm(a)) - nb) / max(nb, 1e-3 * scale), k))
        worst_e = max(worst_e, ((float(np.abs(a - b).max()) - 1e-7 * scale) / float(np.abs(b).max()), k))
    print('gradient parity: worst norm deviation %.3g (%s), worst entry deviation %.3g of the tensor maximum (%s)'
          % (worst_n + worst_e))
    for k in want:
        a, b = got[k].astype(np.float64), np.asarray(want[k], np.float64)
        nb = float(np.linalg.norm(b))
        assert abs(float(np.linalg.norm(a)) - nb) < norm_tol * max(nb, 1e-3 * scale), (k, float(np.linalg.norm(a)), nb)
        assert float(np.abs(a - b).max()) <= entry_tol * float(np.abs(b).max()) + 1e-7 * scale, k


def test_config4_train_step_n4096_vs_reference():
    """The reference's own train-mode forward + backward at N=4096 (fixture F8): loss and gradient norms."""
    import hplflownet_amd as H
    z = np.load(os.path.join(GOLD, 'models_large.npz'))
    tag = 'train_n4096_s0'
    pc1, pc2, sf = synthetic_pair(4096, 0)
    m, _ = full_model(evaluate=False)
    m.train()
    gen = H.GenerateDataUnsymmetric(model_args(False), device=DEV, wide_up=m.lattice_hint())
    t1, t2, tsf, lat = gen([pc1, pc2, sf])
    lat.prepare(for_training=True)
    flow = m(t1[None], t2[None], lat)
    loss = torch.norm(flow - tsf[None], p=2, dim=1).mean()                     # main.py:213
    loss.backward()
    assert abs(float(loss) - float(z[tag + '_loss'])) < 1e-4
    ref = z[tag + '_flow']
    assert np.abs(subsample(flow.detach()[0].cpu().numpy()) - ref).max() < 2e-4 * max(1.0, float(np.abs(ref).max()))
    names = bytes(z[tag + '_gradnames']).decode().split('\n')
    want = dict(zip(names, z[tag + '_gradnorm']))
    got = {k: float(p.grad.norm()) for k, p in m.named_parameters()}
    assert set(got) == set(want)
    scale = max(want.values())
    for k in want:
        assert abs(got[k] - want[k]) < 2e-3 * max(want[k], 1e-3 * scale), (k, got[k], want[k])


def test_config4_train_step_n8192_vs_oracle():
    """BASELINE config 4 at its size: one training step (forward, EPE3D loss, backward, Adam lr 1e-4) of the full
    model at N=8192 on a device-built lattice -- the tap-list weight gradient, the mirrored-gather data gradient
    and the tap-group passes at H = 25 841 / 34 631.  Every parameter gradient against the float64 torch oracle
    (pinned to the reference by tests/test_oracle_torch.py), then the optimiser update itself."""
    import hplflownet_amd as H
    from hplflownet_amd import ops
    from oracle import lattice_oracle as LO, torch_oracle as TO
    pc1, pc2, sf = synthetic_pair(8192, 0)
    m, sd = full_model(evaluate=False)
    m.train()
    bank = ops.BANK
    ops.enable_weight_bank(True)          # the training driver's configuration (engine.Trainer, bench.py --train)
    try:
        opt = torch.optim.Adam(m.parameters(), lr=1e-4, weight_decay=0)        # main.py:138-140
        gen = H.GenerateDataUnsymmetric(model_args(False), device=DEV, wide_up=m.lattice_hint())
        t1, t2, tsf, lat = gen([pc1, pc2, sf])
        lat.prepare(for_training=True)
        before = {k: p.detach().clone() for k, p in m.named_parameters()}
        flow = m(t1[None], t2[None], lat)
        loss = torch.norm(flow - tsf[None], p=2, dim=1).mean()
        opt.zero_grad(set_to_none=True)
        loss.backward()
        grads = {k: p.grad.detach().cpu().numpy() for k, p in m.named_parameters()}
        opt.step()
        torch.cuda.synchronize()
    finally:
        ops.BANK = bank
    gd = LO.generate_data(pc1, pc2, SCALES_FILTER_MAP)
    flow_o, loss_o, grads_o = TO.model_step(sd, pc1.T, pc2.T, sf.T, gd, dtype=torch.float64)
    assert abs(float(loss) - loss_o) < 1e-4
    assert np.abs(flow.detach()[0].cpu().numpy() - flow_o).max() < 2e-4 * max(1.0, float(np.abs(flow_o).max()))
    # fp32 sums over up to 35 k vertices against the exact gradient.  Measured on this pair (both arithmetic paths):
    # norms within 2e-5 .. 3e-5, every entry within 1e-4 of its tensor's largest -- the bars are 2e-4 / 5e-4 (round 2
    # accepted 1e-3 / 1e-2 without having measured what the kernels need; the isolated kernels hold 1e-4 below)
    _grads_close(grads, grads_o, 2e-4, 5e-4)
    # first Adam step from zero moments: every entry moves by lr * g / (|g| + eps) -- at most lr, towards -sign(g)
    for k, p in m.named_parameters():
        d = (p.detach() - before[k]).cpu().numpy()
        assert np.isfinite(d).all() and np.abs(d).max() <= 1e-4 * (1 + 1e-3), k
        g = grads[k]
        big = np.abs(g) > 1e-6 * max(1e-30, np.abs(g).max())
        assert np.array_equal(np.sign(d[big]), -np.sign(g[big])), k


def test_lattice_rebuild_determinism_n8192():
    """Two independent builds of the same pair's lattice (row orders are filled with atomics, so tile membership
    differs between builds) give bit-identical tables and a bit-identical flow: absent taps multiply zeros, and
    split-K cuts the contraction at slice indices, not at positions of a tile's slice list (gconv.hip)."""
    import hplflownet_amd as H
    pc1, pc2, sf = synthetic_pair(8192, 3)
    m, _ = full_model()
    m.eval()
    gen = H.GenerateDataUnsymmetric(model_args(), device=DEV, wide_up=m.lattice_hint())
    flows, perms = [], []
    for _ in range(3):
        t1, t2, _, lat = gen([pc1, pc2, sf])
        lat.prepare()
        with torch.no_grad():
            flows.append(m(t1[None], t2[None], lat).clone())
        perms.append(lat)
    a, b = H.to_reference_format(perms[0]), H.to_reference_format(perms[1])
    for x, y in zip(a, b):
        for k in x:
            assert (torch.equal(x[k], y[k]) if torch.is_tensor(x[k]) else x[k] == y[k]), k
    assert torch.equal(flows[0], flows[1]) and torch.equal(flows[0], flows[2])


def test_grad_allreduce_through_rccl_single_rank():
    """parallel.GradAllReducer over backend "nccl" (= RCCL) in a process group of one rank: the hooks launch every
    bucket's asynchronous all-reduce inside backward, the means are written back, gradients are unchanged."""
    import torch.distributed as dist
    import hplflownet_amd as H
    from hplflownet_amd import parallel
    if dist.is_initialized():
        pytest.skip('a process group already exists in this process')
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    dist.init_process_group('nccl', init_method='tcp://127.0.0.1:%d' % port, rank=0, world_size=1,
                            device_id=torch.device('cuda', 0))
    try:
        args = types.SimpleNamespace(dim=3, scales_filter_map=SCALES_FILTER_MAP[:5], evaluate=False, use_leaky=True,
                                     bcn_use_bias=True, bcn_use_norm=True, last_relu=False, DEVICE='cuda')
        m = H.HPLFlowNetShallow(args)
        fill_module_(m, 1.0, 'hash')
        m = m.to(DEV)
        parallel.broadcast_parameters(m)
        pc1, pc2, sf = synthetic_pair(1024, 0)
        gen = H.GenerateDataUnsymmetric(args, device=DEV)
        t1, t2, tsf, lat = gen([pc1, pc2, sf])
        lat.prepare(for_training=True)

        def run(reducer):
            m.zero_grad(set_to_none=True)
            loss = torch.norm(m(t1[None], t2[None], lat) - tsf[None], p=2, dim=1).mean()
            loss.backward()
            launched = reducer._next if reducer is not None else 0
            if reducer is not None:
                reducer()
            return launched, {k: p.grad.clone() for k, p in m.named_parameters()}
        _, plain = run(None)
        red = parallel.GradAllReducer(m.parameters(), bucket_bytes=1 << 20, single_rank=True)
        assert len(red.buckets) >= 3 and red.overlap
        launched, reduced = run(red)
        assert launched == len(red.buckets)          # every bucket went out from a gradient hook, inside backward
        scale = max(float(g.abs().max()) for g in plain.values())
        for k in plain:                              # (the weight gradient uses fp32 atomics: equal to rounding)
            assert float((plain[k] - reduced[k]).abs().max()) <= 1e-4 * float(plain[k].abs().max()) + 1e-7 * scale, k
        assert parallel.max_over_ranks(1.5, device=DEV) == 1.5
    finally:
        dist.destroy_process_group()


def test_weight_gradient_and_mirrored_data_gradient_at_bench_size_vs_float64():
    """The two backward kernels of the dominant layer in isolation, at its real size and on its real table (bcn1_ blur:
    H = 25 8xx vertices, 15 taps, C = 580, O = 1 024): the tap-list weight gradient (atomics over vertex slabs) and the
    mirrored-gather data gradient against float64.  The whole-model test above accepts entries within 1 % of a tensor's
    largest because LeakyReLU kinks upstream move single vertices; the kernels themselves hold 1e-4."""
    import hplflownet_amd as H
    from hplflownet_amd import ops
    pc1, pc2, sf = synthetic_pair(8192, 0)
    gen = H.GenerateDataUnsymmetric(model_args(False), device=DEV)
    t1, t2, _, lat = gen([pc1, pc2, sf])
    tb = lat.levels[0].blur[0]
    nbr = tb.t
    F, M = nbr.shape
    C, O = 580, 1024
    g = torch.Generator(device='cpu').manual_seed(7)
    A = torch.randn(M, C, generator=g).to(DEV)
    dY = (torch.randn(M, O, generator=g) * torch.exp(0.5 * torch.randn(M, 1, generator=g))).to(DEV)
    # ---- weight gradient dWt[f*C + c, o] = sum_m A[nbr[f, m], c] * dY[m, o]
    dWt = ops.wgrad_raw(A, nbr, M, C, F, dY, O, taps=ops.tap_lists(nbr))
    A64, dY64 = A.double(), dY.double()
    worst = 0.0
    for f in range(F):
        idx = nbr[f].long()
        rows = torch.where((idx >= 0)[:, None], A64[idx.clamp(min=0)], torch.zeros((), dtype=torch.float64, device=DEV))
        ref = rows.t() @ dY64                                           # [C, O]
        err = float((dWt[f * C:(f + 1) * C, :O].double() - ref).abs().max())
        worst = max(worst, err / float(ref.abs().max()))
    assert worst < 1e-4, worst
    # ---- data gradient through the symmetric table: dA[v, c] = sum_f sum_o dY[nbr[F-f][v], o] * W[(f), c, o] (mirrored taps)
    assert tb.symmetric
    W = (torch.randn(O, C, F, 1, generator=g) / (C * F) ** 0.5).to(DEV)
    A.requires_grad_(True)
    y = ops.gconv(A, W, None, nbr, M, F, bwd_mode=tb.bwd_mode(M), row_perm=tb.perm, taps=tb.taps)
    y.backward(dY)
    dA = A.grad
    ref = torch.zeros(M, C, dtype=torch.float64, device=DEV)
    W64 = W.double()[:, :, :, 0]                                        # [O, C, F]
    for f in range(F):
        idx = nbr[f].long()
        contrib = dY64 @ W64[:, :, f]                                   # [M, C]: what row m sends to its tap-f source
        ok = idx >= 0
        ref.index_add_(0, idx[ok], contrib[ok])
    rel = float((dA.double() - ref).abs().max() / ref.abs().max())
    assert rel < 1e-4, rel
