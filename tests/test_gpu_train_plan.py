"""GPU: the native training step (hplflownet_amd.train_plan: forward + EPE3D loss + backward as ONE program of csrc/executor.hip)
against the autograd path over the same kernels (ops.GConvFn / SplatFn / SliceFn, itself pinned to the reference's gradients by
tests/test_gpu_autograd.py, test_gpu_bench_size.py and fixture F5/F9): same flow, same loss, every parameter gradient entry by entry."""
import types

import numpy as np
import pytest
import torch

from hplflownet_amd.synthetic import SCALES_FILTER_MAP, fill_module_, synthetic_pair

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def _setup(arch, n, seed=0):
    import hplflownet_amd as H
    nl = 7 if arch == 'HPLFlowNet' else 5
    a = types.SimpleNamespace(dim=3, scales_filter_map=SCALES_FILTER_MAP[:nl], evaluate=False, use_leaky=True, bcn_use_bias=True,
                              bcn_use_norm=True, last_relu=False, DEVICE='cuda')
    model = getattr(H, arch)(a)
    fill_module_(model, 1.0, 'hash')
    model = model.to(DEV).train()
    gen = H.GenerateDataUnsymmetric(a, device=DEV, wide_up=model.lattice_hint())
    pc1, pc2, sf = synthetic_pair(n, seed)
    t = [torch.from_numpy(np.ascontiguousarray(x.T)).to(DEV) for x in (pc1, pc2, sf)]
    return model, gen, t


def _autograd_step(model, t, lat):
    for p in model.parameters():
        p.grad = None
    flow = model(t[0][None], t[1][None], lat)
    loss = torch.norm(flow - t[2][None], p=2, dim=1).mean()
    loss.backward()
    return flow.detach().clone(), float(loss), {k: p.grad.detach().clone() for k, p in model.named_parameters()}


@pytest.mark.parametrize('arch,n,native_lat', [('HPLFlowNetShallow', 512, False), ('HPLFlowNet', 1024, True), ('HPLFlowNet', 4096, True)])
def test_native_step_matches_autograd(arch, n, native_lat):
    from hplflownet_amd.train_plan import TrainPlan
    model, gen, t = _setup(arch, n)
    lat = (gen.build_native(t[0], t[1]).device_lattice() if native_lat else gen.build(t[0], t[1])).prepare(True)
    flow_a, loss_a, grads_a = _autograd_step(model, t, lat)
    for side in (False, True):
        plan = TrainPlan(model, side_stream=side)
        r = plan.step(t[0], t[1], t[2], lat)
        assert r is not None
        plan.finish()
        torch.cuda.synchronize()
        flow_n, loss_n = r
        assert float((flow_n - flow_a).abs().max()) <= 1e-5 * float(flow_a.abs().max())
        assert abs(float(loss_n) - loss_a) <= 1e-5 * abs(loss_a)
        worst = ('', 0.0)
        for k, p in model.named_parameters():
            ga, gn = grads_a[k], p.grad
            assert gn.data_ptr() >= plan.gflat.data_ptr() and gn.shape == ga.shape          # still a view of the arena
            err = float((gn - ga).abs().max()) / max(float(ga.abs().max()), 1e-20)
            if err > worst[1]:
                worst = (k, err)
        # both paths sum the same products; they differ where sums are atomic (weight-gradient slabs, corr2 scatter)
        assert worst[1] < 2e-4, worst
        # a second step on the same plan (buffers reused, arenas re-zeroed): the same numbers
        g1 = plan.gflat.clone()
        for _ in range(6 if side else 1):      # (with the side stream: weight gradients beside the data-gradient chain, several times)
            plan.step(t[0], t[1], t[2], lat)
            plan.finish()
            torch.cuda.synchronize()
            assert float((plan.gflat - g1).abs().max()) <= 2e-4 * float(g1.abs().max())
        del plan


def test_trainer_takes_the_native_step_and_matches_the_autograd_loop():
    """engine.Trainer with and without the native step: three Adam steps on the same pair end at the same weights."""
    import os
    from hplflownet_amd import engine
    pc1, pc2, sf = synthetic_pair(512, 0)
    data = [tuple(torch.from_numpy(np.ascontiguousarray(a.T)).to(DEV) for a in (pc1, pc2, sf))]
    runs = {}
    for native in (True, False):
        tr = engine.Trainer('HPLFlowNetShallow', DEV, lr=1e-4, init='hash', native_step=native)
        losses = [tr.train_epoch(data) for _ in range(3)]
        assert (tr.native_steps == 3) if native else (tr.native_steps == 0)
        runs[native] = (losses, {k: p.detach().clone() for k, p in tr.model.named_parameters()})
    for a, b in zip(runs[True][0], runs[False][0]):
        assert abs(a - b) < 2e-3 * abs(b)
    for k in runs[True][1]:
        d = float((runs[True][1][k] - runs[False][1][k]).abs().max())
        assert d <= 6.1e-4, (k, d)                       # (Adam moves an entry by <= lr per step whatever the gradient's size)


def test_adam_flat_is_torch_adam():
    """hpl_adam_flat against torch.optim.Adam (main.py:138-140: lr 1e-4, weight_decay 0) on the same gradients, five steps, an odd
    length (scalar tail) and gradients over twelve decades."""
    from hplflownet_amd import _lib
    torch.manual_seed(3)
    n = 100003
    p0 = torch.randn(n, device=DEV)
    ref = torch.nn.Parameter(p0.clone())
    opt = torch.optim.Adam([ref], lr=1e-4)
    p, m, v = p0.clone(), torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    b1, b2, eps, lr = 0.9, 0.999, 1e-8, 1e-4
    scale = torch.pow(10.0, torch.randint(-8, 4, (n,), device=DEV).float())       # (per entry, the same in every step)
    for t in range(1, 6):
        g = torch.randn(n, device=DEV) * scale
        g[::97] = 0.0
        ref.grad = g.clone()
        opt.step()
        _lib.check(_lib.load().hpl_adam_flat(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), n, lr, b1, b2, eps, t, _lib.stream()),
                   'hpl_adam_flat')
        st = opt.state[ref]
        # (a step moves an entry by <= lr = 1e-4; one rounding of a parameter of size 4 is 4.8e-7)
        assert float((p - ref.detach()).abs().max()) <= 1e-6
        # (torch's unfused step rounds (1 - beta2) g g in another order; the lerp may cancel: the bar is in units of the entry's gradients)
        ratio = (m - st['exp_avg']).abs() / scale
        assert float(ratio.max()) <= 2e-6, (t, float(ratio.max()))
        assert torch.allclose(v, st['exp_avg_sq'], rtol=1e-5, atol=0)


def test_plan_adam_step_keeps_the_optimizer_state_and_follows_a_loaded_checkpoint():
    """TrainPlan.adam_step: the parameters are views of one flat array, the optimiser's moments views of two more, 'step' counts;
    opt.state_dict() keeps torch's format, and a state loaded with opt.load_state_dict() is adopted at the next step.  Two trainers,
    one stepping through the plan and one through torch's Adam on the same gradients, stay together."""
    from hplflownet_amd import engine
    pc1, pc2, sf = synthetic_pair(512, 0)
    data = [tuple(torch.from_numpy(np.ascontiguousarray(a.T)).to(DEV) for a in (pc1, pc2, sf))]
    tr = engine.Trainer('HPLFlowNetShallow', DEV, lr=1e-4, init='hash', native_step=True)
    tr.train_epoch(data)
    tr.train_epoch(data)
    plan = tr.tplan
    assert plan is not None and isinstance(plan._adam, list)
    params = list(tr.model.parameters())
    lo, hi = plan.pflat.data_ptr(), plan.pflat.data_ptr() + 4 * plan.pflat.numel()
    assert all(lo <= p.data_ptr() < hi for p in params)
    sd = tr.opt.state_dict()
    assert len(sd['state']) == len(params) and all(float(st['step']) == 2.0 for st in sd['state'].values())
    assert all(st['exp_avg'].shape == p.shape and st['exp_avg_sq'].shape == p.shape for st, p in zip(sd['state'].values(), params))
    # a checkpoint round trip: the loaded moments are adopted, the step count goes on from 2
    ck = tr.state()
    assert all(v.data_ptr() < lo or v.data_ptr() >= hi for v in ck['state_dict'].values() if v.is_cuda)      # tensors of their own
    tr2 = engine.Trainer('HPLFlowNetShallow', DEV, lr=1e-4, init='hash', native_step=True)
    tr2.model.load_state_dict({k[len('module.'):]: v for k, v in ck['state_dict'].items()})
    tr2.opt.load_state_dict(ck['optimizer'])
    l3a, l3b = tr.train_epoch(data), tr2.train_epoch(data)
    assert abs(l3a - l3b) < 1e-4 * abs(l3a)
    assert float(next(iter(tr2.opt.state_dict()['state'].values()))['step']) == 3.0
    for a, b in zip(tr.model.parameters(), tr2.model.parameters()):
        assert float((a - b).abs().max()) <= 2e-6          # (same state, same gradients up to the atomics of the weight gradients)
    # the inference path sees the stepped weights (version counters bumped behind the launch)
    with torch.no_grad():
        lat = tr.gen.build(data[0][0], data[0][1])
        tr.model.eval()
        f1 = tr.model(data[0][0][None], data[0][1][None], lat).clone()
        tr.model.train()
    tr.train_epoch(data)
    with torch.no_grad():
        tr.model.eval()
        f2 = tr.model(data[0][0][None], data[0][1][None], lat)
    assert float((f1 - f2).abs().max()) > 0.0


def test_training_lattices_die_with_their_last_reference():
    """A training lattice (device tables + tap lists) pins a 130-MB arena at N = 8 192: it must be freed by reference counting when
    the step that used it is over, not wait for Python's cycle collector (round 5: the second cloud's tables held the pair, the pair the
    cloud -- `bench.py --train` grew by 60-90 MB per step).  With the collector switched off an epoch leaves nothing behind."""
    import gc
    from hplflownet_amd import engine
    mk = lambda s: tuple(torch.from_numpy(np.ascontiguousarray(a.T)).to(DEV) for a in synthetic_pair(2048, s))
    data = [mk(s) for s in range(3)] * 4
    tr = engine.Trainer('HPLFlowNetShallow', DEV, lr=1e-4, init='hash')
    tr.train_epoch(data[:3])
    gc.collect()
    gc.disable()
    try:
        torch.cuda.synchronize()
        base = torch.cuda.memory_allocated()
        tr.train_epoch(data)
        torch.cuda.synchronize()
        assert torch.cuda.memory_allocated() <= base + (1 << 20), (torch.cuda.memory_allocated() - base) / 2 ** 20
    finally:
        gc.enable()


@pytest.mark.parametrize('fill', [float('nan'), 1e30])
def test_native_step_does_not_see_what_the_workspace_held(fill):
    """The training program lays its matrices out by lifetime and scales the wide launches' operands by the largest magnitude
    of what they read: neither may touch a cell this step has not written.  A workspace full of NaN -- or of 1e30, which a
    reduction over a stale cell would pick up as the scale and silently lose the small operands' low bits -- gives the flow, the
    loss and the forward-determined quantities of a clean one, bit for bit; gradients as close as two clean runs are (their slab
    sums are atomic)."""
    from hplflownet_amd.train_plan import TrainPlan
    model, gen, t = _setup('HPLFlowNet', 4096)
    lat = gen.build_native(t[0], t[1]).device_lattice().prepare(True)
    plan = TrainPlan(model, side_stream=True)

    def run(value):
        ws = plan._ws.get('train')
        if ws is not None and value is not None:
            torch.cuda.synchronize()
            ws[:ws.numel() // 16 * 16].view(torch.float32).fill_(value)
        flow, loss = plan.step(t[0], t[1], t[2], lat)
        plan.finish()
        torch.cuda.synchronize()
        return flow.clone(), float(loss), plan.gflat.clone()

    run(None)                                   # (allocates the workspace)
    f0, l0, g0 = run(0.0)
    f1, l1, g1 = run(fill)
    f2, l2, g2 = run(0.0)
    assert torch.equal(f0, f1) and l0 == l1 and bool(torch.isfinite(g1).all())
    scale = float(g0.abs().max())
    assert float((g1 - g0).abs().max()) <= max(2.0 * float((g2 - g0).abs().max()), 1e-6 * scale)


def test_refused_lattice_takes_the_documented_fallback_in_the_plans_bucket_order():
    """ADVICE (round 5): (a) a lattice the native program cannot take -- the reference's wire format: no pair tables -- makes step()
    return None (the documented autograd fallback) instead of raising from inside tables(); (b) the fallback's all-reduce goes out
    in the plan's own bucket order (what ranks on the native program issue), not the reducer's index order."""
    from hplflownet_amd.train_plan import TrainPlan
    model, gen, t = _setup('HPLFlowNetShallow', 512)
    plan = TrainPlan(model)
    lat = gen.build(t[0], t[1])
    import hplflownet_amd as H
    assert plan.step(t[0], t[1], t[2], H.to_reference_format(lat)) is None          # the reference's generated_data (a list of dicts)
    bogus = types.SimpleNamespace(levels=[])                       # not a lattice this plan accepts
    assert plan.step(t[0], t[1], t[2], bogus) is None
    seen = []
    plan.reducer.launch_flat = lambda b: seen.append(b)
    plan.reducer.finish_flat = lambda: seen.append('finish')
    plan.reduce_fallback()
    assert seen == list(plan.bucket_order) + ['finish'] and sorted(plan.bucket_order) == list(range(len(plan.reducer.buckets)))
