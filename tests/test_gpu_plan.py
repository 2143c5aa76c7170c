"""GPU: the native forward executor (hplflownet_amd.plan / csrc/executor.hip) issues exactly the launches of the
Python inference path -- bit-identical flows -- for both models, frustum and surface-like pairs (the two orders
of the Up layers are chosen per level at run time), ragged pairs, after in-place weight updates and after the
parameters were replaced; and it is the path a plain `model(pc1, pc2, lattice)` takes."""
import types

import numpy as np
import pytest
import torch

from hplflownet_amd.synthetic import SCALES_FILTER_MAP, fill_module_, surface_pair, synthetic_pair

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def make(cls, nsc):
    import hplflownet_amd as H
    args = types.SimpleNamespace(dim=3, scales_filter_map=SCALES_FILTER_MAP[:nsc], evaluate=True, use_leaky=True,
                                 bcn_use_bias=True, bcn_use_norm=True, last_relu=False, DEVICE='cuda')
    m = getattr(H, cls)(args)
    fill_module_(m, 1.0, 'hash')
    m = m.to(DEV).eval()
    return m, H.GenerateDataUnsymmetric(args, device=DEV, wide_up=m.lattice_hint())


def both(m, t1, t2, lat):
    with torch.no_grad():
        m.native_forward = True
        a = m(t1[None], t2[None], lat).clone()
        m.native_forward = False
        b = m(t1[None], t2[None], lat).clone()
        m.native_forward = True
    return a, b


CASES = [('HPLFlowNet', 7, 'frustum', 8192, 8192), ('HPLFlowNet', 7, 'surface', 8192, 8192),
         ('HPLFlowNet', 7, 'frustum', 300, 211), ('HPLFlowNetShallow', 5, 'frustum', 4096, 4096),
         ('HPLFlowNetShallow', 5, 'surface', 2048, 1500), ('HPLFlowNet', 7, 'frustum', 40, 40)]


@pytest.mark.parametrize('cls,nsc,kind,n1,n2', CASES)
def test_native_plan_equals_python_path(cls, nsc, kind, n1, n2):
    m, gen = make(cls, nsc)
    pc1, pc2, sf = (surface_pair if kind == 'surface' else synthetic_pair)(max(n1, n2), 4)
    t1, t2, _, lat = gen([pc1[:n1], pc2[:n2], sf[:n1]])
    a, b = both(m, t1, t2, lat)
    assert tuple(a.shape) == (1, 3, n1) and torch.isfinite(a).all()
    assert torch.equal(a, b)
    # which order each Up layer took is a run-time decision per level: both occur over these cases
    shrink = [lv.clouds[0].N < lv.H[0] for lv in lat.levels[:nsc]]
    assert isinstance(shrink[0], bool)
    # a second pair through the same plan (other sizes, other workspace slot)
    pc1b, pc2b, sfb = synthetic_pair(max(n1, n2) // 2 + 3, 9)
    t1b, t2b, _, latb = gen([pc1b, pc2b, sfb])
    a2, b2 = both(m, t1b, t2b, latb)
    assert torch.equal(a2, b2)
    assert torch.equal(both(m, t1, t2, lat)[0], a)


def test_native_plan_follows_weight_updates_and_replacement():
    m, gen = make('HPLFlowNetShallow', 5)
    pc1, pc2, sf = synthetic_pair(1024, 1)
    t1, t2, _, lat = gen([pc1, pc2, sf])
    a0, b0 = both(m, t1, t2, lat)
    plan0 = m.forward_plan()
    with torch.no_grad():                                   # what an optimiser step does: in-place, version bump
        for p in m.parameters():
            p.mul_(1.01)
    a1, b1 = both(m, t1, t2, lat)
    assert m.forward_plan() is plan0                        # same storage: images refreshed, plan kept
    assert torch.equal(a1, b1) and not torch.equal(a1, a0)
    sd = {k: v.clone() * 0.5 for k, v in m.state_dict().items() if v.is_floating_point()}
    m.load_state_dict(sd, strict=False)                     # copies in place
    a2, b2 = both(m, t1, t2, lat)
    assert torch.equal(a2, b2) and not torch.equal(a2, a1)
    m.float().to('cpu').to(DEV)                             # parameters replaced by new tensors -> new plan
    a3, b3 = both(m, t1, t2, lat)
    assert m.forward_plan() is not plan0
    assert torch.equal(a3, b3) and torch.equal(a3, a2)


def test_native_plan_profile_brackets_the_wide_convs():
    m, gen = make('HPLFlowNet', 7)
    pc1, pc2, sf = synthetic_pair(8192, 0)
    t1, t2, _, lat = gen([pc1, pc2, sf])
    plan = m.forward_plan()
    with torch.no_grad():
        plan(t1, t2, lat)
        plan.profile(1)
        plan(t1, t2, lat)
        plan(t1, t2, lat)
        plan.profile(-1)
    n, ms = plan.profile_read()
    assert n == 8 and 0.2 < ms / n < 5.0                    # bcn1_, bcn2_ blur convs as two tap-group passes each
    assert plan.profile_read() == (0, 0.0)


def test_reference_format_lattice_still_takes_the_python_path():
    import hplflownet_amd as H
    m, gen = make('HPLFlowNetShallow', 5)
    pc1, pc2, sf = synthetic_pair(512, 2)
    t1, t2, _, lat = gen([pc1, pc2, sf])
    with torch.no_grad():
        a = m(t1[None], t2[None], lat)
        b = m(t1[None], t2[None], H.to_reference_format(lat))
    assert float((a - b).abs().max()) < 1e-5 * max(1.0, float(b.abs().max()))


@pytest.mark.parametrize('cls,nsc,kind,n1,n2', [('HPLFlowNet', 7, 'frustum', 8192, 8192), ('HPLFlowNet', 7, 'surface', 8192, 8000),
                                                ('HPLFlowNetShallow', 5, 'frustum', 4096, 4096), ('HPLFlowNet', 7, 'frustum', 50, 37)])
def test_native_lattice_builder_equals_python_driver(cls, nsc, kind, n1, n2):
    """csrc/lattice_builder.hip issues the stage calls of lattice.GenerateDataUnsymmetric.build_steps: every table of
    every level is bit-identical (the row orders are filled with atomics and compared as permutations of equal tap
    masks), and the forward on the native lattice equals the forward on the Python-built one."""
    import hplflownet_amd as H
    m, gen = make(cls, nsc)
    pc1, pc2, sf = (surface_pair if kind == 'surface' else synthetic_pair)(max(n1, n2), 6)
    t1 = torch.from_numpy(pc1[:n1].T.copy()).to(DEV)
    t2 = torch.from_numpy(pc2[:n2].T.copy()).to(DEV)
    lat_py = gen.build(t1, t2).prepare()
    lat_nv = gen.build_native(t1, t2)
    torch.cuda.synchronize()
    assert lat_nv.H == [lv.H for lv in lat_py.levels]
    a, b = H.to_reference_format(lat_nv), H.to_reference_format(lat_py)
    for L, (x, y) in enumerate(zip(a, b)):
        for k in y:
            assert (torch.equal(x[k], y[k]) if torch.is_tensor(y[k]) else x[k] == y[k]), (L, k)
    for L, (x, y) in enumerate(zip(lat_nv.levels, lat_py.levels)):
        assert all(torch.equal(p, q) for p, q in zip(x.pair.csr(), y.pair.csr())), L
        assert torch.equal(x.emg_pair, y.emg_pair)
        for tx, ty in ((x.blur.pair, y.blur.pair), (x.blur[0], y.blur[0])):
            assert (tx._perm is None) == (ty.perm is None) if tx._perm is not False else True
            if tx._perm is not None and tx._perm is not False:
                # same multiset of rows, and rows at the same position have the same tap mask
                assert torch.equal(torch.sort(tx._perm)[0], torch.arange(tx.t.shape[1], device=DEV, dtype=torch.int32))
                mask = lambda t, p: ((t.t[:, p.long()] >= 0).long() * (2 ** torch.arange(t.t.shape[0], device=DEV))[:, None]).sum(0)
                assert torch.equal(mask(tx, tx._perm), mask(ty, ty.perm))
    with torch.no_grad():
        y_nv = m(t1[None], t2[None], lat_nv).clone()
        y_py = m(t1[None], t2[None], lat_py).clone()
        m.native_forward = False
        y_nv_python_path = m(t1[None], t2[None], lat_nv).clone()
        m.native_forward = True
    assert torch.equal(y_nv, y_py) and torch.equal(y_nv_python_path, y_py)


def test_native_lattice_pipeline_and_arena_growth(monkeypatch):
    import hplflownet_amd as H
    from hplflownet_amd.lattice import LatticePipeline
    monkeypatch.setenv('HPL_LATTICE_FUSED', '0')       # the staged driver's arena is a guess that HPL_ENOMEM corrects
    m, gen = make('HPLFlowNetShallow', 5)
    gen.native_builder().bytes_per_point = 40          # far too small: the builder reports HPL_ENOMEM, the arena doubles
    sizes = [700, 64, 2000, 17]
    pairs = []
    for s, n in enumerate(sizes):
        p1, p2, _ = synthetic_pair(n, 30 + s)
        pairs.append((torch.from_numpy(p1.T.copy()).to(DEV), torch.from_numpy(p2.T.copy()).to(DEV)))
    side = torch.cuda.Stream()
    pipe = LatticePipeline(gen, lambda i: pairs[i], 0, len(pairs), depth=3, stream=side, native=True)
    for want in range(len(pairs)):
        (i, item), lat, ev = pipe.get()
        assert i == want
        ev.synchronize()
        ref = H.to_reference_format(gen.build(*pairs[i]))
        for L, (x, y) in enumerate(zip(H.to_reference_format(lat), ref)):
            for k in y:
                assert (torch.equal(x[k], y[k]) if torch.is_tensor(y[k]) else x[k] == y[k]), (i, L, k)
    assert gen.native_builder().bytes_per_point > 40


def test_threaded_native_lattice_pipeline():
    """The producer-thread form of the pipeline hands out the same lattices, in order, and surfaces errors."""
    import hplflownet_amd as H
    from hplflownet_amd.lattice import LatticePipeline
    m, gen = make('HPLFlowNetShallow', 5)
    pairs = []
    for s_, n in enumerate([900, 300, 1500, 64, 700]):
        p1, p2, _ = synthetic_pair(n, 50 + s_)
        pairs.append((torch.from_numpy(p1.T.copy()).to(DEV), torch.from_numpy(p2.T.copy()).to(DEV)))
    side = torch.cuda.Stream()
    pipe = LatticePipeline(gen, lambda i: pairs[i], 0, len(pairs), depth=2, stream=side, native=True, threaded=True)
    outs = []
    with torch.no_grad():
        for want in range(len(pairs)):
            (i, item), lat, ev = pipe.get()
            assert i == want and item is pairs[i]
            torch.cuda.current_stream().wait_event(ev)
            outs.append(m(item[0][None], item[1][None], lat).clone())
    with pytest.raises(StopIteration):
        pipe.get()
    with torch.no_grad():
        for i, (a, b) in enumerate(pairs):
            assert torch.equal(outs[i], m(a[None], b[None], gen.build(a, b)))

    def bad(i):
        raise ValueError('no such pair')
    pipe = LatticePipeline(gen, bad, 0, 2, depth=2, stream=side, native=True, threaded=True)
    with pytest.raises(ValueError):
        pipe.get()
