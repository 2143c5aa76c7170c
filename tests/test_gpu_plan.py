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
    assert n == 8 and 0.5 < ms / n < 5.0                    # bcn1_, bcn2_ blur convs as two tap-group passes each
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
