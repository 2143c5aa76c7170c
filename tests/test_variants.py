"""The constructor switches the main fixtures leave at their defaults -- use_leaky=False (ReLU, models/module_utils.py:14-17),
use_bias=False (no slice bias, models/bilateralNN.py:115-117), use_norm=False at the Down layers' real width C = 68
(models/bilateralNN.py:168-186) -- against tests/golden/layer_variants.npz, which tools/make_variant_fixture.py produced
by running the REFERENCE: the float64 torch oracle on the CPU (pins the oracle), the HIP layers and a whole shallow model
with all three switches flipped on the GPU (forward, every gradient)."""
import os
import sys
import types

import numpy as np
import pytest
import torch

from common import GOLD, oracle_lattice, rel_err
from hplflownet_amd.synthetic import SCALES_FILTER_MAP, closed_form_fill, fill_module_, subsample, synthetic_pair

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools'))
from make_variant_fixture import BCL_VARIANTS, CORR_VARIANTS  # noqa: E402  (the case table only; nothing of the reference)

TOL, RTOL = 1e-5, 3e-4          # oracle (float64) vs the reference's fp32: tests/test_oracle_torch.py
GTOL_FWD, GTOL = 2e-5, 5e-4     # HIP vs the reference: tests/test_gpu_layers.py


def Z():
    return np.load(os.path.join(GOLD, 'layer_variants.npz'))


def variant_convs(cin, couts, last_relu):
    convs, c = [], cin
    for i, o in enumerate(couts):
        F = 15 if i == 0 else 1
        bare = i == len(couts) - 1 and not last_relu
        base = 'blur_conv.%d.' % i + ('' if bare else 'composed_module.0.')
        convs.append((base, closed_form_fill(base + 'weight', (o, c, F, 1))[..., 0], closed_form_fill(base + 'bias', (o,))))
        c = o
    return convs


# ----------------------------------------------------------------------------------------------- CPU: the oracle
@pytest.mark.parametrize('case', BCL_VARIANTS, ids=[c[0] for c in BCL_VARIANTS])
def test_oracle_bilateral_conv_variants(case):
    from oracle import torch_oracle as TO
    tag, lvl, cin, couts, do_splat, do_slice, use_bias, use_leaky, use_norm, last_relu = case
    z = Z()
    _, _, _, gd = oracle_lattice(256)
    g = TO.lattice([gd[lvl]])[0]
    T = lambda a: TO._t(a, torch.float64)
    named = variant_convs(cin, couts, last_relu)
    convs = [(T(W).requires_grad_(True), T(b).requires_grad_(True)) for _, W, b in named]
    bias = T(closed_form_fill('slice_bias', (couts[-1],))).requires_grad_(True) if (do_slice and use_bias) else None
    nfeat = g['pc1_barycentric'].shape[1] if do_splat else g['pc1_hash_cnt']
    x = T(closed_form_fill(tag + '_x', (1, cin, nfeat))[0] * np.float32(np.sqrt(cin))).requires_grad_(True)
    y = TO.bilateral_conv_forward(x, convs, bias, g['pc1_barycentric'] if do_splat else None,
                                  g['pc1_lattice_offset'] if do_splat else None, g['pc1_blur_neighbors'],
                                  g['pc1_barycentric'] if do_slice else None, g['pc1_lattice_offset'] if do_slice else None,
                                  do_splat, do_slice, use_norm, use_leaky, last_relu, chunk=97)
    assert rel_err(subsample(y.detach().numpy()), z[tag + '_y']) < TOL
    go = T(closed_form_fill(tag + '_g', (1,) + tuple(y.shape))[0] * np.float32(np.sqrt(y.shape[0])))
    (y * go).sum().backward()
    assert rel_err(subsample(x.grad.numpy()), z[tag + '_gx']) < 5 * TOL
    params = bytes(z[tag + '_params']).decode().split('\n')
    assert ('bias' in params) == (do_slice and use_bias)          # the reference registers the slice bias only then
    if bias is not None:
        assert rel_err(bias.grad.numpy(), z[tag + '_g_bias']) < RTOL
    for (base, _, _), (W, b) in zip(named, convs):
        assert rel_err(subsample(W.grad.numpy()), z[tag + '_g_' + base + 'weight'].reshape(-1)) < RTOL
        assert rel_err(b.grad.numpy(), z[tag + '_g_' + base + 'bias']) < RTOL


def test_oracle_variant_model():
    from oracle import torch_oracle as TO
    from test_oracle_layers import hash_fill
    z = Z()
    pc1, pc2, sf, gd = oracle_lattice(256)
    names = bytes(z['vm_gradnames']).decode().split('\n')
    assert not any(n.endswith('_.bias') for n in names)              # bcn_use_bias=False: no slice-bias parameters
    import hplflownet_amd as H
    args = types.SimpleNamespace(dim=3, scales_filter_map=SCALES_FILTER_MAP[:5], evaluate=True, use_leaky=False,
                                 bcn_use_bias=False, bcn_use_norm=False, last_relu=False, DEVICE='cpu')
    shapes = {k: tuple(v.shape) for k, v in H.HPLFlowNetShallow(args).state_dict().items() if v.dtype == torch.float32}
    assert sorted(shapes) == sorted(names)
    sd = {k: hash_fill(k, s) for k, s in shapes.items()}
    flow, loss, grads = TO.model_step(sd, pc1.T, pc2.T, sf.T, gd[:5], shallow=True, use_leaky=False, use_norm=False)
    assert abs(loss - float(z['vm_loss'])) < 1e-4
    assert np.abs(flow - z['vm_flow']).max() < 2e-4 * max(1.0, np.abs(z['vm_flow']).max())
    ref = dict(zip(names, z['vm_gradnorm']))
    for k, g in grads.items():
        assert abs(np.linalg.norm(g) - ref[k]) < 2e-3 * max(ref[k], 1e-3), k


# ----------------------------------------------------------------------------------------------- GPU: the HIP layers
def T_(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to('cuda')


@pytest.mark.gpu
@pytest.mark.parametrize('case', BCL_VARIANTS, ids=[c[0] for c in BCL_VARIANTS])
def test_gpu_bilateral_conv_variants(case):
    import hplflownet_amd as H
    tag, lvl, cin, couts, do_splat, do_slice, use_bias, use_leaky, use_norm, last_relu = case
    z = Z()
    _, _, _, gd = oracle_lattice(256)
    g = gd[lvl]
    m = H.BilateralConvFlex(3, 1, cin, couts, 'cuda', use_bias, use_leaky, use_norm, do_splat, do_slice, last_relu, chunk_size=-1)
    fill_module_(m)
    assert [k for k, _ in m.named_parameters()] == bytes(z[tag + '_params']).decode().split('\n')
    if do_slice and use_bias:
        with torch.no_grad():
            m.bias.copy_(torch.from_numpy(closed_form_fill('slice_bias', (couts[-1],))))
    m = m.to('cuda')
    nfeat = g['pc1_barycentric'].shape[1] if do_splat else g['pc1_hash_cnt']
    x = T_(closed_form_fill(tag + '_x', (1, cin, nfeat)) * np.float32(np.sqrt(cin))).requires_grad_(True)
    y = m(x, T_(g['pc1_barycentric'])[None] if do_splat else None, T_(g['pc1_lattice_offset'])[None] if do_splat else None,
          T_(g['pc1_blur_neighbors'])[None], T_(g['pc1_barycentric'])[None] if do_slice else None,
          T_(g['pc1_lattice_offset'])[None] if do_slice else None)
    assert rel_err(subsample(y.detach().cpu().numpy()[0]), z[tag + '_y']) < GTOL_FWD
    go = T_(closed_form_fill(tag + '_g', tuple(y.shape)) * np.float32(np.sqrt(y.shape[1])))
    (y * go).sum().backward()
    assert rel_err(subsample(x.grad.cpu().numpy()[0]), z[tag + '_gx']) < 5 * GTOL_FWD
    for name, p in m.named_parameters():
        assert rel_err(subsample(p.grad.cpu().numpy()), z[tag + '_g_' + name].reshape(-1)) < GTOL, name


@pytest.mark.gpu
@pytest.mark.parametrize('case', CORR_VARIANTS, ids=[c[0] for c in CORR_VARIANTS])
def test_gpu_bilateral_corr_variants(case):
    import hplflownet_amd as H
    tag, lvl, prev_dim, corr_outs, outs, use_bias, use_leaky, use_norm = case
    z = Z()
    _, _, _, gd = oracle_lattice(256)
    g = gd[lvl]
    m = H.BilateralCorrelationFlex(3, 1, 1, 64, corr_outs, outs, 'cuda', use_bias, use_leaky, use_norm, prev_dim, False, chunk_size=-1)
    fill_module_(m)
    m = m.to('cuda')
    H1, H2 = g['pc1_hash_cnt'], g['pc2_hash_cnt']
    f1 = T_(closed_form_fill(tag + '_f1', (1, 64, H1)) * 8).requires_grad_(True)
    f2 = T_(closed_form_fill(tag + '_f2', (1, 64, H2)) * 8).requires_grad_(True)
    prev = T_(closed_form_fill(tag + '_prev', (1, prev_dim, g['pc1_barycentric'].shape[1])) * 8).requires_grad_(True)
    y = m(f1, f2, prev, T_(g['pc1_barycentric'])[None], T_(g['pc1_lattice_offset'])[None], T_(g['pc1_corr_indices'])[None],
          T_(g['pc2_corr_indices'])[None], H1, H2)
    assert rel_err(subsample(y.detach().cpu().numpy()[0]), z[tag + '_y']) < GTOL_FWD
    go = T_(closed_form_fill(tag + '_g', tuple(y.shape)) * 8)
    (y * go).sum().backward()
    assert rel_err(subsample(f1.grad.cpu().numpy()[0]), z[tag + '_gf1']) < 5 * GTOL_FWD
    assert rel_err(subsample(f2.grad.cpu().numpy()[0]), z[tag + '_gf2']) < 5 * GTOL_FWD
    assert rel_err(subsample(prev.grad.cpu().numpy()[0]), z[tag + '_gprev']) < 5 * GTOL_FWD
    for name, p in m.named_parameters():
        assert rel_err(subsample(p.grad.cpu().numpy()), z[tag + '_g_' + name].reshape(-1)) < GTOL, name


@pytest.mark.gpu
@pytest.mark.parametrize('native', [True, False], ids=['native-plan', 'python-path'])
def test_gpu_variant_model(native):
    """HPLFlowNetShallow with ReLU, no BCL biases, no density normalisation: device lattice + forward (both issue paths)
    + backward against the reference's run."""
    import hplflownet_amd as H
    z = Z()
    args = types.SimpleNamespace(dim=3, scales_filter_map=SCALES_FILTER_MAP[:5], evaluate=True, use_leaky=False,
                                 bcn_use_bias=False, bcn_use_norm=False, last_relu=False, DEVICE='cuda')
    m = H.HPLFlowNetShallow(args)
    fill_module_(m, 1.0, 'hash')
    m = m.to('cuda')
    pc1, pc2, sf = synthetic_pair(256, 0)
    gen = H.GenerateDataUnsymmetric(args, device='cuda', wide_up=m.lattice_hint())
    t1, t2, tsf, lat = gen([pc1, pc2, sf])
    m.native_forward = native
    with torch.no_grad():
        y = m.eval()(t1[None], t2[None], lat)
    ref = z['vm_flow']
    assert np.abs(y[0].cpu().numpy() - ref).max() < 2e-4 * max(1.0, np.abs(ref).max())
    flow = m.train()(t1[None], t2[None], lat)
    loss = torch.norm(flow - tsf[None], p=2, dim=1).mean()
    assert abs(float(loss) - float(z['vm_loss'])) < 1e-4
    loss.backward()
    want = dict(zip(bytes(z['vm_gradnames']).decode().split('\n'), z['vm_gradnorm']))
    for k, p in m.named_parameters():
        assert abs(float(p.grad.norm()) - want[k]) < 2e-3 * max(want[k], 1e-3), k
