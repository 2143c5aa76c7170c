"""CPU: module construction reproduces the reference's state_dict names and shapes (F6)."""
import json
import os
import types

import pytest

from common import GOLD
from hplflownet_amd.synthetic import SCALES_FILTER_MAP


def _args(n):
    return types.SimpleNamespace(dim=3, scales_filter_map=SCALES_FILTER_MAP[:n], evaluate=True, use_leaky=True,
                                 bcn_use_bias=True, bcn_use_norm=True, last_relu=False, DEVICE='cuda')


@pytest.mark.parametrize('name,n', [('HPLFlowNet', 7), ('HPLFlowNetShallow', 5)])
def test_state_dict_matches_reference(name, n):
    import hplflownet_amd as H
    manifest = json.load(open(os.path.join(GOLD, 'state_dict.json')))[name]
    m = getattr(H, name)(_args(n))
    sd = m.state_dict()
    assert set(sd.keys()) == set(manifest.keys())
    for k, v in sd.items():
        dt, *shape = manifest[k]
        assert list(v.shape) == shape and str(v.dtype) == 'torch.' + dt, k


def test_load_reference_style_checkpoint(tmp_path):
    """A checkpoint shaped like the reference's (DataParallel 'module.' prefix, 'state_dict' entry,
    main_utils.py:54-64) loads with strict=True, buffers included."""
    import torch
    import hplflownet_amd as H
    from hplflownet_amd.flownet import load_reference_checkpoint
    from hplflownet_amd.synthetic import fill_module_
    src = H.HPLFlowNetShallow(_args(5))
    fill_module_(src, 1.0, 'hash')
    ckpt = {'epoch': 3, 'arch': 'HPLFlowNetShallow', 'min_loss': 0.1,
            'state_dict': {'module.' + k: v.clone() for k, v in src.state_dict().items()}}
    path = str(tmp_path / 'checkpoint.pth.tar')
    torch.save(ckpt, path)
    dst = H.HPLFlowNetShallow(_args(5))
    res = load_reference_checkpoint(dst, path, strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    for (ka, a), (kb, b) in zip(sorted(src.state_dict().items()), sorted(dst.state_dict().items())):
        assert ka == kb and torch.equal(a, b)
