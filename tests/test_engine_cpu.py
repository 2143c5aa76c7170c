"""CPU: the train/eval driver's host logic -- metric formulas, loss, and the
checkpoint dict layout / file naming of the reference (main.py:183-189, main_utils.py:54-64)."""
import os

import numpy as np
import torch

from hplflownet_amd import engine
from hplflownet_amd.flownet import HPLFlowNetShallow, load_reference_checkpoint


def test_metrics_match_reference_formulas():
    rng = np.random.RandomState(0)
    gt = rng.randn(500, 3).astype(np.float32) * 0.3
    pred = gt + rng.randn(500, 3).astype(np.float32) * 0.08
    m = engine.flow_metrics(torch.from_numpy(pred), torch.from_numpy(gt))
    l2 = np.linalg.norm(gt - pred, axis=-1)                          # evaluation_utils.py:9-19 restated
    rel = l2 / (np.linalg.norm(gt, axis=-1) + 1e-4)
    want = {'EPE3D': l2.mean(), 'Acc3DS': np.logical_or(l2 < 0.05, rel < 0.05).mean(),
            'Acc3DR': np.logical_or(l2 < 0.1, rel < 0.1).mean(), 'Outliers': np.logical_or(l2 > 0.3, rel > 0.1).mean()}
    for k in want:
        assert abs(m[k] - want[k]) < 1e-6, k
    flow = torch.from_numpy(pred.T[None])
    assert abs(float(engine.epe3d_loss(flow, torch.from_numpy(gt.T[None]))) - l2.mean()) < 1e-6


def test_checkpoint_layout_round_trip(tmp_path):
    tr = engine.Trainer.__new__(engine.Trainer)                      # host-side state only: no device needed
    tr.arch, tr.epoch, tr.min_loss, tr.rank = 'HPLFlowNetShallow', 0, None, 0
    tr.model = HPLFlowNetShallow(engine.model_args(5, device='cpu'))
    tr.opt = torch.optim.Adam(tr.model.parameters(), lr=1e-4, weight_decay=0)
    tr.epoch, tr.min_loss = 11, 0.25
    path = tr.save_checkpoint(str(tmp_path), is_best=True)
    names = sorted(os.listdir(str(tmp_path)))
    assert names == ['checkpoint.pth.tar', 'checkpoint_11.pth.tar', 'model_best.pth.tar']     # epoch % 10 == 1
    ck = torch.load(path, map_location='cpu')
    assert sorted(ck) == ['arch', 'epoch', 'min_loss', 'optimizer', 'state_dict']
    assert all(k.startswith('module.') for k in ck['state_dict'])     # DataParallel-style keys (main.py:104,186)
    other = HPLFlowNetShallow(engine.model_args(5, device='cpu'))
    with torch.no_grad():
        for p in other.parameters():
            p.add_(1.0)
    load_reference_checkpoint(other, ck, strict=True)
    for (k, a), (_, b) in zip(tr.model.state_dict().items(), other.state_dict().items()):
        assert torch.equal(a, b), k
    tr2 = engine.Trainer.__new__(engine.Trainer)
    tr2.model, tr2.opt = other, torch.optim.Adam(other.parameters(), lr=1e-4)
    tr2.resume(path)
    assert tr2.epoch == 11 and tr2.min_loss == 0.25


def test_init_weights_and_shards():
    m = HPLFlowNetShallow(engine.model_args(5, device='cpu'))
    torch.manual_seed(0)
    engine.init_weights_(m, 'xavier')
    w = m.bcn1.blur_conv[0].weight                                    # (64, 68, 15, 1): fan_in 68*15, fan_out 64*15
    want = (2.0 / ((68 + 64) * 15)) ** 0.5
    assert abs(float(w.std()) - want) < 0.05 * want
    assert all(float(p.abs().max()) == 0.0 for n, p in m.named_parameters() if n.endswith('bias') and 'conv' in n)
    try:
        engine.init_weights_(m, 'nope')
        assert False
    except NotImplementedError:
        pass
    sh = [engine._Shard(list(range(10)), r, 4) for r in range(4)]
    assert sorted(x for s in sh for x in (s[i] for i in range(len(s)))) == list(range(10))
    assert [s[0] for s in sh] == [0, 1, 2, 3] and len(engine._Shard(list(range(10)), 1, 4, limit=2)) == 2
