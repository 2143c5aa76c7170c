"""GPU: the train / evaluate driver (hplflownet_amd.engine) replaying what the REFERENCE's
own loop did (fixture F9 of tools/make_fixtures.py: /root/reference/main.py:203-217 with Adam lr 1e-4 on
HPLFlowNetShallow, N=256, three steps on one pair) and its metrics (/root/reference/evaluation_utils.py:4-19)."""
import os

import numpy as np
import pytest
import torch

from common import GOLD
from hplflownet_amd.synthetic import synthetic_pair

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def test_three_adam_steps_replay_reference_loop():
    from hplflownet_amd import engine
    z = np.load(os.path.join(GOLD, 'train_steps.npz'))
    tr = engine.Trainer('HPLFlowNetShallow', DEV, lr=1e-4, init='hash')
    start = {k: p.detach().clone() for k, p in tr.model.named_parameters()}
    pc1, pc2, sf = synthetic_pair(256, 0)
    data = [tuple(torch.from_numpy(np.ascontiguousarray(a.T)).to(DEV) for a in (pc1, pc2, sf))]
    losses = [tr.train_epoch(data) for _ in range(3)]             # one pair per epoch = one optimiser step each
    want = z['adam3_losses']
    # step 1 is the plain forward; steps 2 and 3 see weights moved by Adam (+-lr per entry on the first step:
    # entries whose gradient is ~0 may move the other way, hence the looser bar -- the reference itself differs
    # by 1e-7 relative between two runs of this script)
    assert abs(losses[0] - want[0]) < 1e-4
    assert abs(losses[1] - want[1]) < 2e-3 * want[1] and abs(losses[2] - want[2]) < 2e-3 * want[2]
    names = bytes(z['adam3_names']).decode().split('\n')
    params = dict(tr.model.named_parameters())
    assert sorted(params) == sorted(names)           # (registration order differs: the wiring here is table driven)
    for k, n_ref, d_ref in zip(names, z['adam3_norms'], z['adam3_delta']):
        p = params[k].detach().double()
        assert abs(float(p.norm()) - n_ref) < 1e-4 * max(n_ref, 1e-3), k
        # three steps of size <= lr per entry, mostly in the same direction as the reference's
        d = float((p - start[k].double()).norm())
        assert abs(d - d_ref) < 0.15 * d_ref + 1e-9, (k, d, d_ref)
    w = params['bcn1_.blur_conv.0.weight'].detach().cpu().numpy().reshape(-1)[::61]
    assert np.abs(w - z['adam3_bcn1_w']).max() < 6.1e-4          # |3 steps| <= 3 lr; opposite signs at most 6 lr
    assert np.mean(np.abs(w - z['adam3_bcn1_w']) < 2e-5) > 0.9   # and nearly every entry took the same path


def test_flow_metrics_match_reference_evaluate_3d():
    from hplflownet_amd import engine
    z = np.load(os.path.join(GOLD, 'train_steps.npz'))
    got = engine.flow_metrics(torch.from_numpy(z['metrics_pred']).to(DEV), torch.from_numpy(z['metrics_gt']).to(DEV))
    epe, acc_s, acc_r, out = z['metrics_ref']
    assert abs(got['EPE3D'] - epe) < 1e-6
    assert abs(got['Acc3DS'] - acc_s) < 1e-6 and abs(got['Acc3DR'] - acc_r) < 1e-6 and abs(got['Outliers'] - out) < 1e-6
