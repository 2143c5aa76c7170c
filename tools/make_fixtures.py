#!/usr/bin/env python
"""Generate the golden vectors under tests/golden/ by running the REFERENCE itself.

Runs only in the build container (needs /root/reference, imported through
tools/ref_import.py with the numba/khash shims; `--khash ref` uses the reference's
own khash compiled into oracle/_ref).  Outputs are data only: inputs are
regenerated from seeds / closed-form fills (hplflownet_amd/synthetic.py), the
files hold the reference's outputs.  Fixture ids follow SURVEY.md §8(c5):

  F1 constants.npz      neighbour offsets r=1,2 (Traverse), elevation matrix, canonical
  F2 keys_n1024.npz     get_keys_and_barycentric on the N=1024 seed-0 pair, scales 3 and 1
  F3 lattice_n256.npz   full 7-level generated_data, N=256 seed 0 (int tables as int32)
     lattice_n1024.npz  levels 0..2 in full + sha256 of every array of all 7 levels
  F4 layers.npz         single-layer forward outputs and gradients (arrays > 16384 elements
                        stored as a flat stride-5 subsample, synthetic.subsample)
  F5 models.npz         whole-model outputs, EPE3D loss, per-parameter grad norms
  F6 state_dict.json    parameter/buffer names + shapes of both models
  F8 models_large.npz   benchmark-size runs of the reference itself (BASELINE configs 3 and 4): full HPLFlowNet
                        forward at N=8192 on two frustum seeds and one surface pair (flow subsample + EPE3D loss),
                        full HPLFlowNet train-mode (chunked blur, HPLFlowNet.py:17) forward + backward at N=4096
                        (loss, flow subsample, per-parameter gradient norms)
  F9 train_steps.npz    three optimiser steps of the reference training loop (main.py:203-217: Adam lr 1e-4,
                        EPE3DLoss mean) on HPLFlowNetShallow, N=256: losses, per-parameter norms after step 3,
                        a strided sample of one weight; evaluate_3d (evaluation_utils.py:4-19) on a fixed
                        prediction / ground-truth pair
"""
import argparse
import hashlib
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

from ref_import import import_reference  # noqa: E402
from hplflownet_amd.synthetic import (SCALES_FILTER_MAP, closed_form_fill,  # noqa: E402
                                      fill_module_, subsample, surface_pair, synthetic_pair)

GOLD = os.path.join(ROOT, 'tests', 'golden')


def sha(a):
    a = np.ascontiguousarray(a)
    return hashlib.sha256(a.tobytes()).hexdigest()


def model_args(nscales, evaluate=True):
    return types.SimpleNamespace(dim=3, scales_filter_map=SCALES_FILTER_MAP[:nscales], evaluate=evaluate,
                                 use_leaky=True, bcn_use_bias=True, bcn_use_norm=True, last_relu=False,
                                 DEVICE='cpu')


def gd_to_numpy(gd):
    out = []
    for d in gd:
        out.append({k: (v.numpy() if hasattr(v, 'numpy') else np.int64(v)) for k, v in d.items()})
    return out


def gd_batched(gd):
    """What default_collate does to generated_data (adds B=1; ints -> 1-elem tensors)."""
    out = []
    for d in gd:
        out.append({k: (v[None] if hasattr(v, 'numpy') else torch.tensor([v])) for k, v in d.items()})
    return out


def save_lattice(path, gdn, full_levels):
    arrs = {}
    digest = {}
    for l, d in enumerate(gdn):
        for k, v in d.items():
            v = np.asarray(v)
            digest['L%d_%s' % (l, k)] = sha(v.astype(np.int64) if v.dtype.kind == 'i' else v)
            if l in full_levels:
                arrs['L%d_%s' % (l, k)] = v.astype(np.int32) if v.dtype.kind == 'i' else v
    arrs['sha256_json'] = np.frombuffer(json.dumps(digest, sort_keys=True).encode(), dtype=np.uint8)
    np.savez_compressed(path, **arrs)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--khash', default='ref', choices=['ref', 'dict'])
    ap.add_argument('--only', default='')
    a = ap.parse_args()
    only = set(a.only.split(',')) if a.only else None

    def want(x):
        return only is None or x in only

    R = import_reference('ref' if a.khash == 'ref' else 'dict')
    T = R.T
    os.makedirs(GOLD, exist_ok=True)
    torch.manual_seed(0)
    gen7 = T.GenerateDataUnsymmetric(types.SimpleNamespace(dim=3, scales_filter_map=SCALES_FILTER_MAP))

    # ------------------------------------------------------------------ F1
    if want('F1'):
        gen_r2 = T.GenerateDataUnsymmetric(types.SimpleNamespace(dim=3, scales_filter_map=[[1., 2, 2, 2]]))
        np.savez_compressed(os.path.join(GOLD, 'constants.npz'),
                            offsets_r1=gen7.radius2offset[1].astype(np.int32),
                            offsets_r2=gen_r2.radius2offset[2].astype(np.int32),
                            elevate=gen7.elevate_mat.numpy(), canonical=gen7.canonical.numpy().astype(np.int32),
                            expected_std=np.float64(gen7.expected_std))

    # ------------------------------------------------------------------ F2
    pc1k, pc2k, sfk = synthetic_pair(1024, 0)
    if want('F2'):
        out = {}
        for s in (3.0, 1.0):
            for nm, pc in (('pc1', pc1k), ('pc2', pc2k)):
                p = torch.from_numpy(pc.T.copy())
                p *= s
                keys, bary, emg = gen7.get_keys_and_barycentric(p)
                tag = '%s_s%d' % (nm, int(s))
                out[tag + '_keys'] = keys.astype(np.int32)
                out[tag + '_bary'] = bary.numpy()
                out[tag + '_emg'] = emg.numpy()
        np.savez_compressed(os.path.join(GOLD, 'keys_n1024.npz'), **out)

    # ------------------------------------------------------------------ F3
    lat = {}
    for n, full in ((256, range(7)), (1024, range(3))):
        pc1, pc2, sf = synthetic_pair(n, 0)
        _, _, _, gd = gen7([pc1.copy(), pc2.copy(), sf.copy()])
        lat[n] = (pc1, pc2, sf, gd)
        if want('F3'):
            save_lattice(os.path.join(GOLD, 'lattice_n%d.npz' % n), gd_to_numpy(gd), set(full))
        print('lattice N=%d H1=%s' % (n, [d['pc1_hash_cnt'] for d in gd]))

    # ------------------------------------------------------------------ F4
    if want('F4'):
        out = {}
        gd256 = lat[256][3]
        gd1024 = lat[1024][3]

        def run_bcl(tag, gd_in, gd_out, cin, couts, do_splat, do_slice, last_relu=False, use_norm=True,
                    n_in=None):
            m = R.BilateralConvFlex(3, 1, cin, couts, 'cpu', True, True, use_norm, do_splat, do_slice,
                                    last_relu, chunk_size=-1)
            fill_module_(m)
            if do_slice:
                with torch.no_grad():
                    m.bias.copy_(torch.from_numpy(closed_form_fill('slice_bias', (couts[-1],))))
            H = gd_in['pc1_hash_cnt']
            n_feat = gd_in['pc1_barycentric'].shape[1] if do_splat else H
            x = torch.from_numpy(closed_form_fill(tag + '_x', (1, cin, n_feat)) * np.float32(np.sqrt(cin)))
            x.requires_grad_(True)
            y = m(x,
                  gd_in['pc1_barycentric'][None] if do_splat else None,
                  gd_in['pc1_lattice_offset'][None] if do_splat else None,
                  gd_in['pc1_blur_neighbors'][None],
                  gd_out['pc1_barycentric'][None] if do_slice else None,
                  gd_out['pc1_lattice_offset'][None] if do_slice else None)
            g = torch.from_numpy(closed_form_fill(tag + '_g', tuple(y.shape)) * np.float32(np.sqrt(y.shape[1])))
            (y * g).sum().backward()
            out[tag + '_y'] = subsample(y.detach().numpy()[0])
            out[tag + '_gx'] = subsample(x.grad.numpy()[0])
            for name, p in m.named_parameters():
                out[tag + '_g_' + name] = subsample(p.grad.numpy())

        # BASELINE config 1: splat+blur+slice at N=1024 level 0
        run_bcl('cfg1', gd1024[0], gd1024[0], 68, [64, 64], True, True)
        # Down (splat, no slice) level 0 and level 2 of N=256
        run_bcl('down0', gd256[0], None, 68, [64, 64], True, False)
        run_bcl('down2', gd256[2], None, 68, [64, 64], True, False)
        # Up (no splat, slice): features live on level-2 vertices, sliced to level-2 input points
        run_bcl('up2', gd256[2], gd256[2], 36, [32, 32], False, True)
        # single-conv variants used by the shallow model, last_relu and no-norm switches
        run_bcl('down1_single', gd256[1], None, 68, [64], True, False)
        run_bcl('up1_single_relu', gd256[1], gd256[1], 20, [32], False, True, last_relu=True)
        run_bcl('cfg_nonorm', gd256[0], gd256[0], 12, [16, 16], True, True, use_norm=False)

        def run_corr(tag, lvl, prev_dim, corr_outs, outs):
            g = gd256[lvl]
            m = R.BilateralCorrelationFlex(3, 1, 1, 64, corr_outs, outs, 'cpu', True, True, True, prev_dim,
                                           False, chunk_size=-1)
            fill_module_(m)
            H1, H2 = g['pc1_hash_cnt'], g['pc2_hash_cnt']
            f1 = torch.from_numpy(closed_form_fill(tag + '_f1', (1, 64, H1)) * 8).requires_grad_(True)
            f2 = torch.from_numpy(closed_form_fill(tag + '_f2', (1, 64, H2)) * 8).requires_grad_(True)
            prev = None
            if prev_dim:
                n_in = g['pc1_barycentric'].shape[1]
                prev = torch.from_numpy(closed_form_fill(tag + '_prev', (1, prev_dim, n_in)) * 8)
                prev.requires_grad_(True)
            y = m(f1, f2, prev,
                  g['pc1_barycentric'][None] if prev_dim else None,
                  g['pc1_lattice_offset'][None] if prev_dim else None,
                  g['pc1_corr_indices'][None], g['pc2_corr_indices'][None], H1, H2)
            go = torch.from_numpy(closed_form_fill(tag + '_g', tuple(y.shape)) * 8)
            (y * go).sum().backward()
            out[tag + '_y'] = subsample(y.detach().numpy()[0])
            out[tag + '_gf1'] = subsample(f1.grad.numpy()[0])
            out[tag + '_gf2'] = subsample(f2.grad.numpy()[0])
            if prev is not None:
                out[tag + '_gprev'] = subsample(prev.grad.numpy()[0])
            for name, p in m.named_parameters():
                out[tag + '_g_' + name] = subsample(p.grad.numpy())

        run_corr('corr_noprev', 2, 0, [32, 32], [64, 64])
        run_corr('corr_prev', 3, 64, [32, 32], [64, 64])
        run_corr('corr_shallow', 4, 64, [32], [32])

        # sparse_sum alone (a1)
        idx = (gd256[0]['pc1_lattice_offset'] + 1).reshape(1, -1)
        vals = torch.from_numpy(closed_form_fill('ss_vals', (idx.shape[1], 5))).requires_grad_(True)
        ss = R.sparse_sum(idx, vals, torch.Size([gd256[0]['pc1_hash_cnt'] + 1, 5]), False)
        (ss * ss).sum().backward()
        out['ss_y'] = ss.detach().numpy()
        out['ss_gvals'] = vals.grad.numpy()
        np.savez_compressed(os.path.join(GOLD, 'layers.npz'), **out)

    # ------------------------------------------------------------------ F5 / F6
    if want('F5') or want('F6'):
        out = {}
        manifest = {}
        for tag, cls, nsc, n in (('shallow_n256', R.HPLFlowNetShallow, 5, 256),
                                 ('shallow_n1024', R.HPLFlowNetShallow, 5, 1024),
                                 ('full_n256', R.HPLFlowNet, 7, 256)):
            pc1, pc2, sf, gd = lat[n]
            m = cls(model_args(nsc))
            fill_module_(m, 1.0, 'hash')
            manifest[cls.__name__] = {k: [str(v.dtype).replace('torch.', '')] + list(v.shape)
                                      for k, v in m.state_dict().items()}
            p1 = torch.from_numpy(pc1.T.copy())[None]
            p2 = torch.from_numpy(pc2.T.copy())[None]
            y = m(p1, p2, gd_batched(gd[:nsc]))
            tgt = torch.from_numpy(sf.T.copy())[None]
            loss = torch.norm(y - tgt, p=2, dim=1).mean()          # main.py:213, epe3d_loss.py:9-10
            loss.backward()
            out[tag + '_flow'] = y.detach().numpy()[0]
            out[tag + '_loss'] = np.float64(loss.item())
            names = [k for k, _ in m.named_parameters()]
            out[tag + '_gradnorm'] = np.array([p.grad.norm().item() for _, p in m.named_parameters()])
            out[tag + '_gradnames'] = np.frombuffer('\n'.join(names).encode(), dtype=np.uint8)
            print(tag, 'loss', loss.item(), 'flow abs mean', y.abs().mean().item())
        if want('F5'):
            np.savez_compressed(os.path.join(GOLD, 'models.npz'), **out)
        if want('F6'):
            with open(os.path.join(GOLD, 'state_dict.json'), 'w') as f:
                json.dump(manifest, f, indent=0, sort_keys=True)

    # ------------------------------------------------------------------ F8: benchmark-size runs of the reference
    if want('F8'):
        out = {}

        def ref_lattice(pc1, pc2, sf):
            _, _, _, gd = gen7([pc1.copy(), pc2.copy(), sf.copy()])
            return gd

        for tag, (pc1, pc2, sf) in (('full_n8192_s0', synthetic_pair(8192, 0)),
                                    ('full_n8192_s1', synthetic_pair(8192, 1)),
                                    ('surf_n8192_s0', surface_pair(8192, 0))):
            gd = ref_lattice(pc1, pc2, sf)
            m = R.HPLFlowNet(model_args(7, evaluate=True))
            fill_module_(m, 1.0, 'hash')
            with torch.no_grad():
                y = m(torch.from_numpy(pc1.T.copy())[None], torch.from_numpy(pc2.T.copy())[None], gd_batched(gd))
            loss = torch.norm(y - torch.from_numpy(sf.T.copy())[None], p=2, dim=1).mean()
            out[tag + '_flow'] = subsample(y.numpy()[0])
            out[tag + '_loss'] = np.float64(loss.item())
            out[tag + '_H1'] = np.array([d['pc1_hash_cnt'] for d in gd], np.int64)
            print(tag, 'loss', loss.item(), 'H1', out[tag + '_H1'].tolist())
        # BASELINE config 4 at half size: train mode = chunk_size 25 M (HPLFlowNet.py:17), forward + backward
        pc1, pc2, sf = synthetic_pair(4096, 0)
        gd = ref_lattice(pc1, pc2, sf)
        m = R.HPLFlowNet(model_args(7, evaluate=False))
        fill_module_(m, 1.0, 'hash')
        m.train()
        y = m(torch.from_numpy(pc1.T.copy())[None], torch.from_numpy(pc2.T.copy())[None], gd_batched(gd))
        loss = torch.norm(y - torch.from_numpy(sf.T.copy())[None], p=2, dim=1).mean()
        loss.backward()
        tag = 'train_n4096_s0'
        out[tag + '_flow'] = subsample(y.detach().numpy()[0])
        out[tag + '_loss'] = np.float64(loss.item())
        names = [k for k, _ in m.named_parameters()]
        out[tag + '_gradnorm'] = np.array([p.grad.norm().item() for _, p in m.named_parameters()])
        out[tag + '_gradnames'] = np.frombuffer('\n'.join(names).encode(), dtype=np.uint8)
        print(tag, 'loss', loss.item())
        np.savez_compressed(os.path.join(GOLD, 'models_large.npz'), **out)

    # ------------------------------------------------------------------ F9: the reference's training loop + metrics
    if want('F9'):
        out = {}
        pc1, pc2, sf, gd = lat[256]
        m = R.HPLFlowNetShallow(model_args(5, evaluate=False))
        fill_module_(m, 1.0, 'hash')
        m.train()
        opt = torch.optim.Adam(m.parameters(), lr=1e-4, weight_decay=0)             # main.py:138-140
        p1, p2 = torch.from_numpy(pc1.T.copy())[None], torch.from_numpy(pc2.T.copy())[None]
        tgt = torch.from_numpy(sf.T.copy())[None]
        losses = []
        for _ in range(3):                                                           # main.py:209-217
            output = m(p1, p2, gd_batched(gd[:5]))
            loss = torch.norm(output - tgt, p=2, dim=1).mean()
            opt.zero_grad()
            loss.backward()
            opt.step()
            losses.append(loss.item())
        names = [k for k, _ in m.named_parameters()]
        out['adam3_losses'] = np.array(losses, np.float64)
        out['adam3_names'] = np.frombuffer('\n'.join(names).encode(), dtype=np.uint8)
        out['adam3_norms'] = np.array([p.detach().double().norm().item() for _, p in m.named_parameters()])
        out['adam3_bcn1_w'] = dict(m.named_parameters())['bcn1_.blur_conv.0.weight'].detach().numpy().reshape(-1)[::61].copy()
        # per-parameter change against the start: what three Adam steps of lr 1e-4 did
        start = R.HPLFlowNetShallow(model_args(5, evaluate=False))
        fill_module_(start, 1.0, 'hash')
        out['adam3_delta'] = np.array([(p.detach() - q.detach()).double().norm().item()
                                       for (_, p), (_, q) in zip(m.named_parameters(), start.named_parameters())])
        print('adam3 losses', losses)
        # metrics: evaluation_utils.evaluate_3d uses np.float (removed in numpy >= 1.24): alias for the call
        if not hasattr(np, 'float'):
            np.float = float
        import evaluation_utils as EU
        rng = np.random.RandomState(5)
        gt = rng.normal(0, 0.4, (1024, 3)).astype(np.float32)
        pred = (gt + rng.normal(0, 0.08, gt.shape) * rng.uniform(0, 3, (1024, 1))).astype(np.float32)
        out['metrics_pred'] = pred
        out['metrics_gt'] = gt
        out['metrics_ref'] = np.array(EU.evaluate_3d(pred, gt), np.float64)
        print('evaluate_3d', out['metrics_ref'])
        np.savez_compressed(os.path.join(GOLD, 'train_steps.npz'), **out)
    for fn in sorted(os.listdir(GOLD)):
        print('%-24s %8.1f KB' % (fn, os.path.getsize(os.path.join(GOLD, fn)) / 1024.))


if __name__ == '__main__':
    main()
