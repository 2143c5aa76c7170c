"""Differentiable CPU restatement of the reference's bilateral layers and models (torch, fp32 or fp64).

TEST INFRASTRUCTURE ONLY.  Only tests/ and tools/ may import this module; nothing under
hplflownet_amd/ does.  It exists because the numpy oracle (bcl_oracle.py) has a hand-written backward
for BilateralConvFlex only: here every gradient -- BilateralCorrelationFlex, whole models, training steps
at the benchmark size -- comes from torch autograd over a plain restatement of the reference's forward
(the floating-point reference this tier allows for a floating-point kernel).  Running it in float64 gives
the error of BOTH fp32 implementations (the reference's and the HIP one) against the same exact answer.

Parity status: PINNED against the reference's own outputs (tests/test_oracle_torch.py): F4 single layers
(forward + every gradient), F5 whole models (flow, loss, gradient norms), F8 the N=4096 train-mode step.

Layout follows the reference: channel-first (C, N) tensors with the batch dimension (always 1) dropped,
int64 index tables.  The gathers of the blur / correlation are evaluated vertex chunk by vertex chunk under
torch.utils.checkpoint (recomputed in backward), so the (C, 15, 15, H) tensor the reference materialises
(models/bnn_flow.py:189-199) never exists in full.
"""
import numpy as np
import torch
from torch.utils.checkpoint import checkpoint

LEAKY_RATE = 0.1   # models/module_utils.py:6


def _t(a, dtype):
    if torch.is_tensor(a):
        return a.to(dtype) if a.is_floating_point() else a
    a = np.asarray(a)
    t = torch.from_numpy(np.ascontiguousarray(a))
    return t.to(dtype) if t.is_floating_point() else t.long()


def leaky(x, use_leaky=True):
    """models/module_utils.py:14,33,50."""
    return torch.nn.functional.leaky_relu(x, LEAKY_RATE if use_leaky else 0.0)


def sparse_sum(idx, values, rows):
    """models/bilateralNN.py:9-30: out[idx[j], :] += values[j, :]."""
    out = torch.zeros((rows, values.shape[1]), dtype=values.dtype)
    return out.index_add(0, idx.reshape(-1), values)


def splat(features, bary, off, H, use_norm=True):
    """models/bilateralNN.py:151-186: features (C, N), bary (4, N), off (4, N) -> (C, H+1), column 0 the null vertex."""
    C = features.shape[0]
    tmp = (bary[None, :, :] * features[:, None, :]).reshape(C, -1).t()          # (4N, C)
    idx = (off + 1).reshape(-1)
    S = sparse_sum(idx, tmp, H + 1).t()
    if use_norm:
        w = sparse_sum(idx, bary.reshape(-1, 1), H + 1)[:, 0]
        S = S * (1.0 / (w + 1e-5))[None, :]
    return S


def _conv_stack(x2d, convs, acts, use_leaky):
    cur = x2d
    for (W, b), act in zip(convs, acts):
        cur = W.reshape(W.shape[0], -1) @ cur + b[:, None]
        if act:
            cur = leaky(cur, use_leaky)
    return cur


def bilateral_conv_forward(features, convs, bias, in_bary, in_off, blur_nbr, out_bary, out_off, do_splat, do_slice,
                           use_norm=True, use_leaky=True, last_relu=False, chunk=4096):
    """models/bilateralNN.py:122-238.  convs: list of (W (O, C, F), b (O,))."""
    H = blur_nbr.shape[-1]
    if do_splat:
        S = splat(features, in_bary, in_off, H, use_norm)
    else:                                                                        # :190-196
        S = torch.cat([torch.zeros((features.shape[0], 1), dtype=features.dtype), features], dim=1)
    acts = [True] * (len(convs) - 1) + [bool(last_relu)]
    flat = [t for Wb in convs for t in Wb]

    def blur(S, nbr, *flat):
        cv = [(flat[2 * i], flat[2 * i + 1]) for i in range(len(flat) // 2)]
        X = S[:, nbr + 1]                                                        # (C, F, h)  :215-217
        return _conv_stack(X.reshape(-1, X.shape[-1]), cv, acts, use_leaky)      # :219
    outs = [checkpoint(blur, S, blur_nbr[:, s:s + chunk], *flat, use_reentrant=False) for s in range(0, H, chunk)]
    Y = torch.cat(outs, dim=1)
    if not do_slice:
        return Y
    out = (out_bary[None] * Y[:, out_off]).sum(dim=1)                            # :226-231
    if bias is not None:
        out = out + bias[:, None]                                                # :235-236
    return out


def bilateral_corr_forward(feat1, feat2, prev_corr, bary1, off1, corr_idx1, corr_idx2, corr_convs, blur_convs,
                           use_norm=True, use_leaky=True, last_relu=False, chunk=256):
    """models/bnn_flow.py:96-210.  corr_convs: [(W (O, Ctot, K), b)], then (O, C, 1); blur_convs as above.
    Channel order of the Conv3d input: [prev | feat1 | feat2] (:168,199)."""
    H1 = feat1.shape[1]
    z = torch.zeros((feat1.shape[0], 1), dtype=feat1.dtype)
    S1 = torch.cat([z, feat1], dim=1)
    S2 = torch.cat([z, feat2], dim=1)
    if prev_corr is not None:                                                    # :119-151,164-165
        S1 = torch.cat([splat(prev_corr, bary1, off1, H1, use_norm), S1], dim=0)
    F = corr_idx2.shape[0]
    acts_b = [True] * (len(blur_convs) - 1) + [bool(last_relu)]
    nc = len(corr_convs)
    flat = [t for Wb in list(corr_convs) + list(blur_convs) for t in Wb]

    def body(S1, S2, i1, i2, *flat):
        cc = [(flat[2 * i], flat[2 * i + 1]) for i in range(nc)]
        bc = [(flat[2 * i], flat[2 * i + 1]) for i in range(nc, len(flat) // 2)]
        X1 = S1[:, i1 + 1]                                                       # (C1, K, h)  :189-191
        X1 = X1[:, None].expand(X1.shape[0], F, X1.shape[1], X1.shape[2])        # repeat over f :192
        X2 = S2[:, i2 + 1]                                                       # (C, F, K, h) :195-197
        cur = torch.cat([X1, X2], dim=0)                                         # :199
        for i, (W, b) in enumerate(cc):                                          # Conv3d stack :202
            if i == 0:
                cur = torch.einsum('ock,cfkh->ofh', W, cur)
            else:
                cur = torch.einsum('oc,cfh->ofh', W.reshape(W.shape[0], -1), cur)
            cur = leaky(cur + b[:, None, None], use_leaky)
        return _conv_stack(cur.reshape(-1, cur.shape[-1]), bc, acts_b, use_leaky)   # :205
    outs = [checkpoint(body, S1, S2, corr_idx1[:, s:s + chunk], corr_idx2[:, :, s:s + chunk], *flat, use_reentrant=False)
            for s in range(0, H1, chunk)]
    return torch.cat(outs, dim=1)


def conv1d(x, W, b, act, use_leaky=True):
    y = W.reshape(W.shape[0], -1) @ x + b[:, None]
    return leaky(y, use_leaky) if act else y


def _bcl_params(sd, name):
    convs, i = [], 0
    while True:
        k = '%s.blur_conv.%d.composed_module.0.weight' % (name, i)
        k2 = '%s.blur_conv.%d.weight' % (name, i)
        if k in sd:
            convs.append((sd[k][..., 0], sd[k.replace('weight', 'bias')]))
        elif k2 in sd:
            convs.append((sd[k2][..., 0], sd[k2.replace('weight', 'bias')]))
        else:
            break
        i += 1
    return convs, sd.get(name + '.bias')


def _corr_params(sd, name):
    cc, i = [], 0
    while ('%s.corr_conv.%d.composed_module.0.weight' % (name, i)) in sd:
        k = '%s.corr_conv.%d.composed_module.0.weight' % (name, i)
        cc.append((sd[k][:, :, 0, :, 0], sd[k.replace('weight', 'bias')]))
        i += 1
    bc, _ = _bcl_params(sd, name)
    return cc, bc


def parameters(state_dict, dtype=torch.float64, requires_grad=True):
    """numpy / torch state_dict -> dict of leaf tensors (float entries only; index buffers dropped)."""
    out = {}
    for k, v in state_dict.items():
        t = _t(v, dtype)
        if t.is_floating_point():
            out[k] = t.clone().requires_grad_(requires_grad)
    return out


def lattice(gd, dtype=torch.float64):
    """generated_data (list of dicts of numpy arrays / ints) -> the same with torch tensors."""
    return [{k: (_t(v, dtype) if isinstance(v, np.ndarray) or torch.is_tensor(v) else int(v)) for k, v in d.items()}
            for d in gd]


def hplflownet_forward(sd, pc1, pc2, gd, shallow=False, use_leaky=True, last_relu=False, use_norm=True):
    """models/HPLFlowNet.py:238-430 / models/HPLFlowNet_shallow.py:171-311 from `sd` = parameters(...);
    pc1, pc2 (3, N) tensors, gd = lattice(...).  Returns the flow (3, N)."""
    def stack(x, prefix, n):
        for i in range(n):
            x = conv1d(x, sd['%s.%d.composed_module.0.weight' % (prefix, i)],
                       sd['%s.%d.composed_module.0.bias' % (prefix, i)], True, use_leaky)
        return x
    f1 = stack(pc1, 'conv1', 3)
    f2 = stack(pc2, 'conv1', 3)
    nlev = 5 if shallow else 7
    down1, corrs = [], []
    prev = None
    for L in range(nlev):
        convs, _ = _bcl_params(sd, 'bcn%d' % (L + 1))
        res = []
        for which, f in (('pc1', f1), ('pc2', f2)):
            x = torch.cat([gd[L][which + '_el_minus_gr'], f], dim=0)
            res.append(bilateral_conv_forward(x, convs, None, gd[L][which + '_barycentric'],
                                              gd[L][which + '_lattice_offset'], gd[L][which + '_blur_neighbors'],
                                              None, None, True, False, use_norm, use_leaky, last_relu))
        f1, f2 = res
        down1.append(f1)
        if L >= 2:
            j = L - 1
            cc, bc = _corr_params(sd, 'corr%d' % j)
            c = bilateral_corr_forward(f1, f2, prev, gd[L]['pc1_barycentric'] if prev is not None else None,
                                       gd[L]['pc1_lattice_offset'] if prev is not None else None,
                                       gd[L]['pc1_corr_indices'], gd[L]['pc2_corr_indices'], cc, bc, use_norm, use_leaky,
                                       last_relu)
            if shallow:
                if L + 1 < nlev:
                    c = torch.cat([gd[L + 1]['pc1_el_minus_gr'], c], dim=0)
                c = stack(c, 'corr%d_refine' % j, 3)
            corrs.append(c)
            prev = c
    up = None
    for L in reversed(range(nlev)):
        convs, bias = _bcl_params(sd, 'bcn%d_' % (L + 1))
        if L == nlev - 1:
            x = torch.cat([corrs[-1], down1[L]], dim=0)
        else:
            parts = [gd[L + 1]['pc1_el_minus_gr'], up]
            if L >= 2:
                parts.append(corrs[L - 2])
            parts.append(down1[L])
            x = torch.cat(parts, dim=0)
        up = bilateral_conv_forward(x, convs, bias, None, None, gd[L]['pc1_blur_neighbors'], gd[L]['pc1_barycentric'],
                                    gd[L]['pc1_lattice_offset'], False, True, use_norm, use_leaky, last_relu)
    x = conv1d(up, sd['conv2.composed_module.0.weight'], sd['conv2.composed_module.0.bias'], True, use_leaky)
    x = conv1d(x, sd['conv3.composed_module.0.weight'], sd['conv3.composed_module.0.bias'], True, use_leaky)
    return conv1d(x, sd['conv4.weight'], sd['conv4.bias'], False)


def epe3d_loss(flow, sf):
    """models/epe3d_loss.py:9-10 on (3, N) tensors."""
    return torch.norm(flow - sf, p=2, dim=0).mean()


def model_step(state_dict, pc1, pc2, sf, gd, shallow=False, dtype=torch.float64, use_leaky=True, use_norm=True):
    """Forward + backward of one pair: -> (flow (3, N) numpy, loss float, {param name: gradient numpy})."""
    sd = parameters(state_dict, dtype)
    flow = hplflownet_forward(sd, _t(pc1, dtype), _t(pc2, dtype), lattice(gd, dtype), shallow=shallow, use_leaky=use_leaky,
                              use_norm=use_norm)
    loss = epe3d_loss(flow, _t(sf, dtype))
    loss.backward()
    grads = {k: v.grad.numpy() for k, v in sd.items() if v.grad is not None}
    return flow.detach().numpy(), float(loss.item()), grads
