"""numpy restatement of the reference's bilateral layers (CPU oracle).

TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and bench.py's
`cpu_baseline` leg may import this module; nothing under hplflownet_amd/ does.
The product path is the HIP library (hplflownet_amd/csrc) and fails loudly when
that library is missing.

Parity status: PINNED against golden vectors produced by the reference itself
(tools/make_fixtures.py -> tests/golden/layers_*.npz, model_*.npz), fp32,
tolerance 1e-5 relative (summation order differs from torch's conv kernels).

Layout follows the reference: channel-first (C, N) float32 arrays with the batch
dimension (always 1, /root/reference/README.md:57) dropped; index tables int64.
Every function cites the reference lines it restates.
"""
import numpy as np

LEAKY_RATE = 0.1   # models/module_utils.py:6


def leaky(x, use_leaky=True):
    """models/module_utils.py:14,33,50: LeakyReLU(0.1) if use_leaky else ReLU."""
    return np.where(x > 0, x, (LEAKY_RATE if use_leaky else 0.0) * x).astype(np.float32)


def leaky_grad(y, use_leaky=True):
    return np.where(y > 0, 1.0, LEAKY_RATE if use_leaky else 0.0).astype(np.float32)


# --------------------------------------------------------------------------- a1
def sparse_sum(indices, values, size):
    """models/bilateralNN.py:9-30.  out[idx[j], :] += values[j, :]."""
    out = np.zeros(size, dtype=np.float32)
    np.add.at(out, np.asarray(indices).reshape(-1), values)
    return out


def sparse_sum_backward(indices, grad_output):
    """models/bilateralNN.py:33-40.  grad_values = grad_output[idx]."""
    return grad_output[np.asarray(indices).reshape(-1)]


# --------------------------------------------------------------------------- a3
def splat(features, bary, off, H, use_norm=True):
    """models/bilateralNN.py:151-186 (and models/bnn_flow.py:119-151).

    features (C, N), bary (4, N), off (4, N) int  ->  (C, H+1); column 0 is the
    all-zero null vertex (the reference's "+1" trick, :158-162)."""
    C, N = features.shape
    tmp = (bary[None, :, :] * features[:, None, :]).reshape(C, -1).T      # (4N, C), :154-156
    idx = (off + 1).reshape(-1)
    S = sparse_sum(idx, tmp, (H + 1, C)).T                                 # (C, H+1), :162-165
    if use_norm:                                                           # :168-186
        w = sparse_sum(idx, bary.reshape(-1, 1).astype(np.float32), (H + 1, 1))[:, 0]
        S = S * (np.float32(1.0) / (w + np.float32(1e-5)))[None, :]
    return S.astype(np.float32)


def conv_stack(x, convs, acts, use_leaky=True):
    """Sequence of Conv2d((F,1)) / Conv2d((1,1)) (+LeakyReLU) on a gathered input.
    x (C, F, H); convs[i] = (W (O, C_i, F_i), b (O,)); returns (O_last, H) and the
    list of per-layer outputs (for backward).  models/bilateralNN.py:94-112, 219."""
    outs = []
    cur = x.reshape(-1, x.shape[-1])                   # (C*F, H), row index c*F + f
    for (W, b), act in zip(convs, acts):
        y = W.reshape(W.shape[0], -1).astype(np.float32) @ cur + b[:, None]
        if act:
            y = leaky(y, use_leaky)
        outs.append(y.astype(np.float32))
        cur = outs[-1]
    return cur, outs


# ------------------------------------------------------------------- a3..a6 fwd
def bilateral_conv_forward(features, convs, bias, in_bary, in_off, blur_nbr, out_bary, out_off,
                           do_splat, do_slice, use_norm=True, use_leaky=True, last_relu=False):
    """models/bilateralNN.py:122-238.  convs: list of (W (O,C,F) , b).  Returns out and a
    cache for backward."""
    H = blur_nbr.shape[-1]
    if do_splat:
        S = splat(features, in_bary, in_off, H, use_norm)
    else:                                                                  # :190-196
        S = np.concatenate([np.zeros((features.shape[0], 1), np.float32), features], axis=1)
    X = S[:, blur_nbr + 1]                                                 # (C, F, H)  :215-217
    acts = [True] * (len(convs) - 1) + [bool(last_relu)]
    Y, outs = conv_stack(X, convs, acts, use_leaky)
    cache = dict(S=S, X=X, outs=outs, acts=acts, H=H)
    if not do_slice:
        return Y, cache
    G = Y[:, out_off]                                                      # (O, 4, N)  :226-228
    out = (out_bary[None] * G).sum(axis=1)                                 # :231
    if bias is not None:
        out = out + bias[:, None]                                          # :235-236
    return out.astype(np.float32), cache


def bilateral_conv_backward(grad_out, cache, features, convs, bias, in_bary, in_off, blur_nbr,
                            out_bary, out_off, do_splat, do_slice, use_norm=True, use_leaky=True):
    """Gradient of bilateral_conv_forward w.r.t. features, conv weights/biases and bias.
    (The reference relies on torch autograd + SparseSum.backward, bilateralNN.py:33-40.)"""
    H = cache['H']
    grads = {}
    if do_slice:
        if bias is not None:
            grads['bias'] = grad_out.sum(axis=1)
        O = grad_out.shape[0]
        gY = np.zeros((O, H), np.float32)
        for r in range(4):
            np.add.at(gY, (slice(None), out_off[r]), grad_out * out_bary[r][None, :])
    else:
        gY = grad_out
    X = cache['X']
    inputs = [X.reshape(-1, X.shape[-1])] + cache['outs'][:-1]
    gconvs = []
    g = gY
    for i in reversed(range(len(convs))):
        if cache['acts'][i]:
            g = g * leaky_grad(cache['outs'][i], use_leaky)
        W, b = convs[i]
        gW = (g @ inputs[i].T).reshape(W.shape)
        gb = g.sum(axis=1)
        gconvs.insert(0, (gW.astype(np.float32), gb.astype(np.float32)))
        g = W.reshape(W.shape[0], -1).T @ g
    gX = g.reshape(X.shape)                                               # (C, F, H)
    C = X.shape[0]
    gS = np.zeros((C, H + 1), np.float32)
    np.add.at(gS, (slice(None), (blur_nbr + 1)), gX)
    if do_splat:
        if use_norm:
            w = sparse_sum((in_off + 1).reshape(-1), in_bary.reshape(-1, 1), (H + 1, 1))[:, 0]
            gS = gS * (np.float32(1.0) / (w + np.float32(1e-5)))[None, :]
        gfeat = np.zeros_like(features)
        for r in range(4):
            gfeat += in_bary[r][None, :] * gS[:, in_off[r] + 1]
    else:
        gfeat = gS[:, 1:]
    grads['features'] = gfeat.astype(np.float32)
    grads['convs'] = gconvs
    return grads


# ------------------------------------------------------------------ a8..a10 fwd
def bilateral_corr_forward(feat1, feat2, prev_corr, bary1, off1, corr_idx1, corr_idx2,
                           corr_convs, blur_convs, use_norm=True, use_leaky=True,
                           last_relu=False, chunk=512):
    """models/bnn_flow.py:96-210.  corr_convs: list of (W (O, Ctot, K) | (O, C, 1), b) for
    the Conv3d stack (first kernel (1, K, 1)); blur_convs: list of (W (O, C, F) | (O,C,1), b).
    Channel order of the Conv3d input is [prev | feat1 | feat2] (:168,199).
    Chunked over vertices (the reference chunks too, :171-208)."""
    H1 = feat1.shape[1]
    z = np.zeros((feat1.shape[0], 1), np.float32)
    S1 = np.concatenate([z, feat1], axis=1)                                # :153-157
    S2 = np.concatenate([z, feat2], axis=1)                                # :159-163
    if prev_corr is not None:                                              # :119-151,164-165
        P = splat(prev_corr, bary1, off1, H1, use_norm)
        S1 = np.concatenate([P, S1], axis=0)
    F = corr_idx2.shape[0]
    outs = []
    acts_b = [True] * (len(blur_convs) - 1) + [bool(last_relu)]
    for s in range(0, H1, chunk):
        e = min(H1, s + chunk)
        X1 = S1[:, corr_idx1[:, s:e] + 1]                                  # (C1, K, h)   :189-191
        X1 = np.broadcast_to(X1[:, None], (X1.shape[0], F) + X1.shape[1:])  # repeat over f :192
        X2 = S2[:, corr_idx2[:, :, s:e] + 1]                               # (C, F, K, h) :195-197
        X = np.concatenate([X1, X2], axis=0)                               # (Ctot, F, K, h) :199
        cur = X
        for i, (W, b) in enumerate(corr_convs):                            # :202
            # Conv3d as one BLAS matmul per displacement tap f: (O, C*K) @ (C*K, h)
            if i == 0:
                Xr = np.ascontiguousarray(cur.transpose(1, 0, 2, 3)).reshape(F, -1, cur.shape[-1])
            else:
                Xr = np.ascontiguousarray(cur.transpose(1, 0, 2))
            y = np.matmul(W.reshape(W.shape[0], -1).astype(np.float32), Xr).transpose(1, 0, 2)   # (O, F, h)
            cur = leaky(y + b[:, None, None], use_leaky)
        y, _ = conv_stack(cur, blur_convs, acts_b, use_leaky)              # :205
        outs.append(y)
    return np.concatenate(outs, axis=1).astype(np.float32)


# ------------------------------------------------------------------ whole model
def conv1d(x, W, b, act, use_leaky=True):
    """models/module_utils.py:9-24 Conv1dReLU with kernel 1 (or bare nn.Conv1d)."""
    y = W.reshape(W.shape[0], -1) @ x + b[:, None]
    return leaky(y, use_leaky) if act else y.astype(np.float32)


def _bcl_params(sd, name):
    convs = []
    i = 0
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
    cc = []
    i = 0
    while ('%s.corr_conv.%d.composed_module.0.weight' % (name, i)) in sd:
        k = '%s.corr_conv.%d.composed_module.0.weight' % (name, i)
        W = sd[k]                                  # (O, C, 1, K, 1)
        cc.append((W[:, :, 0, :, 0], sd[k.replace('weight', 'bias')]))
        i += 1
    bc, _ = _bcl_params(sd, name)
    return cc, bc


def hplflownet_forward(sd, pc1, pc2, gd, shallow=False, use_leaky=True, last_relu=False):
    """Forward of models/HPLFlowNet.py:238-430 (or models/HPLFlowNet_shallow.py:171-311
    when shallow=True) from a numpy state_dict `sd` (keys as in SURVEY.md Appendix C.2, no
    'module.' prefix).  pc1, pc2 (3, N); gd: list of per-level dicts of numpy arrays.
    Level L (0-based) hosts bcn{L+1}, bcn{L+1}_ and, for L >= 2, corr{L-1}."""
    def stack(x, prefix, n):
        for i in range(n):
            x = conv1d(x, sd['%s.%d.composed_module.0.weight' % (prefix, i)],
                       sd['%s.%d.composed_module.0.bias' % (prefix, i)], True, use_leaky)
        return x
    f1 = stack(pc1, 'conv1', 3)
    f2 = stack(pc2, 'conv1', 3)
    nlev = 5 if shallow else 7
    down1, down2, corrs = [], [], []
    prev = None
    for L in range(nlev):
        convs, _ = _bcl_params(sd, 'bcn%d' % (L + 1))
        res = []
        for which, f in (('pc1', f1), ('pc2', f2)):
            x = np.concatenate([gd[L][which + '_el_minus_gr'], f], axis=0)
            y, _ = bilateral_conv_forward(x, convs, None, gd[L][which + '_barycentric'],
                                          gd[L][which + '_lattice_offset'], gd[L][which + '_blur_neighbors'],
                                          None, None, True, False, True, use_leaky, last_relu)
            res.append(y)
        f1, f2 = res
        down1.append(f1)
        down2.append(f2)
        if L >= 2:
            j = L - 1
            cc, bc = _corr_params(sd, 'corr%d' % j)
            c = bilateral_corr_forward(f1, f2, prev,
                                       gd[L]['pc1_barycentric'] if prev is not None else None,
                                       gd[L]['pc1_lattice_offset'] if prev is not None else None,
                                       gd[L]['pc1_corr_indices'], gd[L]['pc2_corr_indices'],
                                       cc, bc, True, use_leaky, last_relu)
            if shallow:      # HPLFlowNet_shallow.py:220,243,266 corr{j}_refine
                if L + 1 < nlev:
                    c = np.concatenate([gd[L + 1]['pc1_el_minus_gr'], c], axis=0)
                c = stack(c, 'corr%d_refine' % j, 3)
            corrs.append(c)
            prev = c
    up = None
    for L in reversed(range(nlev)):                    # bcn7_ ... bcn1_
        convs, bias = _bcl_params(sd, 'bcn%d_' % (L + 1))
        if L == nlev - 1:
            x = np.concatenate([corrs[-1], down1[L]], axis=0)              # HPLFlowNet.py:372
        else:
            parts = [gd[L + 1]['pc1_el_minus_gr'], up]
            if L >= 2:
                parts.append(corrs[L - 2])
            parts.append(down1[L])
            x = np.concatenate(parts, axis=0)                              # :379-423
        up, _ = bilateral_conv_forward(x, convs, bias, None, None, gd[L]['pc1_blur_neighbors'],
                                       gd[L]['pc1_barycentric'], gd[L]['pc1_lattice_offset'],
                                       False, True, True, use_leaky, last_relu)
    x = conv1d(up, sd['conv2.composed_module.0.weight'], sd['conv2.composed_module.0.bias'], True, use_leaky)
    x = conv1d(x, sd['conv3.composed_module.0.weight'], sd['conv3.composed_module.0.bias'], True, use_leaky)
    return conv1d(x, sd['conv4.weight'], sd['conv4.bias'], False)


def epe3d(pred, target):
    """evaluation_utils.py:9-10 / models/epe3d_loss.py:9-10: mean L2 norm over points."""
    return float(np.sqrt(((pred - target) ** 2).sum(axis=0)).mean())
