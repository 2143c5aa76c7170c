"""Native forward plans: a whole inference forward of HPLFlowNet / HPLFlowNetShallow in ONE C call.

`ForwardPlan(model)` writes, once per model, the program the native executor walks
(include/hpl_bcl.h "Native forward executor", csrc/executor.hip): the wiring of
/root/reference/models/HPLFlowNet.py:238-430 (HPLFlowNet_shallow.py:171-311) as a flat list of operations over
symbolic row counts -- exactly the launches flownet._FlowNetBase.forward issues on its pair-batched inference path
(same kernels, same arguments, hence bit-identical flows; tests/test_gpu_plan.py), with every weight image and
bias resolved to a device pointer.  `plan(pc1, pc2, lattice)` then costs one ctypes call + the launches
themselves instead of ~130 Python round trips (1.4 ms -> ~0.4 ms of host time per pair).

Only device-built lattices (hplflownet_amd.lattice) are accepted; anything else -- the reference's wire format,
autograd -- takes the Python path of the model.
"""
import ctypes
import os
import weakref

import time

import torch

from . import _lib, ops
from ._lib import Buf, LevelTables, Op, Ref, Weight, check, ptr, stream
from .bcl import GROUPS_MIN_CHANNELS, _ConvReLU, _conv_of, _slope
from .flownet import DeviceLattice, PairBlur

OP_GCONV, OP_SPLAT, OP_SLICE, OP_COPY, OP_LOAD = 1, 2, 3, 4, 5
OP_GSUM, OP_INVERT = 16, 17
TBL_NONE, TBL_BLUR_PAIR, TBL_BLUR0, TBL_CORR1, TBL_CORR2, TBL_REGULAR, TBL_CSR_PAIR, TBL_CSR_C0, TBL_CLOUD0 = range(9)
ORD_NONE, ORD_PERM, ORD_GROUPS = 0, 1, 2
SYM_ZERO, SYM_N0, SYM_N1, SYM_NP, SYM_LEVEL0 = -1, 0, 1, 2, 8
S_H0, S_H1, S_HP, S_FH0, S_IN0, S_INP, S_FH1 = range(7)
BUF_OUT = -2

#: profiling classes (hpl_op.tag): the wide stencil convs that dominate the step, everything else
TAG_OTHER, TAG_WIDE_BLUR = 0, 1


#: seconds the caller's thread spent waiting for a free workspace slot (bench.py reports it)
WAIT = {'s': 0.0}

def lsym(L, k):
    return SYM_LEVEL0 + 8 * L + k


class _R(object):
    """A column slice of a program buffer (python mirror of hpl_ref)."""
    __slots__ = ('buf', 'row_off', 'rows', 'col_off', 'cols')

    def __init__(self, buf, cols, col_off=0, row_off=SYM_ZERO, rows=SYM_ZERO):
        self.buf, self.cols, self.col_off, self.row_off, self.rows = buf, cols, col_off, row_off, rows

    def c(self):
        return Ref(self.buf, self.row_off, self.rows, self.col_off, self.cols)

    def columns(self, off, n):
        return _R(self.buf, n, self.col_off + off, self.row_off, self.rows)

    def rows_from(self, off_sym, rows_sym):
        return _R(self.buf, self.cols, self.col_off, off_sym, rows_sym)


_NONE = Ref(-1, SYM_ZERO, SYM_ZERO, 0, 0)


def _op(*fields, **kw):
    """hpl_op with the optional second destination (default: none)."""
    o = Op(*fields)
    o.out2 = kw.get('out2') or _NONE
    o.rows2_sym = kw.get('rows2', SYM_ZERO)
    o.b = kw.get('b') or _NONE
    o.flags, o.aux = kw.get('flags', 0), kw.get('aux', 0)
    return o


class _Program(object):
    def __init__(self, bank):
        self.ops, self.bufs, self.bank = [], [], bank
        self.cond = (0, 0)          # (HPL_COND_*, level) stamped on the ops emitted next
        self.weights = []           # (weight param, R, Q, F, sr, sq, sf, base[, mirror]) bank images; ('grad', i) gradient images
        self.biases = []            # tensors (kept alive; combined biases are refreshed in place)
        self.combined = []          # (tensor, (param a, param b))
        self.meta = []              # per op: what train_plan needs to write its gradient (None: no gradient flows through it)
        self.wmeta = {}             # weight index -> (param, C, O, F, Ctot, c0)

    def buf(self, rows_sym, cols):
        self.bufs.append((rows_sym, cols))
        return _R(len(self.bufs) - 1, cols)

    def weight(self, w, C, O, F, Ctot, c0):
        # the image ops._cached_relayout makes: Wt[(f*C + c), o] = W[o, c0 + c, f]
        key = (w, C, O, F, F, Ctot * F, 1, c0 * F)
        self.bank.register(w.detach(), C, O, F, F, Ctot * F, 1, base=c0 * F)
        self.weights.append(key)
        self.wmeta[len(self.weights) - 1] = (w, C, O, F, Ctot, c0)
        return len(self.weights) - 1

    def weight_cols(self, w, C, O, F, Ctot, c0):
        # "taps as column blocks": Wt[c, f*O + o] = W[o, c0 + c, f]  (the per-tap projection of the patch correlation)
        self.bank.register(w.detach(), C, O, F, F, Ctot * F, 1, c0 * F, 2)
        self.weights.append((w, C, O, F, F, Ctot * F, 1, c0 * F, 2))
        self.wmeta[len(self.weights) - 1] = (w, C, O, F, Ctot, c0)
        return len(self.weights) - 1

    def bias(self, b):
        if b is None:
            return -1
        self.biases.append(b)
        return len(self.biases) - 1

    def bias_sum(self, a, b):
        """conv bias + layer bias of the slice-before-1x1 order (bcl.BilateralConvFlex.forward_cl)."""
        if a is None or b is None:
            return self.bias(a if b is None else b)
        t = (a.detach() + b.detach()).contiguous()
        self.combined.append((t, (a, b)))
        return self.bias(t)

    def gconv(self, a, out, M, C, N, wid, bias=-1, act=0, slope=0.1, F=1, level=0, table=TBL_NONE, order=ORD_NONE,
              res=None, res_mod=SYM_ZERO, reg_stride=SYM_ZERO, tag=TAG_OTHER, out2=None, rows2=SYM_ZERO, flags=0, aux=0,
              wcols=False):
        """out2 / rows2: the first `rows2` rows of the result are written to a second view as well (a layer output that
        feeds two concatenation buffers is stored by its producer, not copied)."""
        self.ops.append(_op(OP_GCONV, tag, a.c(), out.c(), res.c() if res is not None else _NONE, M, res_mod, level, table,
                            order, F, C, N, wid, bias, act, slope, 0, reg_stride, 0, *self.cond,
                            out2=out2.c() if out2 is not None else None, rows2=rows2, flags=flags, aux=aux))
        self.meta.append(dict(kind='gconv', a=a, out=out, res=res, M=M, C=C, N=N, wid=wid, bias=bias, act=act, slope=slope, F=F,
                              level=level, table=table, order=order, res_mod=res_mod, reg_stride=reg_stride, out2=out2,
                              rows2=rows2, cond=self.cond, tag=tag, wcols=wcols))

    def gsum(self, a, out, M, K, N, level, bias=-1, act=0, slope=0.1, res=None, res_mod=SYM_ZERO):
        """hpl_gather_sum through the level's pc2 correlation table: out[m] = act(bias + res[m % res_mod] + sum_k a[corr2[k][m], k*N:])."""
        self.ops.append(_op(OP_GSUM, TAG_OTHER, a.c(), out.c(), res.c() if res is not None else _NONE, M, res_mod, level, TBL_CORR2,
                            0, K, N, N, -1, bias, act, slope, 0, SYM_ZERO, 0, *self.cond))
        self.meta.append(dict(kind='gsum', a=a, out=out, res=res, res_mod=res_mod, M=M, K=K, N=N, level=level, bias=bias, act=act,
                              slope=slope, cond=self.cond))

    def splat(self, a, out, level, table, H, C, use_norm, flags=0):
        self.ops.append(_op(OP_SPLAT, TAG_OTHER, a.c(), out.c(), _NONE, H, SYM_ZERO, level, table, 0, 1, C, C, -1, -1, 0, 0.0,
                           int(bool(use_norm)), SYM_ZERO, 0, *self.cond, flags=flags))
        self.meta.append(dict(kind='splat', a=a, out=out, level=level, table=table, H=H, C=C, use_norm=use_norm, cond=self.cond))

    def slice(self, a, out, level, N, C, bias=-1):
        self.ops.append(_op(OP_SLICE, TAG_OTHER, a.c(), out.c(), _NONE, N, SYM_ZERO, level, TBL_CLOUD0, 0, 1, C, C, -1, bias, 0,
                           0.0, 0, SYM_ZERO, 0, *self.cond))
        self.meta.append(dict(kind='slice', a=a, out=out, level=level, N=N, C=C, bias=bias, cond=self.cond))

    def copy(self, a, out, rows, C, level=0, flags=0):
        """a None: el_minus_gr of `level` (both clouds, point-major)."""
        self.ops.append(_op(OP_COPY, TAG_OTHER, a.c() if a is not None else _NONE, out.c(), _NONE, rows, SYM_ZERO, level, 0, 0,
                           1, C, C, -1, -1, 0, 0.0, 0, SYM_ZERO, 0, *self.cond, flags=flags))
        self.meta.append(None)

    def load(self, out, ext, rows):
        self.ops.append(_op(OP_LOAD, TAG_OTHER, _NONE, out.c(), _NONE, rows, SYM_ZERO, 0, 0, 0, 1, 3, 3, -1, -1, 0, 0.0, 0,
                           SYM_ZERO, ext, 0, 0))
        self.meta.append(None)

    def raw(self, kind, a=None, out=None, b=None, M=SYM_ZERO, level=0, table=TBL_NONE, order=ORD_NONE, F=1, C=0, N=0, weight=-1, bias=-1,
            slope=0.0, use_norm=0, reg_stride=SYM_ZERO, flags=0, aux=0):
        """An op of the backward program (train_plan): fields as hpl_op, cond = the current one."""
        self.ops.append(_op(kind, TAG_OTHER, a.c() if a is not None else _NONE, out.c() if out is not None else _NONE, _NONE, M, SYM_ZERO,
                            level, table, order, F, C, N, weight, bias, 0, slope, int(bool(use_norm)), reg_stride, 0, *self.cond,
                            b=b.c() if b is not None else None, flags=flags, aux=aux))
        self.meta.append(None)


def _dense(P, x, conv, act, slope, M, out=None, rows_sym=None):
    """pointwise conv (bcl.pointwise_conv): [M, C_in] -> [M, C_out]."""
    O, C = conv.weight.shape[0], conv.weight.shape[1]
    if out is None:
        out = P.buf(rows_sym, O)
    P.gconv(x, out, M, C, O, P.weight(conv.weight, C, O, 1, C, 0), bias=P.bias(conv.bias), act=1 if act else 0,
            slope=slope)
    return out


def _conv_stack(P, x, mods, M, rows_sym, F, level, table, order, slope, out=None, reg_stride=SYM_ZERO, wide_tag=False,
                out2=None, rows2=SYM_ZERO):
    """bcl._run_conv_stack: first conv through `table` (F taps), the rest 1x1; returns the output ref.  out2 / rows2:
    second destination of the LAST conv (_Program.gconv)."""
    n = len(mods)
    for i, m in enumerate(mods):
        conv = _conv_of(m)
        act = 1 if isinstance(m, _ConvReLU) else 0
        O = conv.weight.shape[0]
        final = i == n - 1
        o = out if (final and out is not None) else P.buf(rows_sym, O)
        second = dict(out2=out2, rows2=rows2) if (final and out2 is not None) else {}
        if i == 0:
            Ctot = conv.weight.numel() // (O * F)
            ordr = order
            if table == TBL_BLUR0:        # Up conv: tap-group passes for wide layers, else the single-pass order
                ordr = ORD_GROUPS if conv.in_channels >= GROUPS_MIN_CHANNELS else ORD_PERM
            P.gconv(x, o, M, Ctot, O, P.weight(conv.weight, Ctot, O, F, Ctot, 0), bias=P.bias(conv.bias), act=act,
                    slope=slope, F=F, level=level, table=table, order=ordr, reg_stride=reg_stride,
                    tag=TAG_WIDE_BLUR if (wide_tag and conv.in_channels >= GROUPS_MIN_CHANNELS) else TAG_OTHER, **second)
        else:
            C = conv.weight.shape[1]
            P.gconv(x, o, M, C, O, P.weight(conv.weight, C, O, 1, C, 0), bias=P.bias(conv.bias), act=act, slope=slope,
                    **second)
        x = o
    return x


def _corr_width(model, L):
    """channels of the (refined) correlation living at level L (flownet._FlowNetBase.__init__: corr_dim)."""
    return 64 if model.REFINE else getattr(model, 'corr%d' % (L - 1)).num_output[-1]


def build_program(model, bank):
    """The pair-batched inference forward of flownet._FlowNetBase.forward, op by op.  The input matrix of every Up
    layer (the reference's torch.cat of el_minus_gr | upper Up output | correlation | Down features) is allocated up
    front and every part is WRITTEN THERE BY ITS PRODUCER: the Up layer above and the correlation layer store straight
    into their columns, the Down layer's last conv stores its cloud-1 rows there as a second destination, the
    el_minus_gr columns are filled by one batched launch (csrc/executor.hip) -- no copy launches."""
    P = _Program(bank)
    nlev = model.NLEV
    sl = _slope(model.use_leaky)
    # ---- the Up layers' input matrices: xb[L] = [H0(L), parts], cols[L] = {part: (column offset, width)}
    xb, cols = {}, {}
    up_w = None
    for L in reversed(range(nlev)):
        layer = getattr(model, 'bcn%d_' % (L + 1))
        parts = []
        if L < nlev - 1:
            parts += [('emg', 4), ('up', up_w)]
        if L >= 2:
            parts.append(('corr', _corr_width(model, L)))
        parts.append(('down', getattr(model, 'bcn%d' % (L + 1)).num_output[-1]))
        if L == nlev - 1:
            parts = [q for q in parts if q[0] == 'corr'] + [q for q in parts if q[0] == 'down']
        col, cols[L] = 0, {}
        for kind, w in parts:
            cols[L][kind] = (col, w)
            col += w
        assert col == layer.num_input, (L, col, layer.num_input)
        xb[L] = P.buf(lsym(L, S_H0), col)
        up_w = layer.num_output[-1]

    def part(L, kind):
        return xb[L].columns(*cols[L][kind])
    # ---- inputs, conv1 (both clouds stacked)
    xin = P.buf(SYM_NP, 3)
    P.load(xin.rows_from(SYM_ZERO, SYM_N0), 0, SYM_N0)
    P.load(xin.rows_from(SYM_N0, SYM_N1), 1, SYM_N1)
    feat_c = model.conv1[-1].conv.out_channels
    x = P.buf(SYM_NP, 4 + feat_c)
    t = xin
    for i, m in enumerate(model.conv1):
        last = i == len(model.conv1) - 1
        t = _dense(P, t, m.conv, True, sl, SYM_NP, out=x.columns(4, feat_c) if last else None, rows_sym=SYM_NP)
    prev = None           # (ref, channels) of the previous correlation output, rows = H0 of its level
    for L in range(nlev):
        layer = getattr(model, 'bcn%d' % (L + 1))
        HP, H0, H1, INP = lsym(L, S_HP), lsym(L, S_H0), lsym(L, S_H1), lsym(L, S_INP)
        cin = layer.num_input
        P.copy(None, x.columns(0, 4), INP, 4, level=L)                      # x[:, :4] = el_minus_gr of the pair
        s = P.buf(HP, cin)
        P.splat(x, s, L, TBL_CSR_PAIR, HP, cin, layer.use_norm)
        c_out = layer.num_output[-1]
        nxt = P.buf(HP, 4 + c_out) if L + 1 < nlev else None
        # both clouds' vertices -> the next level's splat input; cloud 1's rows also -> the Up layer's input matrix
        y = _conv_stack(P, s, list(layer.blur_conv), HP, HP, layer.filter_size, L, TBL_BLUR_PAIR, ORD_PERM, sl,
                        out=nxt.columns(4, c_out) if nxt is not None else None, out2=part(L, 'down'), rows2=H0)
        f1, f2 = y.rows_from(SYM_ZERO, H0), y.rows_from(H0, H1)
        x = nxt
        if L >= 2:
            prev = _corr(P, model, L, f1, f2, prev, sl, part(L, 'corr'))
    # ---- Up path: every layer writes into the 'up' columns of the level below
    for L in reversed(range(nlev)):
        layer = getattr(model, 'bcn%d_' % (L + 1))
        if L < nlev - 1:
            P.copy(None, part(L, 'emg'), lsym(L, S_H0), 4, level=L + 1)     # el_minus_gr of cloud 1 at level L+1
        if L > 0:
            _up_layer(P, (layer, L), xb[L], part(L - 1, 'up'), sl)
    # the last Up layer writes a fresh [N0, HEAD_IN] matrix
    layer = getattr(model, 'bcn1_')
    ybuf = P.buf(SYM_N0, layer.num_output[-1])
    _up_layer(P, (layer, 0), xb[0], ybuf, sl)
    y = _dense(P, ybuf, model.conv2.conv, True, sl, SYM_N0, rows_sym=SYM_N0)
    y = _dense(P, y, model.conv3.conv, True, sl, SYM_N0, rows_sym=SYM_N0)
    _dense(P, y, model.conv4, False, sl, SYM_N0, out=_R(BUF_OUT, 3))
    return P


def _up_layer(P, layer_L, xb, out, sl):
    """bcl.BilateralConvFlex.forward_cl for an Up layer (no splat, slice).  When the last conv is a bias-only 1x1
    and the slice shrinks the row count (input points < vertices: decided per pair, HPL_COND_SHRINK) the 1x1 conv
    runs AFTER the slice on the sliced rows; otherwise the full stack, then slice + bias."""
    layer, L = layer_L[0], layer_L[1]
    H0, IN0 = lsym(L, S_H0), lsym(L, S_IN0)
    mods = list(layer.blur_conv)
    bias = layer.bias if (layer.use_bias and layer.do_slice) else None
    wide = L <= 1                      # profiling class of the two big stencil convs (bcn1_, bcn2_)
    reorder = len(mods) >= 2 and not isinstance(mods[-1], _ConvReLU)
    if reorder:
        P.cond = (1, L)
        y = _conv_stack(P, xb, mods[:-1], H0, H0, layer.filter_size, L, TBL_BLUR0, ORD_PERM, sl, wide_tag=wide)
        conv = mods[-1]
        C1 = conv.weight.shape[1]
        z = P.buf(IN0, C1)
        P.slice(y, z, L, IN0, C1)
        O = conv.weight.shape[0]
        P.gconv(z, out, IN0, C1, O, P.weight(conv.weight, C1, O, 1, C1, 0), bias=P.bias_sum(conv.bias, bias), act=0, slope=0.1)
        P.cond = (2, L)
    y = _conv_stack(P, xb, mods, H0, H0, layer.filter_size, L, TBL_BLUR0, ORD_PERM, sl, wide_tag=wide)
    P.slice(y, out, L, IN0, layer.num_output[-1], bias=P.bias(bias))
    P.cond = (0, 0)


def _corr(P, model, L, f1, f2, prev, sl, dst):
    """bcl.BilateralCorrelationFlex.forward_cl (+ the shallow model's refine stack), its result written into `dst`
    (the correlation columns of the Up layer's input matrix): -> (ref, channels)."""
    j = L - 1
    m = getattr(model, 'corr%d' % j)
    H0, FH0, IN0 = lsym(L, S_H0), lsym(L, S_FH0), lsym(L, S_IN0)
    C, Pd, K, F = m.num_input, m.prev_corr_dim, m.corr_size, m.filter_size
    conv0 = m.corr_conv[0].conv
    w0 = conv0.weight
    O = w0.shape[0]
    Ctot = Pd + 2 * C
    csl = _slope(m.use_leaky)
    f1r, f2r = f1, f2
    a = P.buf(H0, O)
    P.gconv(f1r, a, H0, C, O, P.weight(w0, C, O, K, Ctot, Pd), F=K, level=L, table=TBL_CORR1, order=ORD_PERM, slope=0.1)
    if prev is not None:
        ps = P.buf(H0, Pd)
        P.splat(prev[0], ps, L, TBL_CSR_C0, H0, Pd, m.use_norm)
        a2 = P.buf(H0, O)
        P.gconv(ps, a2, H0, Pd, O, P.weight(w0, Pd, O, K, Ctot, 0), F=K, level=L, table=TBL_CORR1, order=ORD_PERM,
                res=a, res_mod=H0, slope=0.1)
        a = a2
    p = P.buf(FH0, O)
    if O % 4 == 0 and K == 15 and F == 15:
        # every pc2 vertex projected once per correlation tap (dense GEMM 64 -> K*O), then a gather-sum of O-float rows
        z = P.buf(lsym(L, S_H1), K * O)
        P.gconv(f2r, z, lsym(L, S_H1), C, K * O, P.weight_cols(w0, C, O, K, Ctot, Pd + C), wcols=True)
        P.gsum(z, p, FH0, K, O, L, bias=P.bias(conv0.bias), act=1, slope=csl, res=a, res_mod=H0)
    else:
        P.gconv(f2r, p, FH0, C, O, P.weight(w0, C, O, K, Ctot, Pd + C), bias=P.bias(conv0.bias), act=1, slope=csl, F=K, level=L,
                table=TBL_CORR2, res=a, res_mod=H0)
    for mm in list(m.corr_conv)[1:]:
        p = _dense(P, p, mm.conv, True, csl, FH0, rows_sym=FH0)
    width = m.num_output[-1]
    if not model.REFINE:
        c = _conv_stack(P, p, list(m.blur_conv), H0, H0, F, L, TBL_REGULAR, ORD_NONE, csl, out=dst, reg_stride=H0)
        return (c, width)
    if L + 1 < model.NLEV:
        cb = P.buf(H0, 4 + width)
        P.copy(None, cb.columns(0, 4), H0, 4, level=L + 1)
        c = _conv_stack(P, p, list(m.blur_conv), H0, H0, F, L, TBL_REGULAR, ORD_NONE, csl, out=cb.columns(4, width),
                        reg_stride=H0)
        c = cb
    else:
        c = _conv_stack(P, p, list(m.blur_conv), H0, H0, F, L, TBL_REGULAR, ORD_NONE, csl, reg_stride=H0)
    mods = list(getattr(model, 'corr%d_refine' % j))
    for i, mm in enumerate(mods):
        c = _dense(P, c, mm.conv, True, sl, H0, out=dst if i == len(mods) - 1 else None, rows_sym=H0)
    return (c, mods[-1].conv.out_channels)


def level_tables(lat, hint):
    """ctypes array of hpl_level_tables for a device-built lattice (cached on the lattice)."""
    cached = getattr(lat, '_native_tables', None)
    if cached is not None:
        return cached
    n = len(lat.levels)
    arr = (LevelTables * n)()
    keep = []
    for L, lv in enumerate(lat.levels):
        t = arr[L]
        c0, c1 = lv.clouds
        t.n0, t.n1, t.H0, t.H1 = c0.N, c1.N, lv.H[0], lv.H[1]
        t.emg_pair = lv.emg_pair.data_ptr()
        cp, cpt, cw, cn = lv.pair.csr()
        t.csr_ptr, t.csr_pt, t.csr_w, t.csr_norm = cp.data_ptr(), cpt.data_ptr(), cw.data_ptr(), cn.data_ptr()
        t.bary0, t.off0 = c0.bary.data_ptr(), c0.off.data_ptr()
        pb = lv.blur.pair
        t.blur, t.blur_stride = pb.t.data_ptr(), pb.t.stride(0)
        t.tile_bm = ops.TILE_BM
        t.group_tile_bm = ops.GROUP_TILE_BM
        perm = pb.perm
        if perm is not None:
            t.blur_perm = perm.data_ptr()
            if pb.perm_tiles is not None:
                t.blur_perm_tidx, t.blur_perm_tmask = [x.data_ptr() for x in pb.perm_tiles]
        up = lv.blur[0]
        wide = hint[L] if isinstance(hint, (list, tuple)) else hint
        groups = up.groups() if wide else None
        if groups:
            t.n_up_groups = len(groups)
            tiles = up.group_tiles()
            for g, (f0, f1, pm) in enumerate(groups):
                t.up_group_cut[g], t.up_group_cut[g + 1] = f0, f1
                t.up_group_perm[g] = pm.data_ptr()
                if tiles is not None:
                    t.up_group_tidx[g], t.up_group_tmask[g] = tiles[g][0].data_ptr(), tiles[g][1].data_ptr()
                keep.append(pm)
        else:
            t.n_up_groups = 0
            pm = up.perm
            if pm is not None:
                t.up_perm = pm.data_ptr()
                if up.perm_tiles is not None:
                    t.up_perm_tidx, t.up_perm_tmask = [x.data_ptr() for x in up.perm_tiles]
        if lv.corr1 is not None:
            t.corr1, t.corr1_stride = lv.corr1.t.data_ptr(), lv.corr1.t.stride(0)
            pm = lv.corr1.perm
            if pm is not None:
                t.corr1_perm = pm.data_ptr()
                if lv.corr1.perm_tiles is not None:
                    t.corr1_perm_tidx, t.corr1_perm_tmask = [x.data_ptr() for x in lv.corr1.perm_tiles]
            t.corr2 = lv.corr2.t.data_ptr()
    lat._native_tables = (arr, n, keep)
    return lat._native_tables


class ForwardPlan(object):
    def __init__(self, model):
        # The plan is the VALUE of flownet._PLANS (a WeakKeyDictionary keyed by the model): it must not keep its own key
        # alive, or every model that ever ran a native forward would stay resident together with its weight images and
        # workspaces.  The model is held weakly; the parameters are reached through it.
        self._model = weakref.ref(model)
        self.NLEV = model.NLEV
        self.epoch = ops.weight_epoch()
        self.bank = ops.WeightBank()
        self._images_ready = None       # event recorded behind the last write of the weight images / combined biases
        self._images_stream = None
        with torch.no_grad():
            self.prog = self._program(model)
            self.bank.refresh()
        self._mark_images_written()
        self._sig = self._signature()
        L = _lib.load()
        P = self.prog
        ops_arr = (Op * len(P.ops))(*P.ops)
        bufs_arr = (Buf * len(P.bufs))(*[Buf(r, c) for r, c in P.bufs])
        w_arr = (Weight * len(P.weights))()
        # weights of the wide tap-group convs also exist as split images (csrc/gconv3.hip: bf16 MFMA, fp32-exact operands)
        wide = set(o.weight for o in P.ops if o.kind == OP_GCONV and o.table in (TBL_NONE, TBL_BLUR_PAIR, TBL_BLUR0, TBL_CORR1, TBL_CORR2)
                   and not (o.flags & 2) and o.N >= ops.SPLIT3_MIN_N and o.C >= ops.SPLIT3_MIN_C) if ops.SPLIT3 else set()
        self._split3 = {}           # weight index -> (image view, split image)
        done = {}                   # bank job -> split image (several weight indices may name one image)
        for i, key in enumerate(P.weights):
            if key[0] == 'grad':                 # gradient image of a training plan: (pointer, rows, ldw) from _grad_image
                w_arr[i].Wt, w_arr[i].rows, w_arr[i].ldw = self._grad_image(key[1])
                continue
            mirror = key[8] if len(key) > 8 else 0
            job = self.bank.jobs[self.bank._key(key[0], *key[1:8], mirror)]
            k_rows, ldw = job[6], job[7]
            w_arr[i].Wt = self.bank.buf.data_ptr() + 4 * job[2]
            w_arr[i].ldw, w_arr[i].rows = ldw, k_rows
            if i in wide and mirror != 2:
                if id(job) not in done:
                    img = self.bank.buf[job[2]:job[2] + k_rows * ldw].view(k_rows, ldw)
                    done[id(job)] = (img, ops.weight_split3(img))
                    self._split3[i] = done[id(job)]
                w3 = done[id(job)][1]
                w_arr[i].Wt3, w_arr[i].wt3_plane_stride, w_arr[i].wt3_planes = w3.planes.data_ptr(), w3.planes.stride(0), w3.P
                if w3.P == 2:
                    w_arr[i].w_amax = w3.amax.data_ptr()
        self._mark_images_written()
        b_arr = (ctypes.c_void_p * max(1, len(P.biases)))(*[(b if isinstance(b, int) else b.data_ptr()) for b in P.biases])
        self.handle = L.hpl_plan_create(ops_arr, len(P.ops), bufs_arr, len(P.bufs), w_arr, len(P.weights), b_arr,
                                        len(P.biases))
        if not self.handle:
            raise _lib.HplError('hpl_plan_create failed: %s' % L.hpl_last_error().decode())
        self._lib = L
        self.hint = model.lattice_hint()
        self._ws = {}            # workspaces, one per (stream, slot)
        self._slot = 0
        self.slots = 4          # forwards in flight before a workspace is reused (bench: 3 forward streams)
        self._fence = {}

    def _program(self, model):
        return build_program(model, self.bank)

    def __del__(self):
        try:
            if getattr(self, 'handle', None):
                self._lib.hpl_plan_destroy(self.handle)
            self._ws.clear()
        except Exception:
            pass

    @property
    def model(self):
        return self._model()

    def _signature(self):
        m = self._model()
        if m is None:
            return None
        return (self.epoch,) + tuple((p.data_ptr(), p._version) for p in m.parameters())

    def _mark_images_written(self):
        """The weight images and combined biases were just (re)written on the current stream: forwards on OTHER
        streams must not read them before that work has run (bench.py alternates forwards over three streams)."""
        ev = torch.cuda.Event()
        ev.record()
        self._images_ready, self._images_stream = ev, stream()

    def _wait_images(self):
        ev = self._images_ready
        if ev is None:
            return
        if ev.query():
            self._images_ready = None            # seen complete once: visible to every later launch on any stream
        elif stream() != self._images_stream:
            torch.cuda.current_stream().wait_event(ev)

    def fresh(self):
        """False once a parameter was replaced (new storage): the plan holds stale pointers, build a new one."""
        self.epoch = ops.weight_epoch()          # ops.invalidate_weight_cache() bumps it: edits through .data
        sig = self._signature()
        if sig is None:
            return False
        if sig == self._sig:
            return True
        if any(a[0] != b[0] for a, b in zip(sig[1:], self._sig[1:])):
            return False
        with torch.no_grad():                                  # same storage, new values: refresh the images in place
            # forwards still in flight on other streams read the images being overwritten: order the refresh behind them
            cur = torch.cuda.current_stream()
            for ev in self._fence.values():
                if ev is not None and not ev.query():
                    cur.wait_event(ev)
            self.bank.refresh()
            for img, w3 in self._split3.values():
                ops.weight_split3(img, out=w3)
            for t, (a, b) in self.prog.combined:
                t.copy_(a.detach() + b.detach())
        self._mark_images_written()
        self._sig = sig
        return True

    def accepts(self, lat):
        if getattr(lat, 'tables', None) is not None and getattr(lat, 'n_levels', 0) >= self.NLEV:
            return True                                    # lattice.NativeLattice
        if not isinstance(lat, DeviceLattice) or len(lat.levels) < self.NLEV:
            return False
        for lv in lat.levels[:self.NLEV]:
            if lv.pair is None or not isinstance(lv.blur, PairBlur):
                return False
        return True

    def workspace(self, nbytes, dev):
        """Round-robin workspaces: a forward's activations must survive until it has run; the caller's streams are
        ordered by events recorded behind each run."""
        slot = self._slot
        self._slot = (slot + 1) % self.slots
        ev = self._fence.get(slot)
        if ev is not None and not ev.query():
            t = time.perf_counter()
            ev.synchronize()                     # every workspace is still in use: the GPU is the limiter, idle host time
            WAIT['s'] += time.perf_counter() - t
        ws = self._ws.get(slot)
        if ws is None or ws.numel() < nbytes:
            ws = self._ws[slot] = torch.empty(int(nbytes * 1.25), dtype=torch.uint8, device=dev)
        return slot, ws

    def __call__(self, pc1, pc2, lat):
        """pc1, pc2 (1, 3, N) or (3, N) device tensors -> flow (1, 3, N0) (a transposed view of an [N0, 3] matrix)."""
        p1 = pc1[0] if pc1.dim() == 3 else pc1
        p2 = pc2[0] if pc2.dim() == 3 else pc2
        if not (p1.is_contiguous() and p2.is_contiguous() and p1.dtype == torch.float32):
            p1, p2 = p1.contiguous().float(), p2.contiguous().float()
        arr, n, _ = level_tables(lat, self.hint)
        if arr[0].n0 != p1.shape[1] or arr[0].n1 != p2.shape[1]:
            raise _lib.HplError('lattice was built for %d / %d points, got %d / %d'
                                % (arr[0].n0, arr[0].n1, p1.shape[1], p2.shape[1]))
        nlev = self.NLEV
        need = self._lib.hpl_plan_workspace_bytes(self.handle, arr, nlev)
        if need < 0:
            raise _lib.HplError('hpl_plan_workspace_bytes: %s' % self._lib.hpl_last_error().decode())
        slot, ws = self.workspace(need, p1.device)
        self._wait_images()
        out = torch.empty((p1.shape[1], 3), dtype=torch.float32, device=p1.device)
        check(self._lib.hpl_plan_run(self.handle, arr, nlev, ptr(p1), ptr(p2), ptr(out), ws.data_ptr(), ws.numel(),
                                     stream()), 'hpl_plan_run')
        ev = torch.cuda.Event()
        ev.record()
        self._fence[slot] = ev
        return out.t().unsqueeze(0)

    # ---- profiling of the dominant launches (bench.py)
    def profile(self, tag):
        check(self._lib.hpl_plan_profile(self.handle, tag), 'hpl_plan_profile')

    def clock_probe(self, tensor):
        """int64[4] device tensor (or None) the profiled launches stamp their first workgroup's clocks into."""
        self._clk = tensor
        check(self._lib.hpl_plan_clock_probe(self.handle, tensor.data_ptr() if tensor is not None else None),
              'hpl_plan_clock_probe')

    def guard_trips(self):
        """Launches of this plan's runs that took the second pass of the fp16-pair form's range guard (hpl_plan_guard_trips;
        synchronises with the device)."""
        n = ctypes.c_int64(0)
        check(self._lib.hpl_plan_guard_trips(self.handle, ctypes.byref(n)), 'hpl_plan_guard_trips')
        return int(n.value)

    def profile_read(self):
        n, ms = ctypes.c_int(0), ctypes.c_float(0.0)
        check(self._lib.hpl_plan_profile_read(self.handle, ctypes.byref(n), ctypes.byref(ms)), 'hpl_plan_profile_read')
        return n.value, ms.value
