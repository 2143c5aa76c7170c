"""Tensor-level wrappers and autograd glue over the C ABI (include/hpl_bcl.h).

Everything here is host-side plumbing: shapes are checked, outputs are allocated as
torch tensors, and the HIP library does the work on the current stream.  Activations
are channel-last 2-D float32 tensors `[rows, channels]` with unit channel stride (the
row stride may exceed the channel count: column slices of wider buffers are fine).
"""
import collections
import ctypes
import os

import torch

from . import _lib
from ._lib import GConvDesc, RelayoutJob, check, ptr, stream

ACT_NONE, ACT_LEAKY = 0, 1
LEAKY_RATE = 0.1   # reference: models/module_utils.py:6


def _cl(x, what='activation'):
    if x.dim() != 2 or x.dtype != torch.float32 or not x.is_cuda or (x.shape[1] > 1 and x.stride(1) != 1):
        raise _lib.HplError('%s must be a channel-last 2-D float32 device tensor with unit channel stride, '
                            'got shape %s strides %s dtype %s device %s'
                            % (what, tuple(x.shape), x.stride(), x.dtype, x.device))
    return x


def _ld(x):
    return x.stride(0) if x.shape[0] > 1 else max(x.shape[1], x.stride(0))


# --------------------------------------------------------------------------- tables
def narrow(t):
    """int64 device table (reference wire format) -> contiguous int32."""
    if t.dtype == torch.int32:
        return t.contiguous()
    t = t.contiguous()
    out = torch.empty(t.shape, dtype=torch.int32, device=t.device)
    check(_lib.load().hpl_index_narrow(ptr(t), ptr(out), t.numel(), stream()), 'hpl_index_narrow')
    return out


def corr2_permute(t):
    """pc2_corr_indices [F, K, H] (int64 or int32) -> int32 [K, F*H]."""
    t = t.contiguous()
    F, K, H = t.shape
    out = torch.empty((K, F * H), dtype=torch.int32, device=t.device)
    fn = _lib.load().hpl_corr2_permute if t.dtype == torch.int64 else _lib.load().hpl_corr2_permute32
    check(fn(ptr(t), ptr(out), F, K, H, stream()), 'hpl_corr2_permute')
    return out


class CloudTables(object):
    """Device-resident tables of one cloud at one lattice level.

    bary [4, N] f32, off [4, N] i32 (vertex of each (remainder, point)), H vertices,
    CSR of the splat (csr_ptr/csr_pt/csr_w) and the density normaliser `norm` [H]."""

    def __init__(self, bary, off, H, blur=None):
        self.bary = bary.contiguous().float()
        self.off = narrow(off)
        self.N = int(self.bary.shape[1])
        self.H = int(H)
        self.blur = narrow(blur) if blur is not None else None
        self._csr = None
        self._sym = {}

    def csr(self):
        if self._csr is None and getattr(self, '_csr_src', None) is not None:
            pair_csr, n0, h0 = self._csr_src           # (the pair's ARRAYS, not the pair: no reference cycle cloud <-> pair)
            p_ptr, p_pt, p_w, p_norm = pair_csr
            self._csr = (p_ptr[h0:] - 4 * n0, p_pt[4 * n0:] - n0, p_w[4 * n0:], p_norm[h0:])
        if self._csr is None:
            dev = self.bary.device
            csr_ptr = torch.empty(self.H + 1, dtype=torch.int32, device=dev)
            csr_pt = torch.empty(4 * self.N, dtype=torch.int32, device=dev)
            csr_w = torch.empty(4 * self.N, dtype=torch.float32, device=dev)
            norm = torch.empty(self.H, dtype=torch.float32, device=dev)
            scratch = torch.empty(self.H + 1 + 4 * self.N + 1026, dtype=torch.int32, device=dev)
            check(_lib.load().hpl_csr_build(ptr(self.off), ptr(self.bary), 4 * self.N, self.N, self.H, ptr(csr_ptr),
                                            ptr(csr_pt), ptr(csr_w), ptr(norm), ptr(scratch), stream()),
                  'hpl_csr_build')
            self._csr = (csr_ptr, csr_pt, csr_w, norm)
        return self._csr

    def slice_back(self, g, use_norm):
        """Backward of the splat: g [H, C] -> [N, C] = sum_r bary[r, n] * norm[v] * g[v = off[r, n]]."""
        return slice_raw(g, self.bary, self.off, self.N, vscale=self.csr()[3] if use_norm else None)


class PairTables(object):
    """Both clouds of one lattice level treated as one cloud: points [0,N0) + [N0,N0+N1), vertices
    [0,H0) + [H0,H0+H1).  Only what the splat needs (the pair CSR); a vertex belongs to one cloud,
    so every segment equals the per-cloud CSR and the result rows equal the per-cloud results."""

    def __init__(self, c0, c1):
        self.c0, self.c1 = c0, c1
        self.N = c0.N + c1.N
        self.H = c0.H + c1.H
        self._csr = None

    def csr(self):
        if self._csr is None:
            c0, c1 = self.c0, self.c1
            dev = c0.bary.device
            csr_ptr = torch.empty(self.H + 1, dtype=torch.int32, device=dev)
            csr_pt = torch.empty(4 * self.N, dtype=torch.int32, device=dev)
            csr_w = torch.empty(4 * self.N, dtype=torch.float32, device=dev)
            norm = torch.empty(self.H, dtype=torch.float32, device=dev)
            scratch = torch.empty(self.H + 1 + 4 * self.N + 1026, dtype=torch.int32, device=dev)
            check(_lib.load().hpl_csr_build_pair(ptr(c0.off), ptr(c0.bary), c0.N, c0.H, ptr(c1.off), ptr(c1.bary),
                                                 c1.N, c1.H, ptr(csr_ptr), ptr(csr_pt), ptr(csr_w), ptr(norm),
                                                 ptr(scratch), stream()), 'hpl_csr_build_pair')
            self._csr = (csr_ptr, csr_pt, csr_w, norm)
            # the pair CSR is the two per-cloud CSRs laid end to end: cloud 1's is a set of views
            # (no launch), cloud 2's needs its offsets removed (built on first use)
            if c0._csr is None:
                c0._csr = (csr_ptr[:c0.H + 1], csr_pt[:4 * c0.N], csr_w[:4 * c0.N], norm[:c0.H])
            if c1._csr is None:
                c1._csr_src = (self._csr, c0.N, c0.H)
        return self._csr

    def slice_back(self, g, use_norm):
        """Backward of the pair splat: each cloud's rows from its own vertices (one launch per cloud
        into the two halves of one [N0+N1, C] matrix)."""
        c0, c1 = self.c0, self.c1
        norm = self.csr()[3] if use_norm else None
        out = torch.empty((self.N, g.shape[1]), dtype=torch.float32, device=g.device)
        slice_raw(g[:c0.H], c0.bary, c0.off, c0.N, vscale=norm[:c0.H] if use_norm else None, out=out[:c0.N])
        slice_raw(g[c0.H:], c1.bary, c1.off, c1.N, vscale=norm[c0.H:] if use_norm else None, out=out[c0.N:])
        return out


def tap_order(nbr):
    """int32 [F<=15, M] neighbour table -> int32 [M] permutation sorting the rows by tap-presence mask
    (deterministic: ties by row id)."""
    F, M = nbr.shape
    L = _lib.load()
    perm = torch.empty(M, dtype=torch.int32, device=nbr.device)
    scratch = torch.empty((int(L.hpl_tap_order_scratch_ints(M)) + 1) // 2, dtype=torch.int64, device=nbr.device)
    check(L.hpl_tap_order(ptr(nbr), nbr.stride(0), F, M, ptr(perm), ptr(scratch), stream()), 'hpl_tap_order')
    return perm


TILE_BM = 64        # tile height of the gather-GEMM classes that take a row order (csrc/gconv.hip: 64x128, 64x64)
#: HPL_MATH=f32 keeps every gather-GEMM on the fp32 MFMA; default (f16x2): the wide launches run on the fp16 MFMA with every
#: fp32 operand carried as a scaled fp16 pair, HPL_MATH=bf16x3: on the bf16 MFMA with exact bf16 triples (csrc/gconv3.hip: both
#: fp32-class accuracy; tiles 128 rows high)
MATH = os.environ.get('HPL_MATH', 'f16x2')
SPLIT3 = MATH != 'f32'
SPLIT_PLANES = 3 if MATH == 'bf16x3' else 2
GROUP_TILE_BM = 128 if SPLIT3 else 64
#: weight images narrower than this stay fp32-only (the split kernel takes launches with N >= 256, C >= 32)
SPLIT3_MIN_N, SPLIT3_MIN_C = 256, 32


def split3_maybe(M, C, F, N):
    """csrc/gconv_common.h split3_maybe: can this launch run on the split-operand kernel at all?  (Launches that cannot
    skip the reduction of their operand's magnitude and run on the fp32 MFMA.)"""
    if not (C >= SPLIT3_MIN_C and N >= SPLIT3_MIN_N and M >= 1024 and F <= 15):
        return False
    tiles = -(-M // 128) * -(-N // 256)
    if tiles >= 128:
        return True
    if F == 1:
        return M >= 8192
    if M >= 16384:
        return True
    return N % 256 == 0 and min(256 // tiles, -(-F * C // 32) // 16) >= 2


def tile_index(nbr, perm, BM=TILE_BM):
    """int32 [F<=15, M] table + row order (or None) -> (tile_idx int32 [tiles, F, BM], tile_mask int32 [tiles, 8]):
    the gather indices and tap masks of every BM-row tile, precomputed once per lattice (hpl_tile_index)."""
    F, M = nbr.shape
    tiles = (M + BM - 1) // BM
    idx = torch.empty((tiles, F, BM), dtype=torch.int32, device=nbr.device)
    mask = torch.empty((tiles, 8), dtype=torch.int32, device=nbr.device)
    check(_lib.load().hpl_tile_index(ptr(nbr), nbr.stride(0), F, M, ptr(perm), BM, ptr(idx), ptr(mask), stream()),
          'hpl_tile_index')
    return idx, mask


def tap_lists(nbr):
    """int32 [F, M] table -> (list_m, list_row int32 [F*M], tap_ptr int32 [F+1]): the present vertices of
    every tap and their source rows."""
    F, M = nbr.shape
    lm = torch.empty(F * M, dtype=torch.int32, device=nbr.device)
    lr = torch.empty(F * M, dtype=torch.int32, device=nbr.device)
    tp = torch.empty(F + 1, dtype=torch.int32, device=nbr.device)
    scratch = torch.empty(2 * F * ((M + 1023) // 1024) + 1100, dtype=torch.int32, device=nbr.device)
    check(_lib.load().hpl_tap_lists(ptr(nbr), nbr.stride(0), F, M, ptr(lm), ptr(lr), ptr(tp), ptr(scratch),
                                    stream()), 'hpl_tap_lists')
    return lm, lr, tp


def table_symmetry_flag(nbr):
    """Launch the symmetry check of an int32 [F, M] table; -> int32 device tensor [1] (1 = symmetric).
    No host sync: read several flags back together (DeviceLattice.resolve_symmetry)."""
    F, M = nbr.shape
    flag = torch.ones(1, dtype=torch.int32, device=nbr.device)
    check(_lib.load().hpl_table_symmetric(ptr(nbr), nbr.stride(0), F, M, ptr(flag), stream()), 'hpl_table_symmetric')
    return flag


def table_is_symmetric(nbr):
    """nbr[f, h] = g  =>  nbr[(F - f) % F, g] = h  (SURVEY.md fact 7).  One host sync."""
    F, H = nbr.shape
    if F != 15:
        return False
    if nbr.is_cuda and nbr.dtype == torch.int32 and nbr.stride(1) == 1:
        return bool(table_symmetry_flag(nbr).item())
    f = torch.arange(1, F, device=nbr.device)
    g = nbr[f].long()
    valid = g >= 0
    back = nbr[(F - f)].long().gather(1, g.clamp(min=0))
    hh = torch.arange(H, device=nbr.device)[None, :].expand_as(g)
    ok = ((back == hh) | ~valid).all() & (nbr[0].long() == torch.arange(H, device=nbr.device)).all()
    return bool(ok.item())


# --------------------------------------------------------------------------- raw ops
def splat_raw(feat, csr, H, use_norm=True, out=None):
    feat = _cl(feat)
    csr_ptr, csr_pt, csr_w, norm = csr
    C = feat.shape[1]
    if out is None:
        out = torch.empty((H, C), dtype=torch.float32, device=feat.device)
    _cl(out, 'out')
    check(_lib.load().hpl_splat(ptr(feat), _ld(feat), C, ptr(csr_ptr), ptr(csr_pt), ptr(csr_w),
                                ptr(norm) if use_norm else None, H, ptr(out), _ld(out), stream()), 'hpl_splat')
    return out


def slice_raw(Y, bary, off, N, vscale=None, bias=None, out=None):
    Y = _cl(Y)
    C = Y.shape[1]
    if out is None:
        out = torch.empty((N, C), dtype=torch.float32, device=Y.device)
    _cl(out, 'out')
    check(_lib.load().hpl_slice(ptr(Y), _ld(Y), C, ptr(bary), ptr(off), N, ptr(vscale), ptr(bias), ptr(out),
                                _ld(out), stream()), 'hpl_slice')
    return out


def gather_sum_raw(Z, nbr, M, K, N, col_step, bias=None, res=None, res_mod=0, act=ACT_NONE, slope=LEAKY_RATE, out=None):
    """Y[m, n] = act(bias[n] + res[m % res_mod, n] + sum_k Z[nbr[k, m], k*col_step + n])  (hpl_gather_sum)."""
    Z = _cl(Z)
    if out is None:
        out = torch.empty((M, N), dtype=torch.float32, device=Z.device)
    check(_lib.load().hpl_gather_sum(ptr(Z), _ld(Z), ptr(nbr), nbr.stride(0), M, K, N, col_step, ptr(bias), ptr(res),
                                     _ld(res) if res is not None else 0, res_mod, act, slope, ptr(out), _ld(out), stream()),
          'hpl_gather_sum')
    return out


def table_invert(tbl, H0, F, H1):
    """int32 [K, F*H0] table with values in [0, H1) -> int32 [F, K*H1] inverse (hpl_table_invert)."""
    K = tbl.shape[0]
    inv = torch.empty((F, K * H1), dtype=torch.int32, device=tbl.device)
    check(_lib.load().hpl_table_invert(ptr(tbl), tbl.stride(0), K, H0, F, H1, ptr(inv), stream()), 'hpl_table_invert')
    return inv


def round_up(x, m):
    return (x + m - 1) // m * m


def weight_relayout(W, R, Q, F, sr, sq, sf, base=0, fmap=None):
    """-> Wt [roundup(F*R, 32), roundup(Q, 4)] with Wt[(fmap[f]*R + r), q] = W.flat[base + r*sr + q*sq + f*sf]."""
    k_rows = round_up(F * R, 32)
    ldw = round_up(Q, 4)
    Wt = torch.empty((k_rows, ldw), dtype=torch.float32, device=W.device)
    check(_lib.load().hpl_weight_relayout(ptr(W), base, R, Q, F, sr, sq, sf, ptr(fmap), ptr(Wt), k_rows, ldw,
                                          stream()), 'hpl_weight_relayout')
    return Wt


class SplitW(object):
    """Split image of a weight image: `planes` uint8 [P, k_rows/8 * ldw * 16] in MFMA B-fragment order -- P = 3: bf16 planes
    hi / mid / lo with Wt == hi + mid + lo exactly (hpl_weight_split3); P = 2: fp16 planes hi / lo of Wt * s, s the power of
    two that puts the image's largest magnitude `amax` (float32 [1], device) into [2^14, 2^15) (hpl_weight_split2h)."""
    __slots__ = ('planes', 'amax', 'P')

    def __init__(self, planes, amax, P):
        self.planes, self.amax, self.P = planes, amax, P

    def rows_from(self, k0, ldw):
        """The same image from row k0 (a multiple of 8) on: k-blocks of 8 rows, 16 bytes per column."""
        return SplitW(self.planes[:, (k0 // 8) * ldw * 16:], self.amax, self.P)


def weight_split3(Wt, out=None, planes=None):
    """fp32 weight image [k_rows (multiple of 8), ldw] -> SplitW: the operand of the split-operand gather-GEMM (gconv_raw
    Wt3=...); planes = 2 / 3 (default: SPLIT_PLANES, i.e. HPL_MATH); out: a SplitW of the same image to refresh."""
    k_rows, ldw = Wt.shape
    if k_rows % 8 or not Wt.is_contiguous():
        raise _lib.HplError('weight_split3: image must be contiguous with a multiple of 8 rows, got %s' % (tuple(Wt.shape),))
    P = out.P if out is not None else (planes or SPLIT_PLANES)
    if out is None:
        out = SplitW(torch.empty((P, k_rows // 8 * ldw * 16), dtype=torch.uint8, device=Wt.device),
                     torch.zeros(1, dtype=torch.float32, device=Wt.device) if P == 2 else None, P)
    if P == 2:
        check(_lib.load().hpl_weight_split2h(ptr(Wt), k_rows, ldw, ptr(out.planes), out.planes.stride(0), ptr(out.amax), stream()),
              'hpl_weight_split2h')
    else:
        check(_lib.load().hpl_weight_split3(ptr(Wt), k_rows, ldw, ptr(out.planes), out.planes.stride(0), stream()), 'hpl_weight_split3')
    return out


def amax(X, rows=None, cols=None):
    """float32 [1] on the device: the largest magnitude of X[:rows, :cols] (hpl_amax; the scale of a gather-GEMM's fp16-pair
    operands)."""
    p_, ld, r, c = _mat(X, 'amax input')
    out = torch.empty(1, dtype=torch.float32, device=X.device)
    check(_lib.load().hpl_amax(p_, ld, r if rows is None else rows, c if cols is None else cols, ptr(out), stream()), 'hpl_amax')
    return out


def amax_rows(X, rows=None, cols=None):
    """(float32 [1], int32 [1]) on the device: the largest magnitude of X[:rows, :cols] and its range-guard word -- ~bits of the
    smallest non-zero ROW maximum, 0 for an all-zero block (hpl_amax_rows; hpl_gconv_desc.a_guard)."""
    p_, ld, r, c = _mat(X, 'amax input')
    out = torch.empty(1, dtype=torch.float32, device=X.device)
    guard = torch.empty(1, dtype=torch.int32, device=X.device)
    check(_lib.load().hpl_amax_rows(p_, ld, r if rows is None else rows, c if cols is None else cols, ptr(out), ptr(guard), stream()),
          'hpl_amax_rows')
    return out, guard


def _mat(x, what):
    """(data_ptr, leading dimension, rows, cols) of a channel-last 2-D float32 device matrix, validated with one
    stride() / shape read (this sits on the host's critical path: ~100 calls per forward)."""
    st, sh = x.stride(), x.shape
    if len(st) != 2 or x.dtype is not torch.float32 or not x.is_cuda or (st[1] != 1 and sh[1] > 1):
        raise _lib.HplError('%s must be a channel-last 2-D float32 device tensor with unit channel stride, '
                            'got shape %s strides %s dtype %s device %s' % (what, tuple(sh), st, x.dtype, x.device))
    return x.data_ptr(), (st[0] if sh[0] > 1 else max(sh[1], st[0])), sh[0], sh[1]


def gconv_raw(A, nbr, M, C, F, Wt, N, bias=None, act=ACT_NONE, res=None, res_mod=0, out=None,
              scat=None, scat_c=0, naive=False, slope=LEAKY_RATE, row_perm=None, split_k=True, reg_stride=0, tiles=None,
              out2=None, rows2=0, Wt3=None, y_amax=None, guard=True, y_guard=None, guard_trips=None):
    """Y[m, n] = act(bias[n] + res[m % res_mod, n] + sum_{f,c} A[nbr[f, m], c] * Wt[f*C + c, n]).
    row_perm (int32 [M], from tap_order): processing order of the output rows; results are unchanged.
    nbr None and reg_stride > 0: tap f of row m reads row f*reg_stride + m (no table: the displacement
    filter of the correlation layer, whose taps are the F blocks of H1 virtual vertices).
    out2 / rows2: rows m < rows2 of the result are also stored to the matrix (view) `out2`.
    y_amax (float32 [1], device, cleared by the caller): max(y_amax, largest |Y|) is left there; y_guard (int32 [1], cleared by the
    caller): the range-guard word of Y beside it.  guard: fp16-pair launches take the range guard of A with its largest magnitude
    (one pass: hpl_amax_rows) -- a second pass over the residuals when A has a row 2^18 below its largest entry; guard_trips
    (int32 [1], device): += 1 when this launch took it."""
    d = GConvDesc()
    d.A, d.lda, d.rows_a, a_cols = _mat(A, 'activation')
    if nbr is not None:
        nst, nsh = nbr.stride(), nbr.shape
        if nbr.dtype is not torch.int32 or len(nsh) != 2 or nsh[0] != F or nsh[1] != M or nst[1] != 1:
            raise _lib.HplError('neighbour table must be int32 [F=%d, M=%d] with unit column stride, got %s %s'
                                % (F, M, tuple(nsh), nbr.dtype))
        d.nbr, d.nbr_stride = ptr(nbr), nst[0]
    elif F != 1:
        if reg_stride <= 0 or (F - 1) * reg_stride + M > d.rows_a:
            raise _lib.HplError('F > 1 needs a neighbour table or a regular stride inside A (F=%d stride=%d M=%d '
                                'rows=%d)' % (F, reg_stride, M, d.rows_a))
        d.reg_stride = reg_stride
    d.M, d.C, d.F = M, C, F
    if C > a_cols:
        raise _lib.HplError('C=%d exceeds the %d channels of A' % (C, a_cols))
    wsh = Wt.shape
    if wsh[0] < F * C or wsh[1] < N or not Wt.is_contiguous():
        raise _lib.HplError('Wt %s too small for K=%d N=%d' % (tuple(wsh), F * C, N))
    d.Wt, d.ldw, d.N = ptr(Wt), wsh[1], N
    d.w_rows = min(wsh[0], round_up(F * C, 32))     # rows past the image read as zero
    d.act, d.slope = act, slope
    if Wt3 is not None:           # weight_split3 of the image Wt is a row range of (same first row)
        if Wt3.P == 3:
            d.Wt3, d.wt3_plane_stride, d.wt3_planes = ptr(Wt3.planes), Wt3.planes.stride(0), 3
        elif scat is None and split3_maybe(M, C, F, N):
            # fp16 pairs: the launch scales A by its largest magnitude (csrc/gconv_common.h split3_maybe: launches that cannot
            # qualify skip the reduction and run on the fp32 MFMA)
            if guard:
                a_amax, a_guard = amax_rows(A, cols=C)
                d.a_guard = ptr(a_guard)
                if guard_trips is not None:
                    d.guard_trips = ptr(guard_trips)
            else:
                a_amax = amax(A, cols=C)
            d.Wt3, d.wt3_plane_stride, d.wt3_planes = ptr(Wt3.planes), Wt3.planes.stride(0), 2
            d.a_amax, d.w_amax = ptr(a_amax), ptr(Wt3.amax)
    if bias is not None:
        d.bias = ptr(bias)
    if res is not None:
        d.res, d.ldres, rrows, _ = _mat(res, 'res')
        d.res_mod = res_mod or rrows
    if scat is not None:
        if out is None:
            raise _lib.HplError('scatter epilogue needs a zero-initialised `out`')
        d.scat, d.scat_stride, d.scat_c = ptr(scat), scat.stride(0), scat_c
    elif out is None:
        out = torch.empty((M, N), dtype=torch.float32, device=A.device)
    d.Y, d.ldy, _, _ = _mat(out, 'out')
    if out2 is not None:
        d.Y2, d.ldy2, r2, c2 = _mat(out2, 'out2')
        if scat is not None or rows2 > r2 or c2 < N:
            raise _lib.HplError('second destination: %d rows x %d columns for rows2=%d N=%d' % (r2, c2, rows2, N))
        d.rows2 = rows2
    if y_amax is not None:
        d.y_amax = ptr(y_amax)
        if y_guard is not None:
            d.y_guard = ptr(y_guard)
    if row_perm is not None:
        if row_perm.dtype is not torch.int32 or row_perm.numel() != M or not row_perm.is_contiguous():
            raise _lib.HplError('row_perm must be a contiguous int32 tensor of M=%d entries' % M)
        d.row_perm = ptr(row_perm)
        if tiles is not None:         # (tile_idx, tile_mask) of THIS table and row order (tile_index)
            d.tile_idx, d.tile_mask, d.tile_bm = ptr(tiles[0]), ptr(tiles[1]), tiles[0].shape[2]
        if CLOCK_PROBE is not None and N > 64 and M >= 16384:
            d.clock_probe = CLOCK_PROBE.data_ptr()
    st = stream()
    if split_k and scat is None and M * N <= _SPLITK_MAX_ELEMS:
        ws = _splitk_workspace(A.device, st)
        d.ws, d.ws_bytes = ws.data_ptr(), ws.numel() * 4
    lib = _lib.load()
    rc = (lib.hpl_gconv_forward_naive if naive else lib.hpl_gconv_forward)(ctypes.byref(d), st)
    if rc != 0:
        check(rc, 'hpl_gconv_forward')
    return out


#: diagnostic (bench.py): a zeroed int64 device tensor (>= 2 entries); sampled workgroups of the wide row-ordered
#: launches add their residence in shader cycles to [0] and in 100 MHz wall ticks to [1] -- the clock the chip
#: sustains under that kernel
CLOCK_PROBE = None

_SPLITK_MAX_ELEMS = 8 << 20          # split-K is offered for outputs of <= 8 M elements (the launch picks the count)
_SPLITK_WS = {}


def _splitk_workspace(device, st):
    """Per-(device, stream) scratch for split-K partial tiles: 16 M floats = 64 MB (csrc/executor.hip: same size, so that both
    issue paths pick the same split counts: a launch fits its splits to the scratch)."""
    key = (device, st)
    ws = _SPLITK_WS.get(key)
    if ws is None:
        ws = _SPLITK_WS[key] = torch.empty(16 << 20, dtype=torch.float32, device=device)
    return ws


def wgrad_raw(A, nbr, M, C, F, dY, N, taps=None, want_bias=False, reg_stride=0):
    """-> dWt [roundup(F*C,32), roundup(N,4)] = sum_m A[nbr[f,m], c] * dY[m, n]  (and, with want_bias,
    the bias gradient sum_m dY[m, :] from the same launch).
    taps = tap_lists(nbr): sum over the present vertices of each tap only (wide layers)."""
    A, dY = _cl(A), _cl(dY, 'dY')
    kp, ldw = round_up(F * C, 32), round_up(N, 4)
    buf = torch.zeros(kp * ldw + (ldw if want_bias else 0), dtype=torch.float32, device=A.device)
    dWt = buf[:kp * ldw].view(kp, ldw)
    gb = buf[kp * ldw:kp * ldw + N] if want_bias else None
    tl, tr, tp = taps if (taps is not None and nbr is not None) else (None, None, None)
    # wide layers (csrc/wgrad3.hip's test): fp16-pair operands need the largest magnitudes of both
    sa = sd = None
    if SPLIT_PLANES == 2 and SPLIT3 and N >= 256 and C >= 128 and M >= 8192 and (tl is not None or (F == 1 and nbr is None)):
        sa, sd = amax(A, cols=C), amax(dY, rows=M, cols=N)
    check(_lib.load().hpl_gconv_wgrad_scaled(ptr(A), _ld(A), A.shape[0], ptr(nbr), nbr.stride(0) if nbr is not None else 0,
                                             reg_stride if nbr is None else 0, M, C, F, ptr(dY), _ld(dY), N, ptr(dWt), ldw,
                                             ptr(tl), ptr(tr), ptr(tp),
                                             M if tl is not None else 0, ptr(gb), ptr(sa), ptr(sd), stream()),
          'hpl_gconv_wgrad')
    return (dWt, gb) if want_bias else dWt


def colsum(X):
    X = _cl(X)
    out = torch.empty(X.shape[1], dtype=torch.float32, device=X.device)
    check(_lib.load().hpl_colsum(ptr(X), _ld(X), X.shape[0], X.shape[1], ptr(out), stream()), 'hpl_colsum')
    return out


def leaky_bwd(dY, Y, slope=LEAKY_RATE, amax=None):
    """dX = dY * (Y > 0 ? 1 : slope); amax (float32 [1], cleared by the caller): max(amax, largest |dX|) is left there."""
    dY, Y = _cl(dY, 'dY'), _cl(Y, 'Y')
    dX = torch.empty(tuple(Y.shape), dtype=torch.float32, device=Y.device)
    check(_lib.load().hpl_leaky_bwd_amax(ptr(dY), _ld(dY), ptr(Y), _ld(Y), slope, ptr(dX), _ld(dX), Y.shape[0],
                                         Y.shape[1], ptr(amax), stream()), 'hpl_leaky_bwd')
    return dX


# --------------------------------------------------------------------------- autograd
class SplatFn(torch.autograd.Function):
    """splat + density normalisation (models/bilateralNN.py:151-186); backward is a slice
    weighted by the normaliser (SparseSum.backward, bilateralNN.py:33-40)."""

    @staticmethod
    def forward(ctx, feat, cloud, use_norm):
        ctx.cloud, ctx.use_norm = cloud, use_norm
        return splat_raw(feat, cloud.csr(), cloud.H, use_norm)

    @staticmethod
    def backward(ctx, g):
        g = g if g.stride(1) == 1 else g.contiguous()
        return ctx.cloud.slice_back(g, ctx.use_norm), None, None


class SliceFn(torch.autograd.Function):
    """slice + bias (models/bilateralNN.py:223-238); backward w.r.t. Y is an un-normalised splat."""

    @staticmethod
    def forward(ctx, Y, cloud, bias):
        ctx.cloud = cloud
        ctx.has_bias = bias is not None
        return slice_raw(Y, cloud.bary, cloud.off, cloud.N, bias=bias)

    @staticmethod
    def backward(ctx, g):
        c = ctx.cloud
        g = g if g.stride(1) == 1 else g.contiguous()
        gY = splat_raw(g, c.csr(), c.H, use_norm=False)
        gb = colsum(g) if ctx.has_bias else None
        return gY, None, gb


_MIRROR = {}


def _mirror_map(F, device):
    # tap f of the forward table is tap (F - f) % F seen from the neighbour
    key = (F, str(device))
    if key not in _MIRROR:
        _MIRROR[key] = ((F - torch.arange(F, device=device)) % F).to(torch.int32)
    return _MIRROR[key]


class WeightBank(object):
    """Training re-lays every conv weight after each optimiser step, once for the forward and once for
    the data gradient: ~160 small launches per step.  The bank records those requests during the
    first step and from then on refreshes ALL images with one launch (`hpl_weight_relayout_batch`)
    at the start of a step; a lookup is served from the bank while the parameter's version still is
    the one the refresh saw, otherwise the caller re-lays that weight on its own."""

    def __init__(self):
        self.jobs = collections.OrderedDict()     # key -> [weight, args, offset, elems, version]
        self.buf = None
        self.dev_jobs = self.dev_prefix = None
        self.total = 0
        self.dirty = False                        # jobs added since the device tables were built

    @staticmethod
    def _key(weight, R, Q, F, sr, sq, sf, base, mirror):
        return (weight.data_ptr(), tuple(weight.shape), R, Q, F, sr, sq, sf, base, mirror)

    def get(self, weight, R, Q, F, sr, sq, sf, base=0, mirror=False):
        """The [roundup(F*R,32), roundup(Q,4)] image of `weight`, from the bank when fresh."""
        key = self._key(weight, R, Q, F, sr, sq, sf, base, mirror)
        job = self.jobs.get(key)
        k_rows, ldw = round_up(F * R, 32), round_up(Q, 4)
        if job is not None and not self.dirty and job[4] == weight._version and job[0] is weight:
            img = self.buf[job[2]:job[2] + job[3]].view(k_rows, ldw)
            img._hpl_bank_job = job             # split3_of keeps the image's split planes with the job
            return img
        if job is None:
            self.register(weight, R, Q, F, sr, sq, sf, base, mirror)
        fmap = _mirror_map(F, weight.device) if mirror else None
        return weight_relayout(weight, R, Q, F, sr, sq, sf, base=base, fmap=fmap)

    def register(self, weight, R, Q, F, sr, sq, sf, base=0, mirror=0):
        """Record an image without producing it (the native plans: every image exists after the next refresh()).
        mirror 0 / 1: Wt[(f*R + r), q] (taps as stored / in the order (F - f) % F); mirror 2: taps as column blocks,
        Wt[r, f*Q + q] (hpl_relayout_job).  -> the job (offset in job[2] after refresh, image shape (job[6], job[7]))."""
        key = self._key(weight, R, Q, F, sr, sq, sf, base, mirror)
        job = self.jobs.get(key)
        if job is None:
            k_rows, ldw = (round_up(R, 32), round_up(F * Q, 4)) if mirror == 2 else (round_up(F * R, 32), round_up(Q, 4))
            job = self.jobs[key] = [weight, (R, Q, F, sr, sq, sf, base, mirror), None, k_rows * ldw, -1, None, k_rows, ldw]
            self.dirty = True
        return job

    def refresh(self, lo=0, hi=None):
        """One launch: every recorded image (or the images of jobs [lo, hi) in registration order) from the current parameter values."""
        if not self.jobs:
            return
        dev = next(iter(self.jobs.values()))[0].device
        if self.dirty or self.buf is None:
            arr = (RelayoutJob * len(self.jobs))()
            prefix = [0]
            for i, job in enumerate(self.jobs.values()):
                w, (R, Q, F, sr, sq, sf, base, mirror) = job[0], job[1]
                arr[i].W, arr[i].base, arr[i].sr, arr[i].sq, arr[i].sf = w.data_ptr(), base, sr, sq, sf
                arr[i].R, arr[i].Q, arr[i].F, arr[i].mirror = R, Q, F, int(mirror)
                arr[i].ldw = round_up(F * Q, 4) if int(mirror) == 2 else round_up(Q, 4)
                job[2] = prefix[-1]
                prefix.append(prefix[-1] + job[3])
            self.total = prefix[-1]
            self.prefix_host = prefix
            raw = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8)
            self.dev_jobs = raw.to(dev)
            self.dev_prefix = torch.tensor(prefix, dtype=torch.int64, device=dev)
            self.buf = torch.empty(self.total, dtype=torch.float32, device=dev)
            self.dirty = False
        hi = len(self.jobs) if hi is None else hi
        if hi <= lo:
            return
        check(_lib.load().hpl_weight_relayout_batch(self.dev_jobs.data_ptr() + lo * ctypes.sizeof(RelayoutJob), hi - lo,
                                                    self.dev_prefix.data_ptr() + 8 * lo, self.prefix_host[hi] - self.prefix_host[lo],
                                                    ptr(self.buf), stream()), 'hpl_weight_relayout_batch')
        for job in list(self.jobs.values())[lo:hi]:
            job[4] = job[0]._version


#: bank used by the autograd path when a training loop calls BANK.refresh() at the start of its steps
BANK = None


def enable_weight_bank(on=True):
    """Switch the batched weight re-layout of the training path on / off (off: one launch per use)."""
    global BANK
    BANK = WeightBank() if on else None
    return BANK


def _train_relayout(weight, R, Q, F, sr, sq, sf, base=0, mirror=False):
    if BANK is not None:
        return BANK.get(weight, R, Q, F, sr, sq, sf, base, mirror)
    fmap = _mirror_map(F, weight.device) if mirror else None
    return weight_relayout(weight, R, Q, F, sr, sq, sf, base=base, fmap=fmap)


MAX_TAPS_PER_PASS = 15      # hpl_gconv_forward: F <= 15 (LDS-staged index table of a tile)


def gconv_passes(A, nbr, M, C, F, Wt, N, groups=None, bias=None, act=ACT_NONE, res=None, res_mod=0, out=None,
                 slope=LEAKY_RATE, row_perm=None, reg_stride=0, tiles=None, guard=True):
    """gconv_raw, run as one pass per tap group when `groups` = [(f0, f1, perm), ...] is given: pass i
    contracts taps [f0, f1) (rows f0*C.. of Wt, rows f0.. of the table) in its own row order and adds
    to the output of the passes before it; bias / residual enter the first pass, the activation the last."""
    regular = nbr is None and reg_stride > 0
    if not groups and (nbr is not None or regular) and F > MAX_TAPS_PER_PASS:
        # radius-2 stencils (65 taps): the kernel stages the indices of at most 15 taps per tile, so the
        # contraction runs as ceil(F / 15) accumulating passes over consecutive tap ranges
        groups = [(f0, min(F, f0 + MAX_TAPS_PER_PASS), None) for f0 in range(0, F, MAX_TAPS_PER_PASS)]
    # wide stencil layers also carry the split image of their weights: the kernel takes the launch when it is big enough
    W3 = split3_of(Wt) if (SPLIT3 and (nbr is not None or F == 1) and N >= SPLIT3_MIN_N and C >= SPLIT3_MIN_C) else None
    if not groups or not (nbr is not None or regular) or len(groups) < 2:
        return gconv_raw(A, nbr, M, C, F, Wt, N, bias=bias, act=act, res=res, res_mod=res_mod, out=out, slope=slope,
                         row_perm=row_perm, reg_stride=reg_stride, tiles=tiles if not isinstance(tiles, list) else None,
                         Wt3=W3, guard=guard)
    y = out
    gt = tiles if isinstance(tiles, list) and len(tiles) == len(groups) else [None] * len(groups)
    for i, (f0, f1, perm) in enumerate(groups):
        first, last = i == 0, i == len(groups) - 1
        # the same rows of the split image (k-blocks of 8 rows, 16 bytes per column)
        w3 = W3.rows_from(f0 * C, Wt.shape[1]) if (W3 is not None and (f0 * C) % 8 == 0) else None
        # a tap range of a regular pattern is the same pattern over the rows from f0*reg_stride on
        y = gconv_raw(A[f0 * reg_stride:] if regular else A, None if regular else nbr[f0:f1], M, C, f1 - f0,
                      Wt[f0 * C:], N, bias=bias if first else None,
                      act=act if last else ACT_NONE, res=res if first else y, res_mod=res_mod if first else 0,
                      out=y, slope=slope, row_perm=perm, reg_stride=reg_stride, tiles=gt[i] if perm is not None else None,
                      Wt3=w3, guard=guard)
    return y


def _split3_entry(Wt, version):
    ev = torch.cuda.Event()
    w3 = weight_split3(Wt)
    ev.record()                       # behind k_weight_split3: a consumer on another stream waits for THIS, not for the re-layout
    return [w3, version, ev, stream()]


def _split3_wait(entry):
    ev = entry[2]
    if ev is not None and stream() != entry[3]:
        if ev.query():
            entry[2] = None           # seen complete once: visible to every later launch on any stream
        else:
            torch.cuda.current_stream().wait_event(ev)
    return entry[0]


def split3_of(Wt):
    """The split image of a weight image, made once per image: images handed out by a WeightBank keep it with the bank's
    job (the bank returns a fresh view of its buffer on every lookup, so an attribute on the view would never be found
    again; valid while the job's parameter version is the one its last refresh saw), other images carry it as an
    attribute keyed on their version.  Either way the entry holds the event recorded behind the split kernel: a forward
    on another stream that finds the entry waits for the split, not just for the re-layout (_cached_relayout's event)."""
    job = getattr(Wt, '_hpl_bank_job', None)
    if job is not None:
        ent = job[5]
        if ent is None or ent[1] != job[4]:
            ent = job[5] = _split3_entry(Wt, job[4])
        return _split3_wait(ent)
    ent = getattr(Wt, '_hpl_split3', None)
    if ent is None or ent[1] != Wt._version:
        ent = Wt._hpl_split3 = _split3_entry(Wt, Wt._version)
    return _split3_wait(ent)


class GConvFn(torch.autograd.Function):
    """Gathered convolution Y = act(b + sum_f W_f . A[nbr[f]]) with the conv weight in its torch
    layout `weight.view(O, Ctot, F)`; channels [c0, c0+C) of the weight are used.

    bwd_mode: 'mirror' (symmetric table over the same vertex set: gather with mirrored taps),
              'scatter' (any table: fp32 atomics), 'dense' (no table)."""

    @staticmethod
    def forward(ctx, A, weight, bias, nbr, M, c0, C, F, act, res, res_mod, bwd_mode, slope, row_perm=None,
                taps=None, groups=None, reg_stride=0, tiles=None):
        O = weight.shape[0]
        Ctot = weight.numel() // (O * F)
        Wt = _train_relayout(weight, C, O, F, F, Ctot * F, 1, base=c0 * F)
        Y = gconv_passes(A, nbr, M, C, F, Wt, O, groups, bias=bias, act=act, res=res, res_mod=res_mod, slope=slope,
                         row_perm=row_perm, reg_stride=reg_stride, tiles=tiles)
        ctx.reg_stride = reg_stride
        ctx.tiles = tiles            # the mirror backward gathers through the same table in the same row orders
        ctx.groups = groups          # the mirror backward gathers through the same table: same groups
        ctx.slope = slope
        ctx.row_perm = row_perm      # same table in the mirror backward -> same tap masks -> same order
        ctx.taps = taps
        ctx.save_for_backward(A, weight, nbr, Y if act != ACT_NONE else None)
        ctx.cfg = (M, c0, C, F, act, res is not None, res_mod, bwd_mode, bias is not None, Ctot)
        return Y

    @staticmethod
    def backward(ctx, g):
        A, weight, nbr, Y = ctx.saved_tensors
        M, c0, C, F, act, has_res, res_mod, bwd_mode, has_bias, Ctot = ctx.cfg
        O = weight.shape[0]
        g = g if (g.dim() == 2 and g.stride(1) == 1) else g.contiguous()
        if act == ACT_LEAKY:
            g = leaky_bwd(g, Y, ctx.slope)
        gA = gW = gb = gres = None
        if ctx.needs_input_grad[0]:
            rows = A.shape[0]
            if bwd_mode == 'dense':
                WtT = _train_relayout(weight, O, C, 1, Ctot * F, F, 1, base=c0 * F)
                # wide 1x1 layers: the data gradient is a dense GEMM of the same class as their forward (split operands)
                W3 = split3_of(WtT) if (SPLIT3 and C >= SPLIT3_MIN_N and O >= SPLIT3_MIN_C) else None
                gA_c = gconv_raw(g, None, M, O, 1, WtT, C, Wt3=W3, guard=False)       # (gradients run unguarded: HPL_FLAG_NOGUARD)
            elif bwd_mode == 'mirror':
                if rows != M:
                    raise _lib.HplError('mirror backward needs a table over the same vertex set')
                WtT = _train_relayout(weight, O, C, F, Ctot * F, F, 1, base=c0 * F, mirror=True)
                gA_c = gconv_passes(g, nbr, M, O, F, WtT, C, ctx.groups, row_perm=ctx.row_perm, tiles=ctx.tiles, guard=False)
            else:   # scatter: G[m, (f, c)] = g[m] . W[:, c, f], added into row nbr[f, m]
                # columns ordered (f, c): source element (o, f*C + c) = W[o, c0 + c, f]
                Wcols = weight.view(O, Ctot, F)[:, c0:c0 + C, :].permute(0, 2, 1).reshape(O, F * C).contiguous()
                WtS = weight_relayout(Wcols, O, F * C, 1, F * C, 1, 1)
                if bwd_mode == 'regular':
                    # source rows f*stride + m are all distinct: block f of G IS the gradient of rows
                    # [f*stride, f*stride + M) -- a plain GEMM and one strided copy, no atomics
                    G = gconv_raw(g, None, M, O, 1, WtS, F * C)
                    rs = ctx.reg_stride
                    if rs == M and rows == F * M:
                        gA_c = G.view(M, F, C).transpose(0, 1).reshape(F * M, C)
                    else:
                        gA_c = torch.zeros((rows, C), dtype=torch.float32, device=A.device)
                        for f in range(F):
                            gA_c[f * rs:f * rs + M] = G[:, f * C:(f + 1) * C]
                else:
                    gA_c = torch.zeros((rows, C), dtype=torch.float32, device=A.device)
                    gconv_raw(g, None, M, O, 1, WtS, F * C, out=gA_c, scat=nbr, scat_c=C)
            if C == A.shape[1]:
                gA = gA_c
            else:
                gA = torch.zeros_like(A)
                gA[:, :C] = gA_c
        want_gb = has_bias and ctx.needs_input_grad[2]
        if ctx.needs_input_grad[1]:
            dWt = wgrad_raw(A, nbr, M, C, F, g, O, taps=ctx.taps, want_bias=want_gb, reg_stride=ctx.reg_stride)
            if want_gb:
                dWt, gb = dWt
            # the un-layout writes every element of the channel range: zeros only for partial ranges
            gW = torch.empty_like(weight) if (c0 == 0 and C == Ctot) else torch.zeros_like(weight)
            check(_lib.load().hpl_weight_unlayout(ptr(dWt), dWt.shape[1], C, O, F, ptr(gW), c0 * F, F, Ctot * F, 1,
                                                  0, stream()), 'hpl_weight_unlayout')
        elif want_gb:
            gb = colsum(g)
        if has_res and ctx.needs_input_grad[9]:
            if res_mod and res_mod != M:
                gres = g.view(M // res_mod, res_mod, O).sum(dim=0)
            else:
                gres = g
        return gA, gW, gb, None, None, None, None, None, None, gres, None, None, None, None, None, None, None, None


def _cols_image(weight, C, O, F, Ctot, c0):
    """[roundup(C, 32), roundup(F*O, 4)] image with element (c, f*O + o) = weight[o, c0 + c, f] ("taps as column blocks",
    hpl_relayout_job.mirror == 2): from the training bank when it holds a fresh one, else made here."""
    if BANK is not None:
        job = BANK.register(weight, C, O, F, F, Ctot * F, 1, c0 * F, 2)
        if not BANK.dirty and job[4] == weight._version and job[0] is weight:
            return BANK.buf[job[2]:job[2] + job[3]].view(job[6], job[7])
    img = torch.zeros((round_up(C, 32), round_up(F * O, 4)), dtype=torch.float32, device=weight.device)
    img[:C, :F * O] = weight.detach().reshape(O, Ctot, F)[:, c0:c0 + C, :].permute(1, 2, 0).reshape(C, F * O)
    return img


class CorrPc2Fn(torch.autograd.Function):
    """The pc2 half of the patch correlation (models/bnn_flow.py:195-202) over the F*H0 virtual vertices:
    P[f*H0 + h] = act(bias + res[h] + sum_k W_k . f2[corr2[k][f*H0 + h]]), computed as a projection of every pc2 vertex per tap
    (one dense GEMM) and a gather-sum (hpl_gather_sum); the backward gathers through the inverse table (no atomics)."""

    @staticmethod
    def forward(ctx, f2, weight, bias, res, table, H0, F, K, c0, C, slope):
        O = weight.shape[0]
        Ctot = weight.numel() // (O * K)
        Hv = f2.shape[0]
        Z = gconv_raw(f2, None, Hv, C, 1, _cols_image(weight, C, O, K, Ctot, c0), K * O)
        Y = gather_sum_raw(Z, table.t, F * H0, K, O, O, bias=bias, res=res, res_mod=H0, act=ACT_LEAKY, slope=slope)
        ctx.table, ctx.cfg = table, (H0, F, K, c0, C, O, Ctot, Hv, slope)
        ctx.save_for_backward(f2, weight, Y)
        return Y

    @staticmethod
    def backward(ctx, g):
        f2, weight, Y = ctx.saved_tensors
        H0, F, K, c0, C, O, Ctot, Hv, slope = ctx.cfg
        g = g if (g.dim() == 2 and g.stride(1) == 1) else g.contiguous()
        g = leaky_bwd(g, Y, slope)
        gres = g.view(F, H0, O).sum(dim=0) if ctx.needs_input_grad[3] else None
        gb = colsum(g) if ctx.needs_input_grad[2] else None
        gf2 = gW = None
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[1]:
            inv = ctx.table.inverse(H0, F, Hv)
            dZ = gather_sum_raw(g, inv, K * Hv, F, O, 0).view(Hv, K * O)
            if ctx.needs_input_grad[0]:
                WzT = _train_relayout(weight, O, C, K, Ctot * K, K, 1, base=c0 * K)          # [(k*O + o), c]
                gf2 = gconv_raw(dZ, None, Hv, K * O, 1, WzT, C)
            if ctx.needs_input_grad[1]:
                dWz = wgrad_raw(f2, None, Hv, C, 1, dZ, K * O)
                gW = torch.zeros_like(weight)
                gW.view(O, Ctot, K)[:, c0:c0 + C, :] = dWz[:C, :K * O].view(C, K, O).permute(2, 0, 1)
        return gf2, gW, gb, gres, None, None, None, None, None, None, None


def corr_pc2(f2, weight, bias, res, table, H0, F, K, c0, C, slope):
    """-> [F*H0, O]; `table`: bcl.NbrTable of the permuted pc2_corr_indices [K, F*H0]."""
    if torch.is_grad_enabled() and (f2.requires_grad or weight.requires_grad or (res is not None and res.requires_grad)):
        return CorrPc2Fn.apply(f2, weight, bias, res, table, H0, F, K, c0, C, slope)
    O = weight.shape[0]
    Ctot = weight.numel() // (O * K)
    key = (id(weight), c0, C, 'cols')
    hit = _WT_CACHE.get(key)
    if hit is None or hit[2] != weight._version or hit[1] is not weight or hit[3] != weight.data_ptr():
        img = _cols_image(weight, C, O, K, Ctot, c0)
        ev = torch.cuda.Event()
        ev.record()
        hit = _WT_CACHE[key] = [img, weight, weight._version, weight.data_ptr(), ev, stream()]
    elif hit[4] is not None and stream() != hit[5]:
        if hit[4].query():
            hit[4] = None
        else:
            torch.cuda.current_stream().wait_event(hit[4])
    Z = gconv_raw(f2, None, f2.shape[0], C, 1, hit[0], K * O)
    return gather_sum_raw(Z, table.t, F * H0, K, O, O, bias=bias, res=res, res_mod=H0, act=ACT_LEAKY, slope=slope)


def gconv(A, weight, bias, nbr, M, F, act=ACT_NONE, c0=0, C=None, res=None, res_mod=0, bwd_mode='scatter',
          out=None, slope=LEAKY_RATE, row_perm=None, taps=None, tap_groups=None, reg_stride=0, tiles=None):
    """Autograd-aware gathered convolution; with grad disabled it can write into `out`.

    tap_groups: [(f0, f1, perm), ...] -- the contraction is run as one pass per group of
    consecutive taps, each with the rows sorted by the group's own (short) tap mask, the passes
    accumulating into the output (bias in the first, activation in the last).  A 5-bit mask leaves
    ~32 row classes, so 64-row tiles are nearly pure and absent taps are skipped almost exactly
    (43 % instead of 59 % of the slices executed on bcn1_, 72 % instead of 83 % on bcn2_)."""
    O = weight.shape[0]
    Ctot = weight.numel() // (O * F)
    C = Ctot if C is None else C
    if torch.is_grad_enabled() and (A.requires_grad or weight.requires_grad or
                                    (res is not None and res.requires_grad)):
        y = GConvFn.apply(A, weight, bias, nbr, M, c0, C, F, act, res, res_mod, bwd_mode, slope, row_perm,
                          taps() if callable(taps) else taps,
                          tap_groups() if callable(tap_groups) else tap_groups, reg_stride,
                          tiles() if callable(tiles) else tiles)
        if out is not None:
            out.copy_(y)
            return out
        return y
    Wt = _cached_relayout(weight, C, O, F, Ctot, c0)
    groups = tap_groups() if callable(tap_groups) else tap_groups
    return gconv_passes(A, nbr, M, C, F, Wt, O, groups, bias=bias, act=act, res=res, res_mod=res_mod, out=out,
                        slope=slope, row_perm=row_perm, reg_stride=reg_stride, tiles=tiles() if callable(tiles) else tiles)


_WT_CACHE = collections.OrderedDict()
_WT_CACHE_MAX = 512


def invalidate_weight_cache():
    """Drop every cached weight image of the inference path.  The cache keys on the parameter's identity,
    version counter and data pointer, which writes through `param.data` (`p.data.copy_(...)`, reference-style
    `model.apply(init)` with `m.weight.data`) do NOT change: call this after editing weights that way.
    (In-place writes through the parameter under torch.no_grad(), load_state_dict and optimiser steps bump the
    version and need nothing.)  The native forward plans (plan.ForwardPlan) key on the epoch bumped here as well: they
    refresh their weight images and combined biases at their next use."""
    global _WEIGHT_EPOCH
    _WT_CACHE.clear()
    _WEIGHT_EPOCH += 1


_WEIGHT_EPOCH = 0


def weight_epoch():
    return _WEIGHT_EPOCH


def _cached_relayout(weight, C, O, F, Ctot, c0):
    """Inference path: the k-major weight image only changes when the parameter does (tensor identity +
    version counter); the entry keeps the parameter alive, so its id cannot be recycled while cached.
    An image is produced on one stream and may be consumed on others (bench.py alternates forwards over
    several): the entry carries the event recorded behind its re-layout kernel, and a consumer on another
    stream waits for it until it has been seen complete once."""
    key = (id(weight), c0, C)
    hit = _WT_CACHE.get(key)
    if hit is not None and hit[2] == weight._version and hit[1] is weight and hit[3] == weight.data_ptr():
        ev = hit[4]
        if ev is not None:
            st = stream()
            if st != hit[5]:
                if ev.query():
                    hit[4] = None
                else:
                    torch.cuda.current_stream().wait_event(ev)
        return hit[0]
    Wt = weight_relayout(weight.detach(), C, O, F, F, Ctot * F, 1, base=c0 * F)
    ev = torch.cuda.Event()
    ev.record()
    _WT_CACHE[key] = [Wt, weight, weight._version, weight.data_ptr(), ev, stream()]     # (.data swaps keep id and version)
    _WT_CACHE.move_to_end(key)
    if len(_WT_CACHE) > _WT_CACHE_MAX:
        _WT_CACHE.popitem(last=False)
    return Wt
