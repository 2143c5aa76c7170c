"""Drop-in bilateral layers on the HIP library.

Mirrors the reference's operator interface for the hot path -- same class names,
constructor arguments, forward signatures, parameter/buffer names (checkpoints load with
strict=True) and error behaviour (exceptions) -- so that
`models/HPLFlowNet.py` can import these instead of its own:

  BilateralConvFlex          /root/reference/models/bilateralNN.py:46-238
  sparse_sum                 /root/reference/models/bilateralNN.py:9-43
  BilateralCorrelationFlex   /root/reference/models/bnn_flow.py:10-210
  Conv{1,2,3}dReLU           /root/reference/models/module_utils.py:9-59

The arithmetic runs in hand-written HIP kernels through include/hpl_bcl.h; the conv
sub-modules below only hold parameters.  Internally everything is channel-last; the
(B=1, C, N) channel-first tensors of the reference API are accepted and returned as
transposed views, so chained layers exchange data without copies.  B must be 1
(reference: README.md:57).  `chunk_size` is accepted and ignored: the gathered operand is
never materialised, so there is nothing to chunk.
"""
import collections

import os

import torch
import torch.nn as nn

from . import _lib, ops
from .ops import ACT_LEAKY, ACT_NONE, LEAKY_RATE

__all__ = ['BilateralConvFlex', 'BilateralCorrelationFlex', 'sparse_sum', 'Conv1dReLU', 'Conv2dReLU',
           'Conv3dReLU', 'LEAKY_RATE']


# ----------------------------------------------------------------------------- parameter holders
class _ConvReLU(nn.Module):
    conv_cls = None

    def __init__(self, in_channels, out_channels, kernel_size=1, stride=1, padding=0, use_leaky=False,
                 bias=True):
        super(_ConvReLU, self).__init__()
        self.in_channels, self.out_channels, self.kernel_size = in_channels, out_channels, kernel_size
        self.use_leaky = use_leaky
        relu = nn.LeakyReLU(LEAKY_RATE, inplace=True) if use_leaky else nn.ReLU(inplace=True)
        self.composed_module = nn.Sequential(
            self.conv_cls(in_channels, out_channels, kernel_size=kernel_size, stride=stride, padding=padding,
                          bias=bias), relu)

    @property
    def conv(self):
        return self.composed_module[0]


class Conv1dReLU(_ConvReLU):
    """Pointwise Conv1d + (Leaky)ReLU; forward runs the dense gather-GEMM (kernel 1 only)."""
    conv_cls = nn.Conv1d

    def forward(self, x):
        return pointwise_conv(x, self.conv, True, self.use_leaky)


class Conv2dReLU(_ConvReLU):
    conv_cls = nn.Conv2d


class Conv3dReLU(_ConvReLU):
    conv_cls = nn.Conv3d


def _slope(use_leaky):
    return LEAKY_RATE if use_leaky else 0.0   # ReLU == LeakyReLU with slope 0


def to_channel_last(x):
    """(1, C, N) reference layout -> [N, C] view/copy with unit channel stride."""
    if x.dim() == 2:
        return x
    if x.dim() != 3 or x.shape[0] != 1:
        raise _lib.HplError('batch size must be 1 (reference README.md:57), got shape %s' % (tuple(x.shape),))
    t = x[0].t()
    if t.shape[1] > 1 and t.stride(1) != 1:
        t = t.contiguous()
    return t


def to_channel_first(y):
    """[N, C] -> (1, C, N) view (no copy)."""
    return y.t().unsqueeze(0)


def pointwise_conv(x, conv, act, use_leaky, out=None):
    """x (1, C, N) or [N, C]; conv: nn.Conv1d/2d/3d with all-ones kernel."""
    cl = x.dim() == 2
    xc = to_channel_last(x)
    y = ops.gconv(xc, conv.weight, conv.bias, None, xc.shape[0], 1,
                  act=ACT_LEAKY if act else ACT_NONE, bwd_mode='dense', out=out, slope=_slope(use_leaky))
    return y if cl else to_channel_first(y)


# ----------------------------------------------------------------------------- table cache
#: the multi-pass contraction pays when a pass is still long: C >= 256 -> >= 40 slices per pass and tile
GROUPS_MIN_CHANNELS = 256




class NbrTable(object):
    """int32 neighbour table [F, M] on the device + lazily checked symmetry."""

    #: smaller tables are not reordered: on the coarse levels > 90 % of the taps are present (measured:
    #: level 2 of the N=8192 frustum skips < 3 % of the slices), the sort costs more than it saves
    PERM_MIN_ROWS = 16384

    def __init__(self, t):
        self.t = t
        self._sym = None
        self._perm = False      # False = not computed yet, None = not applicable
        self._taps = False
        self._groups = False
        self._perm_tiles = False
        self._group_tiles = False
        #: set by the lattice builder: vertices per input point of this level.  A sparse lattice (few
        #: points per vertex -> most neighbour slots empty) is where the multi-pass contraction pays
        #: (bcn1_, 42 % of the taps present: 2.62 -> 2.08 ms; bcn2_, 72 %: no gain)
        self.vertices_per_point = None

    @property
    def perm(self):
        """Row order grouping vertices by tap-presence mask (see gconv row_perm)."""
        if self._perm is False:
            F, M = self.t.shape
            self._perm = ops.tap_order(self.t) if (1 < F <= 15 and M >= self.PERM_MIN_ROWS) else None
        return self._perm

    #: number of tap groups of the multi-pass contraction (ops.gconv tap_groups); env for A/B runs
    #: (hpl_level_tables carries at most 4 groups: larger values are clamped)
    TAP_GROUPS = 2      # swept 1 / 2 / 3 / 5 end to end: 193 / 205 / 204 / 173 pairs/s (round 2); 1 / 3: 316 / 310 vs 353 (round 4)
    #: levels with at least this many lattice vertices per input point get the passes (level 0: 3.2, level 1:
    #: 1.34 -- bcn1_ and bcn2_; measured end to end 2.0 -> 1.2: 212.5 -> 218.5 pairs/s)
    GROUPS_MIN_SPARSITY = 1.2
    #: ... and at least this many rows: the passes run on the split-operand kernel's 128 x 256 tiles, which take a launch that
    #: fills half the CUs in one round or splits over K (csrc/gconv3.hip) -- clouds of 2 048 points: 384 -> 720 pairs/s, 1 024: 620 -> 930
    GROUPS_MIN_ROWS = 2048

    def groups(self):
        """[(f0, f1, perm)] for TAP_GROUPS groups of consecutive taps, each with its own row order
        (tap_order of the sub-table); None when the table is not reordered at all."""
        if self._groups is False:
            F, M = self.t.shape
            G = self.TAP_GROUPS
            sparse = self.vertices_per_point is None or self.vertices_per_point >= self.GROUPS_MIN_SPARSITY
            if G <= 1 or not sparse or not (1 < F <= 15 and M >= self.GROUPS_MIN_ROWS):
                self._groups = None
            else:
                cuts = [round(i * F / G) for i in range(G + 1)]
                self._groups = [(f0, f1, ops.tap_order(self.t[f0:f1])) for f0, f1 in zip(cuts[:-1], cuts[1:])]
        return self._groups

    @property
    def perm_tiles(self):
        """(tile_idx, tile_mask) of the single-pass row order (ops.tile_index), None without one."""
        if self._perm_tiles is False:
            self._perm_tiles = ops.tile_index(self.t, self.perm) if self.perm is not None else None
        return self._perm_tiles

    def group_tiles(self):
        """[(tile_idx, tile_mask)] aligned with groups(), None without groups."""
        if self._group_tiles is False:
            g = self.groups()
            self._group_tiles = [ops.tile_index(self.t[f0:f1], pm, BM=ops.GROUP_TILE_BM) for f0, f1, pm in g] if g else None
        return self._group_tiles

    #: per-tap vertex lists pay for wide layers only (wgrad tap mode needs C >= 128-ish) and big tables
    TAPS_MIN_ROWS = 4096

    def taps(self):
        """(list, tap_ptr) of the present vertices of every tap (ops.tap_lists), built on first use
        (training only: the weight gradient sums over present vertex-tap pairs)."""
        if self._taps is False:
            F, M = self.t.shape
            self._taps = ops.tap_lists(self.t) if (1 < F <= 15 and M >= self.TAPS_MIN_ROWS) else None
        return self._taps

    def inverse(self, H0, F, Hv):
        """For the permuted pc2_corr_indices [K, F*H0] (values: pc2 vertices < Hv): int32 [F, K*Hv] with
        inverse[f][v*K + k] = the virtual vertex f*H0 + h that gathers v through tap k (ops.table_invert), built once."""
        if getattr(self, '_inv', None) is None:
            self._inv = ops.table_invert(self.t, H0, F, Hv)
        return self._inv

    @property
    def symmetric(self):
        if self._sym is None:
            self._sym = ops.table_is_symmetric(self.t)
        return self._sym

    def bwd_mode(self, rows_a):
        return 'mirror' if (rows_a == self.t.shape[1] and self.symmetric) else 'scatter'


class _TableCache(object):
    """Conversions of the reference's int64 wire tensors, keyed by tensor identity.

    The source tensor is kept alive by the entry, so a recycled data_ptr cannot alias."""

    def __init__(self, capacity=96):
        self.capacity = capacity
        self.d = collections.OrderedDict()

    def get(self, kind, tensors, make):
        key = (kind,) + tuple((t.data_ptr(), tuple(t.shape), t._version) for t in tensors)
        hit = self.d.get(key)
        if hit is not None:
            self.d.move_to_end(key)
            return hit[0]
        val = make()
        self.d[key] = (val, tensors)
        if len(self.d) > self.capacity:
            self.d.popitem(last=False)
        return val


_CACHE = _TableCache()


def _dev(t, like):
    return t if t.device == like.device else t.to(like.device, non_blocking=True)


def cloud_tables(bary, off, H, like):
    """(1,4,N) bary + (1,4,N) int64 offsets -> ops.CloudTables (cached)."""
    def make():
        return ops.CloudTables(_dev(bary, like).reshape(4, -1), _dev(off, like).reshape(4, -1), H)
    return _CACHE.get('cloud', (bary, off), make)


def nbr_table(t, like):
    """(1, F, H) int64 -> NbrTable int32 [F, H] (cached)."""
    def make():
        x = _dev(t, like)
        return NbrTable(ops.narrow(x.reshape(x.shape[-2], x.shape[-1])))
    return _CACHE.get('nbr', (t,), make)


def corr2_table(t, like):
    """(1, F, K, H) int64 -> NbrTable int32 [K, F*H] (cached)."""
    def make():
        x = _dev(t, like)
        return NbrTable(ops.corr2_permute(x.reshape(x.shape[-3], x.shape[-2], x.shape[-1])))
    return _CACHE.get('corr2', (t,), make)


class RegularTable(object):
    """The displacement filter's "table": tap f of vertex h reads row f*H + h.  Nothing is stored -- the kernels
    compute the source row from `reg_stride` (hpl_gconv_desc.reg_stride) -- so nothing needs caching per pair;
    the F*H source rows are distinct, so the data gradient is a plain GEMM + strided copy (bwd mode 'regular')."""
    t = None
    perm = None
    symmetric = False

    def __init__(self, F, H):
        self.F, self.reg_stride = F, H

    def groups(self):
        return None

    def taps(self):
        return None

    def bwd_mode(self, rows_a):
        return 'regular'


def regular_table(F, H, device=None):
    return RegularTable(F, H)


# ----------------------------------------------------------------------------- sparse_sum (a1)
class SparseSum(torch.autograd.Function):
    """out[idx[j], :] += values[j, :]  (reference: models/bilateralNN.py:9-40)."""

    @staticmethod
    def forward(ctx, indices, values, size, cuda):
        if not values.is_cuda:
            raise _lib.HplError('sparse_sum: the HIP path needs device tensors (no CPU fallback)')
        idx = ops.narrow(indices.reshape(-1).to(values.device))
        n = idx.numel()
        H = int(size[0])
        dev = values.device
        csr_ptr = torch.empty(H + 1, dtype=torch.int32, device=dev)
        csr_pt = torch.empty(n, dtype=torch.int32, device=dev)
        csr_w = torch.empty(n, dtype=torch.float32, device=dev)
        norm = torch.empty(H, dtype=torch.float32, device=dev)
        scratch = torch.empty(H + 1 + n + 1026, dtype=torch.int32, device=dev)
        ones = torch.ones(n, dtype=torch.float32, device=dev)
        _lib.check(_lib.load().hpl_csr_build(_lib.ptr(idx), _lib.ptr(ones), n, n, H, _lib.ptr(csr_ptr),
                                             _lib.ptr(csr_pt), _lib.ptr(csr_w), _lib.ptr(norm),
                                             _lib.ptr(scratch), _lib.stream()), 'hpl_csr_build')
        ctx.save_for_backward(indices)
        v = values if values.stride(1) == 1 else values.contiguous()
        return ops.splat_raw(v, (csr_ptr, csr_pt, csr_w, norm), H, use_norm=False)

    @staticmethod
    def backward(ctx, grad_output):
        indices, = ctx.saved_tensors
        g = None
        if ctx.needs_input_grad[1]:
            g = grad_output[indices.reshape(-1).to(grad_output.device), :]
        return None, g, None, None


sparse_sum = SparseSum.apply


# ----------------------------------------------------------------------------- BilateralConvFlex
def _build_conv_stack(n_in, num_output, filter_size, last_relu, use_leaky, first_cls=Conv2dReLU,
                      plain_cls=nn.Conv2d):
    """Conv stack of models/bilateralNN.py:94-112: first kernel (filter_size, 1), the rest (1, 1);
    the last conv is a bare conv unless last_relu."""
    mods = []
    c = n_in
    for i, o in enumerate(num_output):
        ks = (filter_size, 1) if i == 0 else (1, 1)
        last = i == len(num_output) - 1
        if last and not last_relu:
            mods.append(plain_cls(c, o, kernel_size=ks))
        else:
            mods.append(first_cls(c, o, ks, use_leaky=use_leaky))
        c = o
    return nn.Sequential(*mods)


def _conv_of(m):
    return m.conv if isinstance(m, _ConvReLU) else m


def _run_conv_stack(x, stack, table, M, F, use_leaky, out=None):
    """x [rows, C] -> [M, O_last]: first conv gathers through `table` (F taps), the rest are 1x1."""
    n = len(stack)
    for i, m in enumerate(stack):
        conv = _conv_of(m)
        act = ACT_LEAKY if isinstance(m, _ConvReLU) else ACT_NONE
        o = out if i == n - 1 else None
        if i == 0:
            groups = table.groups() if conv.in_channels >= GROUPS_MIN_CHANNELS else None
            x = ops.gconv(x, conv.weight, conv.bias, table.t, M, F, act=act,
                          bwd_mode=table.bwd_mode(x.shape[0]) if torch.is_grad_enabled() else 'scatter',
                          out=o, slope=_slope(use_leaky),
                          row_perm=table.perm if groups is None else None,     # (the passes bring their own orders)
                          taps=table.taps if conv.in_channels >= 100 else None, tap_groups=groups,
                          reg_stride=getattr(table, 'reg_stride', 0),
                          tiles=(table.group_tiles() if groups is not None else table.perm_tiles) if table.t is not None else None)
        else:
            x = ops.gconv(x, conv.weight, conv.bias, None, M, 1, act=act, bwd_mode='dense', out=o,
                          slope=_slope(use_leaky))
    return x


class BilateralConvFlex(nn.Module):
    """DownBCL / UpBCL: splat -> blur (gathered conv stack) -> slice.  Constructor and forward
    signatures of /root/reference/models/bilateralNN.py:47-57,122-125."""

    def __init__(self, d, neighborhood_size, num_input, num_output, DEVICE, use_bias, use_leaky, use_norm,
                 do_splat, do_slice, last_relu, chunk_size=1024 * 1024 * 25):
        super(BilateralConvFlex, self).__init__()
        self.d, self.d1 = d, d + 1
        self.neighborhood_size = neighborhood_size
        self.filter_size = (neighborhood_size + 1) ** self.d1 - neighborhood_size ** self.d1
        self.num_input, self.num_output = num_input, list(num_output)
        self.DEVICE = DEVICE
        self.use_bias, self.use_leaky, self.use_norm = use_bias, use_leaky, use_norm
        self.do_splat, self.do_slice, self.last_relu = do_splat, do_slice, last_relu
        self.MAX_SIZE = chunk_size                     # accepted, unused (no materialised gather)
        self.register_buffer('feat_indices', torch.arange(num_input, dtype=torch.long))
        if do_slice:
            self.register_buffer('out_indices', torch.arange(num_output[-1], dtype=torch.long))
        self.blur_conv = _build_conv_stack(num_input, self.num_output, self.filter_size, last_relu, use_leaky)
        if do_slice and use_bias:
            self.register_parameter('bias', nn.Parameter(torch.zeros((num_output[-1],), dtype=torch.float32)))

    def get_filter_size(self):
        return self.filter_size

    def forward_cl(self, x, in_cloud, blur, out_cloud, out=None):
        """Channel-last core.  x [N_in | H, C_in]; blur: NbrTable [15, H]; clouds: ops.CloudTables."""
        H = blur.t.shape[1]
        if self.do_splat:
            if in_cloud.H != H:
                raise _lib.HplError('splat target has %d vertices, blur table %d' % (in_cloud.H, H))
            if torch.is_grad_enabled() and x.requires_grad:
                s = ops.SplatFn.apply(x, in_cloud, self.use_norm)
            else:
                s = ops.splat_raw(x, in_cloud.csr(), H, self.use_norm)
        else:
            if x.shape[0] != H:
                raise _lib.HplError('features have %d rows, blur table %d vertices' % (x.shape[0], H))
            s = x
        bias = (self.bias if self.use_bias else None) if self.do_slice else None
        mods = list(self.blur_conv)
        if (self.do_slice and len(mods) >= 2 and not isinstance(mods[-1], _ConvReLU) and out_cloud.N < H):
            # Reordering.  The last conv is a bias-only 1x1 (no activation) and the slice is linear, so
            # slice(W y + b) = W slice(y) + b * sum_r(bary_r): run the 1x1 conv on the N_out sliced rows
            # instead of the H lattice vertices (bcn1_: 25 841 -> 8 192 rows; forward, data gradient and
            # weight gradient all shrink).  sum_r(bary_r) = 1 up to fp32 rounding (transforms.py:340-345),
            # the difference (<= 1e-6 |b|) is inside the tolerance; gradients are those of the same function.
            y = _run_conv_stack(s, mods[:-1], blur, H, self.filter_size, self.use_leaky)
            conv = mods[-1]
            b = conv.bias if bias is None else (conv.bias + bias if conv.bias is not None else bias)
            if torch.is_grad_enabled() and y.requires_grad:
                z = ops.SliceFn.apply(y, out_cloud, None)
            else:
                z = ops.slice_raw(y, out_cloud.bary, out_cloud.off, out_cloud.N)
            return ops.gconv(z, conv.weight, b, None, out_cloud.N, 1, act=ACT_NONE, bwd_mode='dense', out=out)
        y = _run_conv_stack(s, self.blur_conv, blur, H, self.filter_size, self.use_leaky,
                            out=None if self.do_slice else out)
        if not self.do_slice:
            return y
        if torch.is_grad_enabled() and (y.requires_grad or (bias is not None and bias.requires_grad)):
            z = ops.SliceFn.apply(y, out_cloud, bias)
            if out is not None:
                out.copy_(z)
                return out
            return z
        return ops.slice_raw(y, out_cloud.bary, out_cloud.off, out_cloud.N, bias=bias, out=out)

    def forward(self, features, in_barycentric, in_lattice_offset, blur_neighbors, out_barycentric,
                out_lattice_offset):
        if features.size(0) != 1:
            raise _lib.HplError('batch size must be 1 (reference README.md:57)')
        x = to_channel_last(features)
        H = blur_neighbors.size(-1)
        blur = nbr_table(blur_neighbors, x)
        in_cloud = cloud_tables(in_barycentric, in_lattice_offset, H, x) if self.do_splat else None
        out_cloud = cloud_tables(out_barycentric, out_lattice_offset, H, x) if self.do_slice else None
        return to_channel_first(self.forward_cl(x, in_cloud, blur, out_cloud))


# ----------------------------------------------------------------------------- BilateralCorrelationFlex
class BilateralCorrelationFlex(nn.Module):
    """CorrBCL: patch correlation + displacement filtering.  Constructor and forward signatures of
    /root/reference/models/bnn_flow.py:11-20,96-99.

    The reference builds a (C1+C2, F, K, H1) tensor in which the pc1 half is repeated over the F
    displacement taps (bnn_flow.py:192); here that half of the Conv3d is contracted once per vertex
    (A-term) and enters the pc2 half's GEMM over the F*H1 virtual vertices as a row-broadcast
    residual, then Conv3d 1x1x1 and the displacement filter Conv2d((F,1)) follow as gather-GEMMs.
    """

    def __init__(self, d, corr_filter_radius, corr_corr_radius, num_input, num_corr_output, num_output, DEVICE,
                 use_bias, use_leaky, use_norm, prev_corr_dim, last_relu, chunk_size=1024 * 1024 * 25):
        super(BilateralCorrelationFlex, self).__init__()
        self.d, self.d1 = d, d + 1
        self.corr_size = (corr_corr_radius + 1) ** self.d1 - corr_corr_radius ** self.d1
        self.filter_size = (corr_filter_radius + 1) ** self.d1 - corr_filter_radius ** self.d1
        self.num_input, self.prev_corr_dim = num_input, prev_corr_dim
        self.num_output = list(num_output)
        self.DEVICE, self.use_norm, self.last_relu, self.use_leaky = DEVICE, use_norm, last_relu, use_leaky
        self.MAX_SIZE = chunk_size
        self.register_buffer('feat_indices', torch.arange(num_input, dtype=torch.long))
        if prev_corr_dim != 0:
            self.register_buffer('feat1_indices', torch.arange(num_input + prev_corr_dim, dtype=torch.long))
        else:
            self.feat1_indices = self.feat_indices
        self.register_buffer('out_indices', torch.arange(num_output[-1], dtype=torch.long))
        mods, c = [], num_input * 2 + prev_corr_dim
        for i, o in enumerate(num_corr_output):
            ks = (1, self.corr_size, 1) if i == 0 else (1, 1, 1)
            mods.append(Conv3dReLU(c, o, ks, use_leaky=use_leaky))
            c = o
        self.corr_conv = nn.Sequential(*mods)
        self.blur_conv = _build_conv_stack(c, self.num_output, self.filter_size, last_relu, use_leaky)

    def get_filter_size(self, dist):
        return (dist + 1) ** self.d1 - dist ** self.d1

    def forward_cl(self, f1, f2, prev, cloud1, corr1, corr2, out=None):
        """f1 [H1, C], f2 [H2, C], prev [N_in, P] or None, corr1 NbrTable [K, H1], corr2 NbrTable
        [K, F*H1] (virtual vertices m = f*H1 + h)."""
        H1, C, P = f1.shape[0], self.num_input, self.prev_corr_dim
        K, F = self.corr_size, self.filter_size
        sl = _slope(self.use_leaky)
        if corr1.t.shape != (K, H1) or corr2.t.shape != (K, F * H1):
            raise _lib.HplError('corr tables %s / %s do not match K=%d F=%d H1=%d'
                                % (tuple(corr1.t.shape), tuple(corr2.t.shape), K, F, H1))
        conv0 = self.corr_conv[0].conv
        w0 = conv0.weight                                  # (O, P + 2C, 1, K, 1), channels [prev | f1 | f2]
        mode1 = corr1.bwd_mode(H1) if torch.is_grad_enabled() else 'scatter'   # symmetry check syncs
        # A-term: pc1 half, independent of the displacement tap
        perm1 = corr1.perm
        a = ops.gconv(f1, w0, None, corr1.t, H1, K, c0=P, C=C, bwd_mode=mode1, row_perm=perm1, tiles=corr1.perm_tiles)
        if prev is not None:
            if P == 0:
                raise _lib.HplError('prev_corr_feat given but prev_corr_dim == 0')
            if torch.is_grad_enabled() and prev.requires_grad:
                ps = ops.SplatFn.apply(prev, cloud1, self.use_norm)
            else:
                ps = ops.splat_raw(prev, cloud1.csr(), H1, self.use_norm)
            a = ops.gconv(ps, w0, None, corr1.t, H1, K, c0=0, C=P, res=a, res_mod=H1, bwd_mode=mode1, row_perm=perm1,
                          tiles=corr1.perm_tiles)
        # B-term over the F*H1 virtual vertices, + broadcast A-term + bias, LeakyReLU
        if w0.shape[0] % 4 == 0 and K == 15 and F == 15:
            # every pc2 vertex projected once per correlation tap, then a gather-sum of 32-float rows (ops.CorrPc2Fn)
            p = ops.corr_pc2(f2, w0, conv0.bias, a, corr2, H1, F, K, P + C, C, sl)
        else:
            p = ops.gconv(f2, w0, conv0.bias, corr2.t, F * H1, K, act=ACT_LEAKY, c0=P + C, C=C, res=a,
                          res_mod=H1, bwd_mode='scatter', slope=sl)
        for m in list(self.corr_conv)[1:]:
            p = ops.gconv(p, m.conv.weight, m.conv.bias, None, F * H1, 1, act=ACT_LEAKY, bwd_mode='dense',
                          slope=sl)
        # displacement filter: tap f of vertex h is row f*H1 + h of p
        reg = regular_table(F, H1, f1.device)
        return _run_conv_stack(p, self.blur_conv, reg, H1, F, self.use_leaky, out=out)

    def forward(self, feat1, feat2, prev_corr_feat, barycentric1, lattice_offset1, pc1_corr_indices,
                pc2_corr_indices, max_hash_cnt1, max_hash_cnt2):
        if feat1.size(0) != 1:
            raise _lib.HplError('batch size must be 1 (reference README.md:57)')
        f1, f2 = to_channel_last(feat1), to_channel_last(feat2)
        if f1.shape[0] != max_hash_cnt1 or f2.shape[0] != max_hash_cnt2:
            raise _lib.HplError('feature rows (%d, %d) != hash counts (%d, %d)'
                                % (f1.shape[0], f2.shape[0], max_hash_cnt1, max_hash_cnt2))
        prev, cloud1 = None, None
        if prev_corr_feat is not None:
            prev = to_channel_last(prev_corr_feat)
            cloud1 = cloud_tables(barycentric1, lattice_offset1, max_hash_cnt1, f1)
        corr1 = nbr_table(pc1_corr_indices, f1)
        corr2 = corr2_table(pc2_corr_indices, f1)
        return to_channel_first(self.forward_cl(f1, f2, prev, cloud1, corr1, corr2))
