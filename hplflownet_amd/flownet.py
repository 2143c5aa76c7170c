"""Model assembly on top of the HIP bilateral layers: HPLFlowNet and HPLFlowNetShallow.

Own counterpart of the reference's callers of the hot path:
same module names, hence the same state_dict keys and shapes as
/root/reference/models/HPLFlowNet.py:11-236 and models/HPLFlowNet_shallow.py:11-169
(checked against tests/golden/state_dict.json), same forward signature
`model(pc1, pc2, generated_data) -> (1, 3, N)`.  The wiring is table driven instead of
spelled out layer by layer, runs channel-last end to end, and replaces every torch.cat of
the reference forward by writes into column slices of one buffer per layer input.

Level L (0-based) hosts bcn{L+1} (Down, shared by both clouds), bcn{L+1}_ (Up) and, for
L >= 2, corr{L-1}.
"""
import os
import weakref

import torch
import torch.nn as nn

from . import _lib, ops
from .bcl import (BilateralConvFlex, BilateralCorrelationFlex, Conv1dReLU, NbrTable, pointwise_conv,
                  to_channel_first, to_channel_last)

__all__ = ['HPLFlowNet', 'HPLFlowNetShallow', 'DeviceLattice', 'PairBlur']


# ----------------------------------------------------------------------------- lattice container
class _Level(object):
    # pair / emg_pair: both clouds as one (ops.PairTables, el_minus_gr [N0+N1, 4]); None when the
    # lattice came from the reference's per-cloud wire format
    __slots__ = ('clouds', 'blur', 'emg', 'corr1', 'corr2', 'H', 'pair', 'emg_pair')

    def __init__(self):
        self.pair = None
        self.emg_pair = None


class PairBlur(object):
    """Blur tables of the two clouds of a level as ONE int32 table [F, H0+H1]: columns [0,H0) are
    cloud 1's vertices, columns [H0,H0+H1) cloud 2's with neighbour ids shifted by H0 (so the table
    indexes the pair's stacked feature matrix).  Behaves like the list [blur1, blur2]:
    [0] is a zero-copy view, [1] (cloud 2's own numbering) is materialised on first use."""

    def __init__(self, table, H0):
        self.pair = NbrTable(table)
        self.H0 = H0
        self._own = [NbrTable(table[:, :H0]), None]

    def __len__(self):
        return 2

    def __getitem__(self, i):
        if i == 1 and self._own[1] is None:
            t = self.pair.t[:, self.H0:]
            self._own[1] = NbrTable(torch.where(t >= 0, t - self.H0, t).contiguous())
            self._own[1]._sym = self.pair._sym
        return self._own[i]

    def __iter__(self):
        return iter((self[0], self[1]))


class DeviceLattice(object):
    """`generated_data` resident on the device in kernel-ready form:
    int32 tables, CSR of each splat, channel-last el_minus_gr.  Built either from the
    reference's list of dicts (host or device tensors, with or without the B=1 dimension
    added by default_collate) or directly by hplflownet_amd.lattice on the GPU."""

    def __init__(self, levels, wide_up=None):
        self.levels = levels
        #: hint from the consumer (model.lattice_hint()), one entry per level (or one value for all): True = the Up
        #: conv of that level is wide enough to run as tap-group passes (the single-pass row order of its table is
        #: never used), False = it is not (the group orders are never used), None = unknown: prepare() builds both
        self.wide_up = wide_up

    def prepare(self, for_training=False):
        """Build every lazily constructed table (CSRs, tap orders, symmetry verdicts) now, on the
        current stream, so that a lattice built on a side stream is complete before it is handed to
        the forward.  for_training: the tables of the per-cloud path + the symmetry read-back."""
        self.prepare_tables(for_training)
        if for_training:
            self.resolve_symmetry()
        return self

    def prepare_tables(self, for_training=False):
        """The launches of prepare() without its read-back (lattice.LatticeBuild overlaps that one)."""
        for L, lv in enumerate(self.levels):
            if lv.pair is not None:
                lv.pair.csr()       # one build; the per-cloud CSRs are views / offset copies of it
            if lv.pair is not None and isinstance(lv.blur, PairBlur):
                # the Down layers run once per pair; cloud 1 alone is splatted only by the correlation
                # layers that take a previous correlation (levels >= 3) and sliced by the Up layers
                tables = [lv.blur.pair, lv.blur[0], lv.corr1]
                if for_training:
                    lv.clouds[0].csr()
            else:
                for c in lv.clouds:
                    c.csr()
                tables = list(lv.blur) + [lv.corr1]
            up = lv.blur[0] if isinstance(lv.blur, PairBlur) else None
            wide = self.wide_up[L] if isinstance(self.wide_up, (list, tuple)) else self.wide_up
            grouped = up is not None and wide is not False and up.groups() is not None   # multi-pass row orders
            if grouped:
                up.group_tiles()
            for tbl in tables:
                if tbl is not None and not (tbl is up and grouped and wide and tbl is not lv.corr1):
                    tbl.perm_tiles          # (builds the row order first)
        return self

    def symmetry_begin(self):
        """Launch the symmetry checks of every blur / corr1 table not decided yet and start the copy of
        their flags to pinned memory -> (todo, event or None); symmetry_finish(todo) after the event."""
        todo = []
        for lv in self.levels:
            tables = [lv.corr1]
            if isinstance(lv.blur, PairBlur):
                tables += [lv.blur.pair]
            else:
                tables += list(lv.blur)
            for t in tables:
                if t is not None and t._sym is None and t.t.shape[0] == 15 and all(t is not u for u, _ in todo):
                    todo.append((t, ops.table_symmetry_flag(t.t)))
        if not todo:
            return (todo, None), None
        flags = torch.cat([f for _, f in todo])
        host = torch.empty(flags.shape, dtype=flags.dtype, pin_memory=True)
        host.copy_(flags, non_blocking=True)
        landed = torch.cuda.Event()
        landed.record()
        return (todo, host), landed

    def symmetry_finish(self, begun):
        todo, host = begun
        if todo:
            for (t, _), v in zip(todo, host.tolist()):
                t._sym = bool(v)
        for lv in self.levels:
            if isinstance(lv.blur, PairBlur):            # the per-cloud views inherit the pair's verdict
                for i in (0, 1):
                    if lv.blur._own[i] is not None and lv.blur._own[i]._sym is None:
                        lv.blur._own[i]._sym = lv.blur.pair._sym
        return self

    def resolve_symmetry(self):
        """Decide `symmetric` of every blur / corr1 table that has not been checked yet with ONE host
        read-back (the backward picks the mirrored-gather or the atomic-scatter form from it)."""
        begun, landed = self.symmetry_begin()
        if landed is not None:
            landed.synchronize()
        return self.symmetry_finish(begun)

    @staticmethod
    def from_generated_data(gd, device):
        levels = []
        for d in gd:
            lv = _Level()

            def t(key):
                v = d[key]
                v = torch.as_tensor(v)
                return v.to(device, non_blocking=True)

            def cnt(key):
                v = d[key]
                return int(v.reshape(-1)[0].item()) if torch.is_tensor(v) else int(v)

            lv.H = (cnt('pc1_hash_cnt'), cnt('pc2_hash_cnt'))
            lv.clouds, lv.blur, lv.emg = [], [], []
            for ci, nm in enumerate(('pc1', 'pc2')):
                bary = t(nm + '_barycentric').reshape(4, -1).float()
                off = t(nm + '_lattice_offset').reshape(4, -1)
                lv.clouds.append(ops.CloudTables(bary, off, lv.H[ci]))
                bl = t(nm + '_blur_neighbors')
                lv.blur.append(NbrTable(ops.narrow(bl.reshape(-1, lv.H[ci]))) if bl.numel() > 1 else None)
                if lv.blur[-1] is not None:      # the same sparsity rule as a device-built lattice (tap groups only where most slots are empty)
                    lv.blur[-1].vertices_per_point = lv.H[ci] / float(max(1, bary.shape[1]))
                lv.emg.append(t(nm + '_el_minus_gr').reshape(4, -1).float().t().contiguous())
            c1 = t('pc1_corr_indices')
            if c1.numel() > 1:
                lv.corr1 = NbrTable(ops.narrow(c1.reshape(-1, lv.H[0])))
                c2 = t('pc2_corr_indices')
                c2 = c2.reshape(-1, lv.corr1.t.shape[0], lv.H[0])
                lv.corr2 = NbrTable(ops.corr2_permute(c2))
                lv.corr2._sym = False
            else:
                lv.corr1 = lv.corr2 = None
            levels.append(lv)
        return DeviceLattice(levels)


def _assemble(rows, parts, device):
    """Concatenate channel blocks into one [rows, sum C] matrix.  A part is (C, tensor) or
    (C, callable(out_view) -> tensor).  Without autograd the blocks are written in place
    (no cat copy for callables); with autograd this is a plain torch.cat."""
    if torch.is_grad_enabled():
        return torch.cat([src(None) if callable(src) else src for _, src in parts], dim=1)
    total = sum(c for c, _ in parts)
    buf = torch.empty((rows, total), dtype=torch.float32, device=device)
    col = 0
    for c, src in parts:
        view = buf[:, col:col + c]
        if callable(src):
            src(view)
        else:
            view.copy_(src)
        col += c
    return buf


# ----------------------------------------------------------------------------- the two models
_PLANS = weakref.WeakKeyDictionary()        # model -> plan.ForwardPlan (kept outside the module: deepcopy / state_dict safe)


class _FlowNetBase(nn.Module):
    """Shared wiring.  Subclasses define SPEC."""

    NLEV = None          # number of lattice levels
    DOWN = None          # num_output of every Down BCL
    CORR = None          # (num_corr_output, num_output) of every CorrBCL
    UP = None            # per level L: num_output of bcn{L+1}_
    REFINE = False       # corr{j}_refine Conv1d stacks (shallow model)
    HEAD_IN = None
    #: on a device-built lattice the Down path runs once per PAIR (both clouds stacked), in inference and
    #: in training; False forces the per-cloud path (what reference-format lattices use)
    pair_batched = True
    #: inference on a device-built lattice runs as ONE native call (plan.ForwardPlan: the same launches issued by
    #: csrc/executor.hip instead of ~130 Python round trips); False forces the Python path below
    native_forward = not os.environ.get('HPL_NO_NATIVE')

    def __init__(self, args):
        super(_FlowNetBase, self).__init__()
        self.scales_filter_map = args.scales_filter_map
        assert len(self.scales_filter_map) == self.NLEV
        sfm = self.scales_filter_map
        dim, leaky = args.dim, args.use_leaky
        self.use_leaky = leaky
        chunk = -1 if getattr(args, 'evaluate', False) else 1024 * 1024 * 25
        self.chunk_size = chunk

        def bcl(n_in, n_out, radius, splat, slice_):
            return BilateralConvFlex(dim, radius, n_in, n_out, args.DEVICE, use_bias=args.bcn_use_bias,
                                     use_leaky=leaky, use_norm=args.bcn_use_norm, do_splat=splat,
                                     do_slice=slice_, last_relu=args.last_relu, chunk_size=chunk)

        self.conv1 = nn.Sequential(Conv1dReLU(dim, 32, use_leaky=leaky), Conv1dReLU(32, 32, use_leaky=leaky),
                                   Conv1dReLU(32, 64, use_leaky=leaky))
        feat = 64
        corr_dim = {}                     # channels of the (refined) correlation living at level L
        for L in range(self.NLEV):
            setattr(self, 'bcn%d' % (L + 1), bcl(feat + dim + 1, self.DOWN, sfm[L][1], True, False))
            if L >= 2:
                j = L - 1
                setattr(self, 'corr%d' % j, BilateralCorrelationFlex(
                    dim, sfm[L][2], sfm[L][3], feat, self.CORR[0], self.CORR[1], args.DEVICE,
                    use_bias=args.bcn_use_bias, use_leaky=leaky, use_norm=args.bcn_use_norm,
                    prev_corr_dim=0 if L == 2 else corr_dim[L - 1], last_relu=args.last_relu,
                    chunk_size=chunk))
                corr_dim[L] = self.CORR[1][-1]
                if self.REFINE:
                    c_in = corr_dim[L] + (dim + 1 if L + 1 < self.NLEV else 0)
                    setattr(self, 'corr%d_refine' % j, nn.Sequential(
                        Conv1dReLU(c_in, 64, use_leaky=leaky), Conv1dReLU(64, 64, use_leaky=leaky),
                        Conv1dReLU(64, 64, use_leaky=leaky)))
                    corr_dim[L] = 64
        up_out = None
        for L in reversed(range(self.NLEV)):
            if L == self.NLEV - 1:
                n_in = corr_dim[L] + feat
            else:
                n_in = dim + 1 + up_out + (corr_dim[L] if L >= 2 else 0) + feat
            setattr(self, 'bcn%d_' % (L + 1), bcl(n_in, self.UP[L], sfm[L][1], False, True))
            up_out = self.UP[L][-1]
        self.conv2 = Conv1dReLU(self.HEAD_IN, 1024, use_leaky=leaky)
        self.conv3 = Conv1dReLU(1024, 512, use_leaky=leaky)
        self.conv4 = nn.Conv1d(512, 3, kernel_size=1)

    def lattice_hint(self):
        """What this model needs of a lattice's lazily built tables (DeviceLattice.wide_up): per level, whether the
        Up conv that gathers through the level's blur table is wide enough for the tap-group passes."""
        from .bcl import GROUPS_MIN_CHANNELS
        return [getattr(self, 'bcn%d_' % (L + 1)).num_input >= GROUPS_MIN_CHANNELS for L in range(self.NLEV)]

    def forward_plan(self):
        """The native plan of this model's inference forward (built on first use, rebuilt when a parameter was
        replaced by a new tensor; in-place parameter updates only refresh its weight images)."""
        from .plan import ForwardPlan
        plan = _PLANS.get(self)
        if plan is None or not plan.fresh():
            plan = _PLANS[self] = ForwardPlan(self)
        return plan

    # -- helpers ------------------------------------------------------------------------
    def _stack(self, x, seq, out=None):
        mods = list(seq)
        for i, m in enumerate(mods):
            x = pointwise_conv(x, m.conv, True, self.use_leaky, out=out if i == len(mods) - 1 else None)
        return x

    def _corr(self, L, lat, feats, prev, corrs, dev):
        lv = lat.levels[L]
        j = L - 1
        c = getattr(self, 'corr%d' % j).forward_cl(feats[0], feats[1], prev,
                                                  lv.clouds[0] if prev is not None else None,
                                                  lv.corr1, lv.corr2)
        if self.REFINE:
            if L + 1 < self.NLEV:
                c = _assemble(c.shape[0], [(4, lat.levels[L + 1].emg[0]), (c.shape[1], c)], dev)
            c = self._stack(c, getattr(self, 'corr%d_refine' % j))
        corrs[L] = c
        return c

    def forward(self, pc1, pc2, generated_data):
        dev = pc1.device
        if not pc1.is_cuda:
            raise _lib.HplError('the HIP path needs device tensors (no CPU fallback)')
        native_lat = getattr(generated_data, 'device_lattice', None)       # lattice.NativeLattice
        # a natively built lattice lives in one arena, usually allocated on the lattice stream: tell the allocator that this
        # stream reads it, so that dropping the lattice right after the call cannot hand the memory to the next build early
        arena = getattr(generated_data, 'arena', None)
        if arena is None:
            arena = getattr(generated_data, '_arena', None)
        if arena is not None:
            arena.record_stream(torch.cuda.current_stream(dev))
        if self.native_forward and self.pair_batched and not torch.is_grad_enabled() and \
                (native_lat is not None or isinstance(generated_data, DeviceLattice)):
            plan = self.forward_plan()
            if plan.accepts(generated_data):
                return plan(pc1, pc2, generated_data)
        if native_lat is not None:
            generated_data = native_lat()
        lat = generated_data if isinstance(generated_data, DeviceLattice) else \
            DeviceLattice.from_generated_data(generated_data[:self.NLEV], dev)
        nlev = self.NLEV
        if torch.is_grad_enabled():
            lat.resolve_symmetry()
            if ops.BANK is not None:
                ops.BANK.refresh()          # all weight images of this step in one launch
        pair = self.pair_batched and \
            all(lv.pair is not None and isinstance(lv.blur, PairBlur) for lv in lat.levels[:nlev])
        down = [[], []]
        corrs = {}
        prev = None
        if pair and torch.is_grad_enabled():
            # the same stacked Down path written with autograd-visible ops (cat instead of in-place columns)
            y = self._stack(torch.cat([to_channel_last(pc1), to_channel_last(pc2)], dim=0), self.conv1)
            for L in range(nlev):
                lv = lat.levels[L]
                layer = getattr(self, 'bcn%d' % (L + 1))
                if L == 0 and (lv.clouds[0].N != pc1.shape[2] or lv.clouds[1].N != pc2.shape[2]):
                    raise _lib.HplError('lattice was built for %d / %d points, got %d / %d'
                                        % (lv.clouds[0].N, lv.clouds[1].N, pc1.shape[2], pc2.shape[2]))
                y = layer.forward_cl(torch.cat([lv.emg_pair, y], dim=1), lv.pair, lv.blur.pair, None)
                feats = [y[:lv.H[0]], y[lv.H[0]:]]
                down[0].append(feats[0])
                down[1].append(feats[1])
                if L >= 2:
                    prev = self._corr(L, lat, feats, prev, corrs, dev)
        elif pair:
            # Both clouds go through conv1 and the Down BCLs as ONE stacked matrix (cloud 2's points and
            # vertices behind cloud 1's; pair CSR, pair blur table): half the launches, and the output
            # of level L is written straight into columns [4, 4+C) of level L+1's input.
            feat_c = self.conv1[-1].conv.out_channels
            n0 = lat.levels[0].pair.N
            xin = torch.empty((n0, pc1.shape[1]), dtype=torch.float32, device=dev)
            if lat.levels[0].clouds[0].N != pc1.shape[2] or lat.levels[0].clouds[1].N != pc2.shape[2]:
                raise _lib.HplError('lattice was built for %d / %d points, got %d / %d'
                                    % (lat.levels[0].clouds[0].N, lat.levels[0].clouds[1].N, pc1.shape[2],
                                       pc2.shape[2]))
            h = pc1.shape[2]
            xin[:h].copy_(to_channel_last(pc1))
            xin[h:].copy_(to_channel_last(pc2))
            x = torch.empty((n0, 4 + feat_c), dtype=torch.float32, device=dev)
            self._stack(xin, self.conv1, out=x[:, 4:])
            for L in range(nlev):
                lv = lat.levels[L]
                layer = getattr(self, 'bcn%d' % (L + 1))
                x[:, :4].copy_(lv.emg_pair)
                c_out = layer.num_output[-1]
                H0, Hp = lv.H[0], lv.pair.H
                nxt = torch.empty((Hp, 4 + c_out), dtype=torch.float32, device=dev) if L + 1 < nlev else None
                y = layer.forward_cl(x, lv.pair, lv.blur.pair, None, out=nxt[:, 4:] if nxt is not None else None)
                feats = [y[:H0], y[H0:]]
                down[0].append(feats[0])
                down[1].append(feats[1])
                x = nxt
                if L >= 2:
                    prev = self._corr(L, lat, feats, prev, corrs, dev)
        else:
            feats = [self._stack(to_channel_last(pc1), self.conv1), self._stack(to_channel_last(pc2), self.conv1)]
            for L in range(nlev):
                lv = lat.levels[L]
                layer = getattr(self, 'bcn%d' % (L + 1))
                for ci in (0, 1):
                    cloud = lv.clouds[ci]
                    x = _assemble(cloud.N, [(4, lv.emg[ci]), (feats[ci].shape[1], feats[ci])], dev)
                    feats[ci] = layer.forward_cl(x, cloud, lv.blur[ci], None)
                    down[ci].append(feats[ci])
                if L >= 2:
                    prev = self._corr(L, lat, feats, prev, corrs, dev)
        up = None        # callable(out_view) producing the previous Up output, or None
        up_c = 0
        for L in reversed(range(nlev)):
            lv = lat.levels[L]
            layer = getattr(self, 'bcn%d_' % (L + 1))
            if L == nlev - 1:
                parts = [(corrs[L].shape[1], corrs[L]), (down[0][L].shape[1], down[0][L])]
            else:
                parts = [(4, lat.levels[L + 1].emg[0]), (up_c, up)]
                if L >= 2:
                    parts.append((corrs[L].shape[1], corrs[L]))
                parts.append((down[0][L].shape[1], down[0][L]))
            x = _assemble(lv.H[0], parts, dev)

            def produce(out, layer=layer, x=x, lv=lv):
                return layer.forward_cl(x, None, lv.blur[0], lv.clouds[0], out=out)
            up, up_c = produce, layer.num_output[-1]
        y = up(None)                                           # [N, HEAD_IN]
        y = pointwise_conv(y, self.conv2.conv, True, self.use_leaky)
        y = pointwise_conv(y, self.conv3.conv, True, self.use_leaky)
        y = pointwise_conv(y, self.conv4, False, self.use_leaky)
        return to_channel_first(y)


class HPLFlowNet(_FlowNetBase):
    """7-level model: /root/reference/models/HPLFlowNet.py:11-430."""
    NLEV = 7
    DOWN = [64, 64]
    CORR = ([32, 32], [64, 64])
    UP = {6: [128, 128], 5: [128, 128], 4: [128, 128], 3: [256, 256], 2: [256, 256], 1: [512, 512],
          0: [1024, 1024]}
    REFINE = False
    HEAD_IN = 1024


class HPLFlowNetShallow(_FlowNetBase):
    """5-level model: /root/reference/models/HPLFlowNet_shallow.py:11-311."""
    NLEV = 5
    DOWN = [64]
    CORR = ([32], [32])
    UP = {4: [64], 3: [64], 2: [64], 1: [64], 0: [128]}
    REFINE = True
    HEAD_IN = 128


def load_reference_checkpoint(model, checkpoint, strict=True):
    """Load a checkpoint written by the reference (`main_utils.save_checkpoint`, main_utils.py:54-64:
    dict with 'state_dict' of the DataParallel-wrapped model, keys prefixed 'module.', main.py:104,122)
    or a bare state_dict into one of the models above.  `checkpoint` is a path or the loaded object."""
    obj = torch.load(checkpoint, map_location='cpu') if isinstance(checkpoint, str) else checkpoint
    sd = obj.get('state_dict', obj) if isinstance(obj, dict) else obj
    sd = {(k[len('module.'):] if k.startswith('module.') else k): v for k, v in sd.items()}
    return model.load_state_dict(sd, strict=strict)
