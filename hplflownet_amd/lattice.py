"""Permutohedral lattice construction on the GPU.

Device counterpart of the reference's `GenerateDataUnsymmetric`
(/root/reference/transforms/transforms.py:264-490): same constructor argument (`args.dim`,
`args.scales_filter_map`), same call convention `gen([pc1, pc2, sf]) -> (pc1.T, pc2.T, sf.T,
generated_data)`, but `generated_data` is a `DeviceLattice` (int32 tables, CSR and
channel-last el_minus_gr already resident in HBM) which the models in
hplflownet_amd.flownet consume directly; `to_reference_format()` gives the reference's
list-of-dicts wire format (int64) for anything else.

The float part is bit-identical to the reference's torch-CPU arithmetic as measured in the
build container (see oracle/lattice_oracle.c); the integer part is exact.  One host read-back
per level remains: the vertex counts size the next level's arrays.  `build()` waits for each;
`LatticeBuild` / `LatticePipeline` keep several pairs under construction on one stream and only
resume a pair once its counts have landed in pinned memory, so the host thread never blocks on
them in steady state (bench.py and engine.Trainer feed the forward this way).
"""
import collections
import ctypes
import math
import os

import numpy as np
import time

import torch

from . import _lib, ops
from ._lib import check, ptr, stream
from .bcl import NbrTable
from .flownet import DeviceLattice, PairBlur, _Level


def _filter_size(radius, d1=4):
    return (radius + 1) ** d1 - radius ** d1


#: seconds the pipeline's host thread spent idle, waiting for vertex counts from the GPU (bench.py reports it)
WAIT = {'s': 0.0}
#: CPU seconds the producer thread of a LatticePipeline spent driving builds (bench.py reports it per step)
BUSY = {'s': 0.0}

class GenerateDataUnsymmetric(object):
    def __init__(self, args, device='cuda', wide_up=None):
        self.wide_up = wide_up          # model.lattice_hint(): which lazily built row orders the consumer uses
        self.d = args.dim
        if self.d != 3:
            raise _lib.HplError('only d = 3 is implemented (reference configs use dim: 3)')
        self.d1 = self.d + 1
        self.scales_filter_map = args.scales_filter_map
        self.expected_std = (self.d + 1) * math.sqrt(2 / 3)        # transforms.py:275
        self.device = torch.device(device)

    def native_builder(self):
        if getattr(self, '_native', None) is None:
            self._native = NativeBuilder(self)
        return self._native

    def native_supported(self):
        """True if csrc/lattice_builder.hip can build this configuration (radius-1 stencils, <= 8 levels)."""
        try:
            self.native_builder()
            return True
        except _lib.HplError:
            return False

    def build_native(self, pc1, pc2):
        """The same lattice from the native builder (one arena, one C call per level half): -> NativeLattice."""
        return NativeLatticeBuild(self, pc1, pc2).finish()

    def build(self, pc1, pc2):
        """pc1, pc2: (3, N) float32 device tensors -> DeviceLattice (blocks on every read-back)."""
        steps = self.build_steps(pc1, pc2)
        try:
            while True:
                next(steps).synchronize()
        except StopIteration as fin:
            return fin.value

    def build_steps(self, pc1, pc2):
        """Generator form of build(): launches everything up to the next read-back of vertex counts
        (an asynchronous copy into pinned memory), yields the event that marks its arrival and must be
        resumed -- on the same stream -- only after that event has completed; returns the DeviceLattice
        (StopIteration.value)."""
        L = _lib.load()
        dev = pc1.device
        pts = [pc1.contiguous().float(), pc2.contiguous().float()]     # level 0: the clouds themselves
        n = [int(pts[0].shape[1]), int(pts[1].shape[1])]
        prev = None           # deeper levels: (vertex keys of the level above, their column stride, divisor)
        levels = []
        nlev = len(self.scales_filter_map)
        counts_host = torch.empty((nlev, 2), dtype=torch.int32, pin_memory=True)
        for idx, (scale, bcn_r, cf_r, cc_r) in enumerate(self.scales_filter_map):
            emg_p = torch.empty((n[0] + n[1], 4), dtype=torch.float32, device=dev)     # both clouds, point-major
            emg = [emg_p[:n[0]], emg_p[n[0]:]]
            keys = [torch.empty((4, n[c], 4), dtype=torch.int32, device=dev) for c in (0, 1)]
            bary = [torch.empty((4, n[c]), dtype=torch.float32, device=dev) for c in (0, 1)]
            if prev is None:
                check(L.hpl_lattice_keys_pair(ptr(pts[0]), ptr(pts[1]), None, None, 0, 0, 1.0, n[0], n[1], float(scale),
                                              ptr(keys[0]), ptr(keys[1]), ptr(bary[0]), ptr(bary[1]), ptr(emg[0]),
                                              ptr(emg[1]), 4, stream()), 'hpl_lattice_keys_pair')
            else:       # the points are the vertices of the level above (transforms.py:461-467), never materialised
                pvk, pstride, div = prev
                check(L.hpl_lattice_keys_pair(None, None, ptr(pvk[0]), ptr(pvk[1]), pstride[0], pstride[1], div, n[0],
                                              n[1], float(scale), ptr(keys[0]), ptr(keys[1]), ptr(bary[0]),
                                              ptr(bary[1]), ptr(emg[0]), ptr(emg[1]), 4, stream()),
                      'hpl_lattice_keys_pair')
            wsb = int(L.hpl_lattice_workspace_bytes(n[0], n[1]))
            ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
            off = [torch.empty((4, n[c]), dtype=torch.int32, device=dev) for c in (0, 1)]
            vk = [torch.empty((4, 4 * n[c]), dtype=torch.int32, device=dev) for c in (0, 1)]
            counts = torch.empty(2, dtype=torch.int32, device=dev)
            check(L.hpl_lattice_hash(ptr(keys[0]), n[0], ptr(keys[1]), n[1], ptr(off[0]), ptr(off[1]), ptr(vk[0]),
                                     ptr(vk[1]), ptr(counts), ptr(ws), wsb, stream()), 'hpl_lattice_hash')
            counts_host[idx].copy_(counts, non_blocking=True)      # sizes of the next arrays
            landed = torch.cuda.Event()
            landed.record()
            yield landed
            H = [int(v) for v in counts_host[idx].tolist()]
            blur_p = None
            blur_ptr = [None, None]
            corr1 = corr2 = None
            if bcn_r != -1:
                # one table for the pair: columns [0,H0) cloud 1, [H0,H0+H1) cloud 2 with ids shifted by H0
                F = _filter_size(bcn_r)
                blur_p = torch.empty((F, H[0] + H[1]), dtype=torch.int32, device=dev)
                blur_ptr = [ptr(blur_p), blur_p.data_ptr() + 4 * H[0]]
            if cf_r != -1:
                if cc_r != bcn_r:           # equal radii: corr1 is the blur table of cloud 1
                    corr1 = torch.empty((_filter_size(cc_r), H[0]), dtype=torch.int32, device=dev)
                corr2 = torch.empty((_filter_size(cc_r), _filter_size(cf_r) * H[0]), dtype=torch.int32, device=dev)
            check(L.hpl_lattice_neighbors(ptr(ws), n[0], n[1], ptr(vk[0]), ptr(vk[1]), H[0], H[1], int(bcn_r),
                                          int(cf_r), int(cc_r), blur_ptr[0], blur_ptr[1], H[0] + H[1], H[0],
                                          ptr(corr1), ptr(corr2), stream()), 'hpl_lattice_neighbors')
            lv = _Level()
            lv.H = (H[0], H[1])
            lv.clouds = [ops.CloudTables(bary[c], off[c], H[c]) for c in (0, 1)]
            lv.blur = PairBlur(blur_p, H[0]) if blur_p is not None else [None, None]
            if blur_p is not None:
                lv.blur[0].vertices_per_point = H[0] / float(n[0])
            lv.emg = emg
            lv.emg_pair = emg_p
            lv.pair = ops.PairTables(lv.clouds[0], lv.clouds[1])
            lv.corr1 = NbrTable(corr1) if corr1 is not None else None
            if cf_r != -1 and cc_r == bcn_r:
                lv.corr1 = lv.blur[0]          # same offsets, same table: share it
            lv.corr2 = NbrTable(corr2) if corr2 is not None else None
            if lv.corr2 is not None:
                lv.corr2._sym = False
            levels.append(lv)
            if idx != nlev - 1:
                div = float(np.float32(self.expected_std * scale))         # transforms.py:462-463
                prev = (vk, [4 * n[0], 4 * n[1]], div)
                n = [H[0], H[1]]
        return DeviceLattice(levels, wide_up=self.wide_up)

    def __call__(self, data):
        pc1, pc2, sf = data
        if pc1 is None:                                                    # transforms.py:360-361
            return None, None, None, None
        with torch.no_grad():
            t1 = torch.as_tensor(pc1).to(self.device).t().contiguous()
            t2 = torch.as_tensor(pc2).to(self.device).t().contiguous()
            tsf = torch.as_tensor(sf).to(self.device).t().contiguous()
            return t1, t2, tsf, self.build(t1, t2)

    def __repr__(self):
        return '%s\n(scales_filter_map: %s\n)' % (self.__class__.__name__, self.scales_filter_map)


class LatticeBuild(object):
    """One pair's lattice under construction on `stream`: advance() launches up to the next read-back
    and returns; `done` / `result` / `event` (recorded on `stream` after the last launch, incl.
    prepare()) once finished."""

    def __init__(self, gen, pc1, pc2, stream=None, for_training=False, prepare=True, tag=None):
        self.stream = stream if stream is not None else torch.cuda.current_stream(pc1.device)
        self.tag = tag
        self.done = False
        self.result = self.event = None
        self._pending = None
        self._steps = self._run(gen, pc1, pc2, for_training, prepare)

    def _run(self, gen, pc1, pc2, for_training, prepare):
        lat = yield from gen.build_steps(pc1, pc2)
        if prepare:
            lat.prepare_tables(for_training)
            if for_training:
                begun, landed = lat.symmetry_begin()
                if landed is not None:
                    yield landed
                lat.symmetry_finish(begun)
        return lat

    def ready(self):
        """True if advance() would not block."""
        return self.done or self._pending is None or self._pending.query()

    def advance(self):
        if self.done:
            return True
        if self._pending is not None:
            self._pending.synchronize()
        with torch.cuda.stream(self.stream), torch.no_grad():
            try:
                self._pending = next(self._steps)
            except StopIteration as fin:
                self.result, self.done, self._pending = fin.value, True, None
                self.event = torch.cuda.Event()
                self.event.record(self.stream)
        return self.done

    def finish(self):
        while not self.advance():
            pass
        return self.result


class LatticePipeline(object):
    """Lattices of consecutive pairs, up to `depth` under construction at once on one stream.

    `source(i)` -> (pc1, pc2) device tensors (3, N) of pair i; pairs first .. first+count-1 are built,
    each exactly once, and handed out in order by get() as ((i, source(i)), DeviceLattice, event); the
    consumer's stream must wait for `event`, and must keep the pair and the lattice referenced until its
    own work on them is done (they were allocated on the lattice stream).  get() resumes, round robin, each
    pair whose counts have already landed until the oldest is complete, and blocks only when no pair can
    move."""

    def __init__(self, gen, source, first, count, depth=2, stream=None, for_training=False, native=False, threaded=False):
        self.native = native            # builds driven by csrc/lattice_builder.hip (NativeLatticeBuild)
        self.gen, self.source, self.depth = gen, source, max(1, int(depth))
        # `stream` may be a list: consecutive pairs are built on alternating streams, so the launch-latency chains of
        # two builds (each ~1 ms of dependent small kernels) overlap on the GPU instead of queueing on one stream
        self.streams = list(stream) if isinstance(stream, (list, tuple)) else [stream]
        self.stream, self.for_training = self.streams[0], for_training
        self._next, self._end = first, first + count
        self._first, self._handed = first, 0
        self._inflight = collections.deque()
        # threaded: a producer thread runs the builds (the reference runs them in DataLoader worker processes,
        # main.py:85-92).  The native builder spends its time inside C calls, which ctypes makes with the GIL released, so
        # the consumer's forward enqueue (also one C call) overlaps it on a second core.
        self._thread = self._queue = None
        self._stop = False
        if threaded and count > 0:
            import queue
            import threading
            self._queue = queue.Queue(maxsize=self.depth)
            dev = torch.cuda.current_device()

            def put(item):
                while not self._stop:               # (a consumer that went away must not leave this thread blocked forever)
                    try:
                        self._queue.put(item, timeout=0.2)
                        return True
                    except queue.Full:
                        pass
                return False

            def produce():
                try:
                    torch.cuda.set_device(dev)
                    for _ in range(count):
                        c0 = time.thread_time()
                        item = self._get()
                        BUSY['s'] += time.thread_time() - c0           # CPU time of this thread (its waits for read-backs excluded)
                        if self._stop or not put(item):
                            return
                except BaseException as e:          # noqa: B902 -- handed to the consumer
                    put(e)
            self._thread = threading.Thread(target=produce, name='hpl-lattice', daemon=True)
            self._thread.start()

    def close(self):
        """Stop the producer thread (if any) and wait for it, then finish the builds still in flight and give their native
        builder handles back (each holds pinned staging memory and events): call when the consumer abandons the pipeline early."""
        self._stop = True
        if self._thread is not None:
            self._thread.join()              # (the thread leaves its loop at the next put(): _stop is set)
            self._thread = None
        while self._inflight:
            b = self._inflight.popleft()
            try:
                b.finish()                   # the launches of a native build were all enqueued by begin: this waits for one read-back
            except Exception:
                pass
            if getattr(b, 'handle', None) is not None:
                b.nb.release(b.handle)
                b.handle = None

    def _top_up(self):
        depth = self.depth
        if self.native:
            nb = self.gen.native_builder()
            if nb.fused and not any(nb.seen):
                # no pair observed yet: every level is bounded at 16 x the cloud size (~1 GB of arena per build at N = 8 192).
                # The first build runs alone; its counts tighten the bounds (~0.26 GB) before further pairs are enqueued.
                depth = 1
        while len(self._inflight) < depth and self._next < self._end:
            st = self.streams[self._next % len(self.streams)]
            with torch.cuda.stream(st):                 # a reader's host-to-device copies belong to this stream too
                item = self.source(self._next)
            cls = NativeLatticeBuild if self.native else LatticeBuild
            b = cls(self.gen, item[0], item[1], st, self.for_training, tag=(self._next, item))
            if not self.native:
                b.advance()
            self._inflight.append(b)
            self._next += 1

    def get(self):
        if self._queue is not None:
            if self._handed >= self._end - self._first:
                raise StopIteration('all %d lattices were handed out' % self._end)
            item = self._queue.get()
            if isinstance(item, BaseException):
                raise item
            self._handed += 1
            return item
        return self._get()

    def _get(self):
        self._top_up()
        if not self._inflight:
            raise StopIteration('all %d lattices were handed out' % self._end)
        head = self._inflight[0]
        while not head.done:
            # resume whoever has its counts, oldest first; launching those stages takes about as long as the
            # read-backs of the others, so the blocking advance of the head below is the rare case
            moved = False
            for b in self._inflight:
                if not b.done and b.ready():
                    b.advance()
                    moved = True
            if not moved:
                t = time.perf_counter()
                while not head.ready():          # the GPU has not delivered the head's vertex counts yet: idle host time
                    pass
                WAIT['s'] += time.perf_counter() - t
                head.advance()
        self._inflight.popleft()
        self._top_up()
        return head.tag, head.result, head.event



def to_reference_format(lat):
    """DeviceLattice -> the reference's generated_data (transforms.py:471-483): list of dicts of
    int64 / float32 tensors (on the device), absent tables as zeros(1)."""
    out = []
    for lv in lat.levels:
        d = {}
        for c, nm in enumerate(('pc1', 'pc2')):
            cl = lv.clouds[c]
            d[nm + '_barycentric'] = cl.bary
            d[nm + '_el_minus_gr'] = lv.emg[c].t().contiguous()
            d[nm + '_lattice_offset'] = cl.off.long()
            d[nm + '_blur_neighbors'] = lv.blur[c].t.long() if lv.blur[c] is not None else \
                torch.zeros(1, dtype=torch.long, device=cl.bary.device)
            d[nm + '_hash_cnt'] = lv.H[c]
        if lv.corr1 is not None:
            K = lv.corr1.t.shape[0]
            H1 = lv.H[0]
            F = lv.corr2.t.shape[1] // H1
            d['pc1_corr_indices'] = lv.corr1.t.long()
            d['pc2_corr_indices'] = lv.corr2.t.view(K, F, H1).permute(1, 0, 2).contiguous().long()
        else:
            z = torch.zeros(1, dtype=torch.long, device=lv.clouds[0].bary.device)
            d['pc1_corr_indices'] = z
            d['pc2_corr_indices'] = z.clone()
        out.append(d)
    return out


# ----------------------------------------------------------------------------- native builder
HPL_ENOMEM = -4


class NativeLattice(object):
    """A lattice built by the native builder (csrc/lattice_builder.hip): the kernel-ready hpl_level_tables of every
    level, all pointing into ONE device arena.  The native forward (plan.ForwardPlan) consumes `tables` as is;
    `.levels` / `.prepare()` materialise the DeviceLattice view of the same memory (zero-copy tensors over the
    arena) for everything else -- training, the reference wire format, the parity tests."""

    def __init__(self, arena, tables, n_levels, extras, wide_up):
        self.arena, self.tables, self.n_levels, self.extras, self.wide_up = arena, tables, n_levels, extras, wide_up
        self._native_tables = (tables, n_levels, [arena])
        self._view = None

    @property
    def H(self):
        return [(int(t.H0), int(t.H1)) for t in self.tables[:self.n_levels]]

    def _t(self, ptr_, dtype, shape):
        if not ptr_:
            return None
        off = ptr_ - self.arena.data_ptr()
        n = int(np.prod(shape)) * 4
        return self.arena[off:off + n].view(dtype).view(shape)

    def device_lattice(self):
        """DeviceLattice over the arena (no copies, no launches)."""
        if self._view is not None:
            return self._view
        levels = []
        for L in range(self.n_levels):
            t = self.tables[L]
            n0, n1, H0, H1 = int(t.n0), int(t.n1), int(t.H0), int(t.H1)
            Hp, Np = H0 + H1, n0 + n1
            lv = _Level()
            lv.H = (H0, H1)
            emg_p = self._t(t.emg_pair, torch.float32, (Np, 4))
            c0 = ops.CloudTables(self._t(t.bary0, torch.float32, (4, n0)), self._t(t.off0, torch.int32, (4, n0)), H0)
            c1 = ops.CloudTables(self._t(self.extras[2 * L], torch.float32, (4, n1)),
                                 self._t(self.extras[2 * L + 1], torch.int32, (4, n1)), H1)
            lv.clouds = [c0, c1]
            lv.emg_pair, lv.emg = emg_p, [emg_p[:n0], emg_p[n0:]]
            lv.pair = ops.PairTables(c0, c1)
            csr = (self._t(t.csr_ptr, torch.int32, (Hp + 1,)), self._t(t.csr_pt, torch.int32, (4 * Np,)),
                   self._t(t.csr_w, torch.float32, (4 * Np,)), self._t(t.csr_norm, torch.float32, (Hp,)))
            lv.pair._csr = csr
            c0._csr = (csr[0][:H0 + 1], csr[1][:4 * n0], csr[2][:4 * n0], csr[3][:H0])
            c1._csr_src = (csr, n0, H0)
            if t.blur:
                F = 15
                blur_p = self._t(t.blur, torch.int32, (F, Hp))
                lv.blur = PairBlur(blur_p, H0)
                lv.blur[0].vertices_per_point = H0 / float(n0)
                pb, up = lv.blur.pair, lv.blur[0]
                if t.blur_perm:
                    tiles = (Hp + 63) // 64
                    pb._perm = self._t(t.blur_perm, torch.int32, (Hp,))
                    pb._perm_tiles = (self._t(t.blur_perm_tidx, torch.int32, (tiles, F, 64)),
                                      self._t(t.blur_perm_tmask, torch.int32, (tiles, 8)))
                elif Hp < NbrTable.PERM_MIN_ROWS:
                    pb._perm = None
                tiles0 = (H0 + 63) // 64
                if t.n_up_groups >= 2:
                    cuts = list(t.up_group_cut[:t.n_up_groups + 1])
                    up._groups = [(cuts[g], cuts[g + 1], self._t(t.up_group_perm[g], torch.int32, (H0,)))
                                  for g in range(t.n_up_groups)]
                    gbm = int(t.group_tile_bm) or 64          # (128: the split-operand kernel's tile height)
                    tg = (H0 + gbm - 1) // gbm
                    up._group_tiles = [(self._t(t.up_group_tidx[g], torch.int32, (tg, cuts[g + 1] - cuts[g], gbm)),
                                       self._t(t.up_group_tmask[g], torch.int32, (tg, 8))) for g in range(t.n_up_groups)]
                if t.up_perm:
                    up._perm = self._t(t.up_perm, torch.int32, (H0,))
                    up._perm_tiles = (self._t(t.up_perm_tidx, torch.int32, (tiles0, F, 64)),
                                      self._t(t.up_perm_tmask, torch.int32, (tiles0, 8)))
                elif H0 < NbrTable.PERM_MIN_ROWS:
                    up._perm = None
            else:
                lv.blur = [None, None]
            if t.corr2:
                K = 15
                if t.corr1 == t.blur:
                    lv.corr1 = lv.blur[0]
                else:
                    lv.corr1 = NbrTable(self._t(t.corr1, torch.int32, (K, H0)))
                lv.corr2 = NbrTable(self._t(t.corr2, torch.int32, (K, 15 * H0)))
                lv.corr2._sym = False
            else:
                lv.corr1 = lv.corr2 = None
            levels.append(lv)
        self._view = DeviceLattice(levels, wide_up=self.wide_up)
        self._view._native_tables = self._native_tables
        self._view._arena = self.arena
        return self._view

    @property
    def levels(self):
        return self.device_lattice().levels

    def prepare(self, for_training=False):
        return self.device_lattice().prepare(for_training)


class NativeBuilder(object):
    """Pool of native builder handles for one GenerateDataUnsymmetric configuration (one handle per pair under
    construction).  radius-1 stencils only (what the shipped configs use); anything else keeps the Python driver."""

    def __init__(self, gen):
        from ._lib import LatticeSpec
        self.gen = gen
        self.lib = _lib.load()
        sfm = gen.scales_filter_map
        n = len(sfm)
        if n > 8 or any(int(r) not in (-1, 1) for lvl in sfm for r in lvl[1:]):
            raise _lib.HplError('the native lattice builder handles radius 1 (or -1) and at most 8 levels')
        sp = LatticeSpec()
        sp.n_levels = n
        hint = gen.wide_up
        for L, (scale, b_r, cf_r, cc_r) in enumerate(sfm):
            sp.scale[L], sp.bcn_radius[L] = float(scale), int(b_r)
            sp.corr_filter_radius[L], sp.corr_corr_radius[L] = int(cf_r), int(cc_r)
            sp.next_divisor[L] = float(np.float32(gen.expected_std * scale))
            w = hint[L] if isinstance(hint, (list, tuple)) else hint
            sp.wide_up[L] = -1 if w is None else int(bool(w))
        G = NbrTable.TAP_GROUPS
        F = 15
        sp.n_groups = G if G >= 2 else 0
        for i in range(G + 1 if G >= 2 else 0):
            sp.group_cut[i] = round(i * F / G)
        sp.groups_min_sparsity = NbrTable.GROUPS_MIN_SPARSITY
        sp.perm_min_rows = NbrTable.PERM_MIN_ROWS
        sp.groups_min_rows = NbrTable.GROUPS_MIN_ROWS
        sp.group_tile_bm = ops.GROUP_TILE_BM
        # fused driver (csrc/lattice_fused.hip): the whole build enqueued by hpl_lattice_begin, one read-back per pair.
        # HPL_LATTICE_FUSED=0 keeps the staged driver (one read-back per level); specs it cannot build stay staged too.
        self.fused = os.environ.get('HPL_LATTICE_FUSED', '1') != '0' and \
            all(int(lvl[1]) == 1 and (int(lvl[2]), int(lvl[3])) in ((-1, -1), (1, 1)) for lvl in sfm)
        sp.fused = 1 if self.fused else 0
        self.spec = sp
        self.free = []
        self.bytes_per_point = 6000            # arena hint of the staged driver, doubled on HPL_ENOMEM
        self.n_levels = n
        # fused builds: per-level vertex bounds (per cloud) = twice the largest count seen so far, rounded up to a power of
        # two (so that the arena layout changes rarely); 0 = the library's default of 16 x the cloud size
        self.bounds = [0] * 8
        self.seen = [0] * 8
        self.fallbacks = 0

    def acquire(self):
        if self.free:
            return self.free.pop()
        h = self.lib.hpl_lattice_create(ctypes.byref(self.spec))
        if not h:
            raise _lib.HplError('hpl_lattice_create: %s' % self.lib.hpl_last_error().decode())
        return h

    def observe(self, counts):
        """Vertex counts of a finished pair -> the bounds of the next fused builds."""
        for L, (h0, h1) in enumerate(counts):
            m = max(h0, h1)
            if m > self.seen[L]:
                self.seen[L] = m
                want = 1 << max(10, (2 * m - 1).bit_length())
                if want > self.bounds[L] or self.bounds[L] == 0:
                    self.bounds[L] = want

    def release(self, h):
        self.free.append(h)

    def __del__(self):
        try:
            for h in self.free:
                self.lib.hpl_lattice_destroy(h)
        except Exception:
            pass


class NativeLatticeBuild(object):
    """LatticeBuild driven by the native builder: same ready() / advance() / finish() protocol, `result` is a
    NativeLattice.  Host cost of a pair's 7 levels: the launches themselves, no per-stage Python."""

    def __init__(self, gen, pc1, pc2, stream=None, for_training=False, prepare=True, tag=None):
        self.gen, self.tag = gen, tag
        self.nb = gen.native_builder()
        self.stream = stream if stream is not None else torch.cuda.current_stream(pc1.device)
        self.pc = (pc1.contiguous().float(), pc2.contiguous().float())
        self.for_training = for_training
        self.done = False
        self.result = self.event = None
        self.handle = self.nb.acquire()
        self._begin()

    def _begin(self):
        n0, n1 = int(self.pc[0].shape[1]), int(self.pc[1].shape[1])
        lib = self.nb.lib
        if self.nb.fused:
            arr = (ctypes.c_int64 * 8)(*self.nb.bounds)
            check(lib.hpl_lattice_set_bounds(self.handle, arr), 'hpl_lattice_set_bounds')
        while True:
            nbytes = (32 << 20) + self.nb.bytes_per_point * (n0 + n1)
            if self.nb.fused:
                # the fused layout is known up front; the staged fallback (a pair that outgrew a bound) reuses the arena
                nbytes = max(nbytes, int(lib.hpl_lattice_arena_bytes(self.handle, n0, n1)) + 4096)
            with torch.cuda.stream(self.stream):
                self.arena = torch.empty(nbytes, dtype=torch.uint8, device=self.pc[0].device)
                rc = self.nb.lib.hpl_lattice_begin(self.handle, ptr(self.pc[0]), ptr(self.pc[1]), n0, n1,
                                                   self.arena.data_ptr(), nbytes, stream())
            if rc != HPL_ENOMEM:
                check(rc, 'hpl_lattice_begin')
                return
            self.nb.bytes_per_point *= 2

    def ready(self):
        return self.done or bool(self.nb.lib.hpl_lattice_ready(self.handle))

    def advance(self):
        if self.done:
            return True
        d = ctypes.c_int(0)
        with torch.cuda.stream(self.stream), torch.no_grad():
            rc = self.nb.lib.hpl_lattice_advance(self.handle, ctypes.byref(d))
            if rc == HPL_ENOMEM:               # the arena overflowed at this level: start over with a bigger one
                self.nb.bytes_per_point *= 2
                self._begin()
                return False
            check(rc, 'hpl_lattice_advance')
            if d.value:
                from ._lib import LevelTables
                n = self.nb.n_levels
                src = self.nb.lib.hpl_lattice_tables(self.handle)
                arr = (LevelTables * n)()
                ctypes.memmove(arr, src, ctypes.sizeof(LevelTables) * n)
                extras = (ctypes.c_void_p * (2 * n))()
                used = ctypes.c_int64(0)
                check(self.nb.lib.hpl_lattice_extras(self.handle, extras, ctypes.byref(used)), 'hpl_lattice_extras')
                if self.nb.fused:
                    st = (ctypes.c_int32 * 3)()
                    self.nb.lib.hpl_lattice_stats(self.handle, st)
                    self.nb.launches = int(st[0])
                    self.nb.fallbacks += 0 if st[1] else 1
                    first = not any(self.nb.seen)
                    self.nb.observe([(int(arr[L].H0), int(arr[L].H1)) for L in range(n)])
                    if first:
                        torch.cuda.empty_cache()       # (the loose first arena's block would otherwise stay reserved beside the tight ones)
                self.nb.release(self.handle)
                self.handle = None
                lat = NativeLattice(self.arena, arr, n, [int(e or 0) for e in extras], self.gen.wide_up)
                lat.arena_used = used.value
                if self.for_training:
                    lat = lat.device_lattice().prepare(True)        # tap lists, symmetry verdicts: the Python tables
                self.result, self.done = lat, True
                self.event = torch.cuda.Event()
                self.event.record(self.stream)
        return self.done

    def finish(self):
        while not self.advance():
            pass
        return self.result
