"""Native training step: forward + EPE3D loss + the whole backward of HPLFlowNet / HPLFlowNetShallow as ONE program of the native
executor (csrc/executor.hip, `hpl_plan_run_range`) instead of ~275 autograd-driven Python calls (ops.GConvFn / SplatFn / SliceFn).

Reference: the training loop main.py:203-217 (`loss.backward()` over models/HPLFlowNet.py:238-430).  What autograd derives there is
written down here once per model, op by op, from the forward program of plan.build_program:

  gather-GEMM  Y = act(b + res + sum_f A[nbr_f] W_f)   ->  g' = g * act'(Y);  d res = g' (summed over the blocks a broadcast residual
               was repeated over);  dW_f = A[nbr_f]^T g', db = colsum(g') (hpl_gconv_wgrad: leaves of the graph, side stream);
               dA = the same gather-GEMM with mirrored taps (symmetric tables: blur, corr1), a GEMM + atomic scatter through the
               table (corr2), a GEMM + row regrouping (the displacement filter's regular pattern), or a plain GEMM (1x1 convs)
  splat        -> slice with the normaliser as vertex scale;      slice + bias -> un-normalised splat, db = colsum

Gradients of a matrix with several consumers are accumulated in place: the first writer stores, later ones add (decided here, when the
program is written, by replaying it on small stand-in sizes).  Weight gradients land in images laid out like the forward weight
images and are un-laid into ONE flat gradient arena (the parameters' .grad are views of it) by one batched launch per all-reduce
bucket, so the bucketed all-reduce (parallel.GradAllReducer) needs no packing copies and can start while the rest of the backward runs.
"""
import ctypes

import numpy as np
import torch

from . import _lib, ops
from ._lib import LevelTables, RelayoutJob, check, ptr, stream
from .bcl import _ConvReLU
from .plan import (BUF_OUT, ForwardPlan, ORD_NONE, OP_GCONV, OP_GSUM, OP_INVERT, S_FH0, S_FH1, S_H0, S_H1, S_HP, S_IN0, S_INP, SYM_N0, SYM_N1,
                   SYM_NP, SYM_LEVEL0, SYM_ZERO, TBL_BLUR0, TBL_BLUR_PAIR, TBL_CORR1, TBL_CORR2, TBL_CSR_C0, TBL_CSR_PAIR, TBL_NONE,
                   TBL_REGULAR, _R, build_program, level_tables, lsym)

OP_WGRAD, OP_LEAKY_BWD, OP_COLSUM, OP_SPLAT_BWD, OP_PSUM, OP_REGROUP, OP_ZERO, OP_EPE3D, OP_VCOPY, OP_UNLAYOUT = range(6, 16)
OP_SPLAT, OP_COPY = 2, 4
F_ACCUM, F_SCATTER, F_TAPS, F_SIDE, F_INVERSE, F_NOGUARD = 1, 2, 4, 8, 16, 32


class _Sim(object):
    """Which cells of every gradient matrix have been written, replayed on stand-in sizes (distinct small numbers that keep
    the relations between the symbols: HP = H0 + H1, FH0 = 15 H0, IN0(L) = H0(L-1)) -- for both outcomes of the per-level
    SHRINK condition (the two alternative op sequences of an Up layer write the same cells)."""

    def __init__(self, P, nlev):
        self.P = P
        sym = {SYM_ZERO: 0, SYM_N0: 3, SYM_N1: 4, SYM_NP: 7}
        n0, n1 = 3, 4
        for L in range(nlev):
            h0, h1 = 5 + 2 * L, 6 + 2 * L
            for k, v in ((S_H0, h0), (S_H1, h1), (S_HP, h0 + h1), (S_FH0, 15 * h0), (S_IN0, n0), (S_INP, n0 + n1), (S_FH1, 15 * h1)):
                sym[lsym(L, k)] = v
            n0, n1 = h0, h1
        self.sym = sym
        self.mask = [{}, {}]         # scenario (0: every level shrinks, 1: none does) -> buffer -> bool [rows, cols]

    def _cells(self, sc, r):
        rows, cols = self.P.bufs[r.buf]
        m = self.mask[sc].get(r.buf)
        if m is None:
            m = self.mask[sc][r.buf] = np.zeros((self.sym[rows], cols), dtype=bool)
        r0 = self.sym[r.row_off]
        r1 = m.shape[0] if r.rows == SYM_ZERO else r0 + self.sym[r.rows]
        assert r1 <= m.shape[0] and r.col_off + r.cols <= cols, 'reference outside its buffer'
        return m[r0:r1, r.col_off:r.col_off + r.cols]

    @staticmethod
    def _scenarios(cond):
        return (0, 1) if cond[0] == 0 else ((0,) if cond[0] == 1 else (1,))

    def written(self, r, cond):
        """True / False if the cells are all / none written (anything else is a bug in the emitter)."""
        res = set()
        for sc in self._scenarios(cond):
            c = self._cells(sc, r)
            if c.size:
                res.add(bool(c.all()) if c.all() or not c.any() else None)
        assert None not in res and len(res) == 1, 'gradient region partly written: %r' % (res,)
        return res.pop()

    def mark(self, r, cond):
        for sc in self._scenarios(cond):
            self._cells(sc, r)[...] = True

    def claim(self, r, cond):
        """-> accumulate?  and marks the cells written."""
        acc = self.written(r, cond)
        self.mark(r, cond)
        return acc

    def periods(self, M, mod):
        return self.sym[M] // self.sym[mod]


class _Backward(object):
    def __init__(self, P, model, params):
        self.P, self.model = P, model
        self.sim = _Sim(P, model.NLEV)
        self.gbuf = {}                 # forward buffer -> gradient buffer
        self.pid = {id(p): i for i, p in enumerate(params)}
        self.params = params
        self.gvec = {}                 # parameter index -> bias index of its gradient vector (resolved to a pointer later)
        self.gimg = {}                 # forward weight key -> (weight index of the gradient image, (param, C, O, F, Ctot, c0))
        self.ready = {}                # parameter index -> index of the last op that writes its gradient
        self.touched = [set(), set()]  # per scenario of the SHRINK condition: parameters whose gradient an emitted op writes
        self.gout = None
        self.inv = {}                  # level -> buffer of the inverse pc2 correlation table

    # ---- references
    def G(self, r):
        if r.buf == BUF_OUT:
            return self.gout
        g = self.gbuf.get(r.buf)
        if g is None:
            rows, cols = self.P.bufs[r.buf]
            g = self.gbuf[r.buf] = self.P.buf(rows, cols).buf
        return _R(g, r.cols, r.col_off, r.row_off, r.rows)

    def _gvec(self, p):
        i = self.pid[id(p)]
        if i not in self.gvec:
            self.P.biases.append(('gvec', i))
            self.gvec[i] = len(self.P.biases) - 1
        return self.gvec[i]

    def _touch(self, p):
        self.ready[self.pid[id(p)]] = len(self.P.ops) - 1
        for sc in _Sim._scenarios(self.P.cond):
            self.touched[sc].add(self.pid[id(p)])

    def _stores(self, p):
        """The op just emitted STORES into p's gradient vector (OP_COLSUM zeroes its target, OP_VCOPY copies over it), the weight
        gradients ACCUMULATE into the zeroed arena: a parameter that feeds two forward ops that both run (a shared module, a bias
        used by a slice and a conv) would silently keep only the last store.  No model here shares one; refuse instead of being
        wrong.  (The two alternative op sequences of an Up layer touch the same parameters under OPPOSITE conditions: only one of
        them runs, so contributions count per scenario, as in _Sim.)"""
        i = self.pid[id(p)]
        for sc in _Sim._scenarios(self.P.cond):
            if i in self.touched[sc]:
                raise _lib.HplError('native backward: parameter %d receives a second gradient contribution through a storing op '
                                    '(shared parameters are not supported by TrainPlan; use the autograd path)' % i)
        self._touch(p)

    def _bank(self, w, R, Q, F, sr, sq, sf, base, mirror):
        self.P.bank.register(w.detach(), R, Q, F, sr, sq, sf, base, mirror)
        self.P.weights.append((w, R, Q, F, sr, sq, sf, base, mirror))
        return len(self.P.weights) - 1

    # ---- the program
    def emit(self):
        P = self.P
        fwd = list(zip(P.ops, P.meta))
        self.gout = P.buf(SYM_N0, 3)
        P.cond = (0, 0)
        P.raw(OP_EPE3D, out=self.gout)
        self.sim.mark(self.gout, (0, 0))
        for op, m in reversed(fwd):
            if m is None:
                continue
            P.cond = m['cond']
            getattr(self, '_' + m['kind'])(m)
        P.cond = (0, 0)

    def _grad_image(self, m, cols=False):
        """weight index of the gradient image of the forward weight image m['wid'] (one per distinct image)."""
        key = self.P.weights[m['wid']]
        kid = (id(key[0]),) + tuple(key[1:])
        if kid not in self.gimg:
            self.P.weights.append(('grad', len(self.gimg)))
            self.gimg[kid] = (len(self.P.weights) - 1, self.P.wmeta[m['wid']], cols)
        return self.gimg[kid][0]

    def _zproj(self, m):
        """Z = f2 . [W_0 | ... | W_{K-1}] (plan._corr: the per-tap projection of the patch correlation): dense GEMM with the taps as
        column blocks.  dW image [C][K*O] (un-laid with mirror = 2), d f2 = dZ . [(k, o) x c] image."""
        P, sim, cond = self.P, self.sim, m['cond']
        w, C, O, F, Ctot, c0 = P.wmeta[m['wid']]
        M, a = m['M'], m['a']
        g = self.G(m['out'])
        assert sim.written(g, cond) and m['res'] is None and not m['act'] and m['bias'] < 0
        P.raw(OP_WGRAD, a=a, b=g, M=M, level=m['level'], table=TBL_NONE, F=1, C=C, N=F * O, weight=self._grad_image(m, True), flags=F_SIDE)
        self._touch(w)
        ga = self.G(a)
        wid = self._bank(w, O, C, F, Ctot * F, F, 1, c0 * F, 0)               # rows (k*O + o), columns c
        P.gconv(g, ga, M, F * O, C, wid, flags=F_NOGUARD | (F_ACCUM if sim.claim(ga, cond) else 0))

    def _gsum(self, m):
        P, sim, cond = self.P, self.sim, m['cond']
        M, K, N, L = m['M'], m['K'], m['N'], m['level']
        g = self.G(m['out'])
        assert sim.written(g, cond)
        if m['act']:
            P.raw(OP_LEAKY_BWD, a=g, b=m['out'], out=g, M=M, N=N, slope=m['slope'])
        if m['res'] is not None:
            gr = self.G(m['res'])
            P.raw(OP_PSUM, a=g, out=gr, M=m['res_mod'], F=sim.periods(M, m['res_mod']), N=N, flags=F_ACCUM if sim.claim(gr, cond) else 0)
        if m['bias'] >= 0:
            b = P.biases[m['bias']]
            P.raw(OP_COLSUM, a=g, M=M, C=N, bias=self._gvec(b), flags=F_SIDE)
            self._stores(b)
        # dZ[(v*K + k)] = sum_f g[inverse[f][v*K + k]]: the same gather-sum through the inverse table (built once per step and level)
        inv = self.inv.get(L)
        if inv is None:
            inv = self.inv[L] = P.buf(lsym(L, S_H1), 15 * K)
            P.raw(OP_INVERT, out=inv, level=L)
        gz = self.G(m['a'])
        assert not sim.claim(gz, cond)
        P.raw(OP_GSUM, a=g, b=inv, out=gz, M=lsym(L, S_FH1), level=L, F=sim.periods(M, m['res_mod']) if m['res'] is not None else 15, C=N, N=N,
              flags=F_INVERSE)

    def _gconv(self, m):
        if m.get('wcols'):
            return self._zproj(m)
        P, sim, cond = self.P, self.sim, m['cond']
        M, N, F = m['M'], m['N'], m['F']
        w, C, O, _, Ctot, c0 = P.wmeta[m['wid']]
        out, a = m['out'], m['a']
        g = self.G(out)
        if g.rows == SYM_ZERO:
            g = g.rows_from(g.row_off, M)
        if m['out2'] is not None:                     # the same values also went to a second matrix: its gradient joins
            g2 = self.G(m['out2'])
            head = g.rows_from(g.row_off, m['rows2'])
            P.copy(g2, head, m['rows2'], N, flags=F_ACCUM if sim.claim(head, cond) else 0)
        assert sim.written(g, cond), 'gradient of a gconv output read before it exists'
        if m['act']:
            P.raw(OP_LEAKY_BWD, a=g, b=out, out=g, M=M, N=N, slope=m['slope'])
        if m['res'] is not None:
            gr = self.G(m['res'])
            per = 1 if m['res_mod'] in (SYM_ZERO, M) else sim.periods(M, m['res_mod'])
            acc = F_ACCUM if sim.claim(gr, cond) else 0
            if per == 1:
                P.copy(g, gr, M, N, flags=acc)
            else:
                P.raw(OP_PSUM, a=g, out=gr, M=m['res_mod'], F=per, N=N, flags=acc)
        # weight (and bias) gradient: a leaf
        gw = self._grad_image(m)
        bias_t = P.biases[m['bias']] if m['bias'] >= 0 else None
        pair = next((ab for t, ab in P.combined if t is bias_t), None) if bias_t is not None else None
        bparam = pair[0] if pair is not None else bias_t
        # (Issuing the wide layers' weight gradients later -- beside the launch-bound chain of the coarse levels instead of beside the
        # equally wide data gradients -- was measured: the small launches then queue behind the wide tiles for CU slots, 13.3 -> 14.0 ms.)
        P.raw(OP_WGRAD, a=a, b=g, M=M, level=m['level'], table=m['table'], F=F, C=C, N=N, weight=gw,
              bias=self._gvec(bparam) if bparam is not None else -1, reg_stride=m['reg_stride'],
              flags=F_SIDE | (F_TAPS if m['table'] == TBL_BLUR0 else 0))
        self._touch(w)
        if bparam is not None:
            self._touch(bparam)
        if pair is not None:                          # conv bias + layer bias were one vector in the forward: one gradient, two owners
            P.raw(OP_VCOPY, weight=self._gvec(pair[0]), bias=self._gvec(pair[1]), N=N, flags=F_SIDE)
            self._stores(pair[1])
        # data gradient
        if a.buf == self.leaf:
            return
        assert a.cols == C, (a.cols, C)
        ga = self.G(a)
        base, sr, sq = c0 * F, Ctot * F, F
        tbl = m['table']
        if tbl == TBL_NONE:
            wid = self._bank(w, O, C, 1, sr, sq, 1, base, 0)
            P.gconv(g, ga, M, O, C, wid, flags=F_NOGUARD | (F_ACCUM if sim.claim(ga, cond) else 0))
        elif tbl in (TBL_BLUR_PAIR, TBL_BLUR0, TBL_CORR1):          # symmetric table over the same vertex set: mirrored taps
            wid = self._bank(w, O, C, F, sr, sq, 1, base, 1)
            P.gconv(g, ga, M, O, C, wid, F=F, level=m['level'], table=tbl, order=m['order'], tag=m['tag'],
                    flags=F_NOGUARD | (F_ACCUM if sim.claim(ga, cond) else 0))
        elif tbl == TBL_CORR2:                                      # G[m, (f, c)] = g[m] . W[:, c, f], added into row corr2[f, m]
            wid = self._bank(w, O, C, F, sr, sq, 1, base, 2)
            if not sim.claim(ga, cond):
                P.raw(OP_ZERO, out=ga)
            P.gconv(g, ga, M, O, F * C, wid, level=m['level'], flags=F_SCATTER | F_NOGUARD, aux=C)
        elif tbl == TBL_REGULAR:                                    # rows f*M + m are distinct: a GEMM and a regrouping
            wid = self._bank(w, O, C, F, sr, sq, 1, base, 2)
            tmp = P.buf(M, F * C)
            P.gconv(g, tmp, M, O, F * C, wid, flags=F_NOGUARD)
            P.raw(OP_REGROUP, a=tmp, out=ga, M=M, F=F, C=C, flags=F_ACCUM if sim.claim(ga, cond) else 0)
        else:
            raise AssertionError('gconv through table kind %d has no backward' % tbl)

    def _splat(self, m):
        g = self.G(m['out'])
        assert self.sim.written(g, m['cond'])
        ga = self.G(m['a']).columns(0, m['C'])
        self.P.raw(OP_SPLAT_BWD, a=g, out=ga, level=m['level'], table=m['table'], C=m['C'], use_norm=m['use_norm'],
                   flags=F_ACCUM if self.sim.claim(ga, m['cond']) else 0)

    def _slice(self, m):
        P, cond = self.P, m['cond']
        g = self.G(m['out'])
        assert self.sim.written(g, cond)
        ga = self.G(m['a']).columns(0, m['C'])
        P.splat(g, ga, m['level'], TBL_CSR_C0, lsym(m['level'], S_H0), m['C'], False, flags=F_ACCUM if self.sim.claim(ga, cond) else 0)
        P.meta[-1] = None
        if m['bias'] >= 0:
            b = P.biases[m['bias']]
            P.raw(OP_COLSUM, a=g, M=m['N'], C=m['C'], bias=self._gvec(b), flags=F_SIDE)
            self._stores(b)


class TrainPlan(ForwardPlan):
    """One model's training step as a native program.  `step(pc1, pc2, sf, lattice)` enqueues forward, loss, backward and the
    un-layout of every weight gradient; the parameters' `.grad` are views of `self.gflat` (do NOT call zero_grad(set_to_none=True)
    on them).  reducer: a parallel.GradAllReducer over the same parameters (its bucket partition lays out gflat; with more than one
    rank a bucket's all-reduce starts as soon as the program has produced it)."""

    def __init__(self, model, reducer=None, side_stream=True):
        self.params = [p for p in model.parameters()]
        if any(not p.requires_grad for p in self.params):
            raise _lib.HplError('the native training step expects every parameter to require a gradient')
        from . import parallel
        self.reducer = reducer if reducer is not None else parallel.GradAllReducer(self.params, overlap=False)
        dev = self.params[0].device
        # flat gradient arena in bucket order; .grad = views
        order = [p for b in self.reducer.buckets for p in b]
        assert len(order) == len(self.params)
        # (every vector starts on a 16-byte boundary: the parameters live in an array of the same layout -- below -- and the kernels
        # read biases and weights as 16-byte words; the <= 3 padding floats behind a vector stay zero in every array)
        self.gflat = torch.zeros(sum((p.numel() + 3) // 4 * 4 for p in order), dtype=torch.float32, device=dev)
        self._goff, self._bucket_of, o = {}, {}, 0
        self.bucket_range = []
        for bi, b in enumerate(self.reducer.buckets):
            o0 = o
            for p in b:
                self._goff[id(p)] = o
                self._bucket_of[id(p)] = bi
                p.grad = self.gflat[o:o + p.numel()].view(p.shape)
                o += (p.numel() + 3) // 4 * 4
            self.bucket_range.append((o0, o))
        self.reducer.adopt_flat(self.gflat, self.bucket_range)
        # The parameters themselves move into ONE flat array of the same order (their .data become views of it, values kept; done
        # before the program below takes any pointer): with the moments of the optimiser laid out likewise, an Adam step is one
        # launch over four flat arrays (adam_step) instead of torch's four multi-tensor launches at a quarter of the bandwidth.
        self.pflat = torch.zeros_like(self.gflat)
        with torch.no_grad():
            for p in order:
                o = self._goff[id(p)]
                v = self.pflat[o:o + p.numel()].view(p.shape)
                v.copy_(p.data)
                p.data = v
        ops.invalidate_weight_cache()          # (images cached for the old storage)
        # the weight images are (re)made on other streams: the copies above must have landed before any of them reads a parameter
        torch.cuda.current_stream(dev).synchronize()
        self._adam = None
        self.loss = torch.zeros(1, dtype=torch.float32, device=dev)
        # The side stream is a HIGH-priority stream: HIP maps the normal-priority streams of a process round robin onto 4 hardware
        # queues, and a process that has made a few (bench.py: three forward streams) gets a side stream that SHARES the queue of
        # the stream the step runs on -- every fence then drains that queue (measured inside bench.py: 22.9 ms per step instead of
        # 13.0; with GPU_MAX_HW_QUEUES=8: 13.2).  High-priority streams have queues of their own.  side_stream may also be a
        # torch.cuda.Stream of the caller's choosing, or False / None for none.
        if isinstance(side_stream, torch.cuda.Stream):
            self._side = side_stream
        else:
            self._side = torch.cuda.Stream(device=dev, priority=-1) if side_stream else None
        super(TrainPlan, self).__init__(model)
        self._finish_unlayout()

    # ---- program: forward (plan.build_program) + loss + backward, then the un-layout ops where their buckets are complete
    def _program(self, model):
        P = build_program(model, self.bank)
        self.n_fwd = len(P.ops)
        self._n_fwd_jobs, self._n_fwd_weights = len(self.bank.jobs), len(P.weights)      # what the forward needs of the bank
        B = _Backward(P, model, self.params)
        B.leaf = next(o.out.buf for o in P.ops if o.kind == 5)          # the stacked input clouds (HPL_OP_LOAD): no gradient
        B.emit()
        self._B = B
        # gradient vectors -> pointers into the arena
        for i, b in enumerate(P.biases):
            if isinstance(b, tuple):
                p = self.params[b[1]]
                P.biases[i] = self.gflat.data_ptr() + 4 * self._goff[id(p)]
        # un-layout: one op per bucket, right behind the last op that writes one of its gradients
        nb = len(self.reducer.buckets)
        ready = [self.n_fwd] * nb
        for pi, at in B.ready.items():
            bi = self._bucket_of[id(self.params[pi])]
            ready[bi] = max(ready[bi], at)
        self.bucket_order = sorted(range(nb), key=lambda b: ready[b])
        ops_, meta = list(P.ops), list(P.meta)
        P.cond = (0, 0)
        for k, bi in enumerate(reversed(self.bucket_order)):             # (back to front: earlier positions stay valid)
            P.raw(OP_UNLAYOUT, aux=self.bucket_order.index(bi), flags=F_SIDE)
            op = P.ops.pop()
            P.meta.pop()
            ops_.insert(ready[bi] + 1, op)
            meta.insert(ready[bi] + 1, None)
        P.ops, P.meta = ops_, meta
        self.cuts = [i + 1 for i, o in enumerate(P.ops) if o.kind == OP_UNLAYOUT]          # op ranges that end with an un-layout
        # gradient images: laid out in un-layout order (bucket by bucket)
        keys = sorted(B.gimg.values(), key=lambda v: self.bucket_order.index(self._bucket_of[id(v[1][0])]))
        self._gimg_keys = keys
        off, self._gimg_off = 0, {}
        for widx, (w, C, O, F, Ctot, c0), cols in keys:
            rows, ldw = (ops.round_up(C, 32), ops.round_up(F * O, 4)) if cols else (ops.round_up(F * C, 32), ops.round_up(O, 4))
            self._gimg_off[widx] = (off, rows, ldw)
            off += rows * ldw
        self.gimg = torch.zeros(max(1, off), dtype=torch.float32, device=self.gflat.device)
        self._grad_widx = {}
        for widx, _, _ in B.gimg.values():
            self._grad_widx[P.weights[widx][1]] = widx
        return P

    def _grad_image(self, idx):
        off, rows, ldw = self._gimg_off[self._grad_widx[idx]]
        return self.gimg.data_ptr() + 4 * off, rows, ldw

    def _finish_unlayout(self):
        keys = self._gimg_keys
        arr = (RelayoutJob * max(1, len(keys)))()
        prefix, first, offs = [0], [0], [0]
        cur = 0
        for j, (widx, (w, C, O, F, Ctot, c0), cols) in enumerate(keys):
            bi = self.bucket_order.index(self._bucket_of[id(w)])
            while cur < bi:
                first.append(j)
                offs.append(prefix[-1])
                cur += 1
            a = arr[j]
            a.W = self.gflat.data_ptr() + 4 * self._goff[id(w)]
            a.base, a.sr, a.sq, a.sf = c0 * F, F, Ctot * F, 1
            a.R, a.Q, a.F, a.mirror, a.ldw = C, O, F, (2 if cols else 0), self._gimg_off[widx][2]
            prefix.append(prefix[-1] + self._gimg_off[widx][1] * self._gimg_off[widx][2])
        nb = len(self.bucket_order)
        while len(first) < nb + 1:
            first.append(len(keys))
            offs.append(prefix[-1])
        dev = self.gflat.device
        self._ul_jobs = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(dev)
        self._ul_prefix = torch.tensor(prefix, dtype=torch.int64, device=dev)
        fa = (ctypes.c_int32 * len(first))(*first)
        oa = (ctypes.c_int64 * len(offs))(*offs)
        check(self._lib.hpl_plan_set_unlayout(self.handle, self._ul_jobs.data_ptr(), self._ul_prefix.data_ptr(), self.gimg.data_ptr(),
                                              fa, oa, nb), 'hpl_plan_set_unlayout')

    # ---- per step
    def tables(self, lat):
        """hpl_level_tables of a prepared training lattice (DeviceLattice with symmetry verdicts), incl. the training-only fields;
        None if a table the mirrored data gradients rely on is not symmetric (the caller then takes the autograd path)."""
        cached = getattr(lat, '_train_tables', None)
        if cached is not None:
            return cached
        # a lattice the native forward cannot take (reference wire format without pair tables, a level without a blur table,
        # too few levels) goes down the documented autograd path instead of raising inside the step
        if not self.accepts(lat) or not hasattr(lat, 'levels') or any(lv.blur[0] is None or lv.clouds[1] is None
                                                                      for lv in lat.levels[:self.NLEV]):
            try:
                lat._train_tables = False
            except AttributeError:          # (the reference's generated_data is a plain list: nothing to cache the verdict on)
                pass
            return False
        arr0, n, keep = level_tables(lat, self.hint)
        n = self.NLEV
        arr = (LevelTables * n)()
        ctypes.memmove(arr, arr0, ctypes.sizeof(LevelTables) * n)
        hold = [arr0, keep]
        for L in range(n):
            lv = lat.levels[L]
            for t in (lv.blur.pair, lv.corr1):
                if t is not None and t.t.shape[0] == 15 and not t.symmetric:
                    lat._train_tables = False
                    return False
            c1 = lv.clouds[1]
            arr[L].bary1, arr[L].off1 = c1.bary.data_ptr(), c1.off.data_ptr()
            taps = lv.blur[0].taps()
            if taps is not None:
                arr[L].up_tap_m, arr[L].up_tap_row, arr[L].up_tap_ptr = [x.data_ptr() for x in taps]
                arr[L].up_tap_max = int(lv.H[0])
                hold.append(taps)
        lat._train_tables = (arr, n, hold)
        return lat._train_tables

    def _split_tables(self):
        """Device tables of hpl_weight_split3_batch: the split images the forward reads / the ones only the backward reads."""
        from ._lib import Split3Job
        out = []
        for sel in (lambda i: i < self._n_fwd_weights, lambda i: i >= self._n_fwd_weights):
            items = [(img, w3) for i, (img, w3) in self._split3.items() if sel(i)]
            if not items:
                out.append(None)
                continue
            arr = (Split3Job * len(items))()
            for a, (img, w3) in zip(arr, items):
                a.Wt, a.dst, a.k_rows, a.ldw, a.plane_stride = img.data_ptr(), w3.planes.data_ptr(), img.shape[0], img.shape[1], w3.planes.stride(0)
                a.planes = w3.P
                if w3.P == 2:
                    a.amax = w3.amax.data_ptr()
            dev = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(self.gflat.device)
            out.append((dev, len(items), max(img.numel() for img, _ in items), int(any(w3.P == 2 for _, w3 in items))))
        return out

    def refresh_weights(self):
        """Start of a step.  On the current stream: the images the FORWARD reads (one re-layout launch, one split launch, the combined
        biases).  On the side stream, beside the forward: the images only the backward reads, and the zeroing of the gradient
        arenas -- the backward program waits for `self._bwd_ready`."""
        if getattr(self, '_split_tabs', None) is None:
            self._split_tabs = self._split_tables()
        L = self._lib

        def split(tab):
            if tab is not None:
                check(L.hpl_weight_split3_batch(tab[0].data_ptr(), tab[1], tab[2], tab[3], stream()), 'hpl_weight_split3_batch')
        main = torch.cuda.current_stream()
        side = self._side if self._side is not None else main
        if side is not main:
            side.wait_stream(main)                # (the optimiser step of the last step, the gradients' last readers)
        self.bank.refresh(0, self._n_fwd_jobs)
        split(self._split_tabs[0])
        for t, (a, b) in self.prog.combined:
            torch.add(a.detach(), b.detach(), out=t)
        with torch.cuda.stream(side):
            self.bank.refresh(self._n_fwd_jobs, None)
            split(self._split_tabs[1])
            self.gflat.zero_()
            self.gimg.zero_()
            self._bwd_ready = torch.cuda.Event()
            self._bwd_ready.record(side)

    @torch.no_grad()
    def step(self, pc1, pc2, sf, lat):
        """Enqueue one training step on the current stream: -> (flow (1, 3, N0), loss tensor [1]) or None when the lattice needs the
        autograd path.  The gradients are complete (and, with several ranks, averaged) once `finish()` has been called."""
        tb = self.tables(lat)
        if tb is False:
            return None
        arr, n, _ = tb
        p1 = pc1[0] if pc1.dim() == 3 else pc1
        p2 = pc2[0] if pc2.dim() == 3 else pc2
        s = sf[0] if sf.dim() == 3 else sf
        if not (p1.is_contiguous() and p2.is_contiguous() and s.is_contiguous()):
            p1, p2, s = p1.contiguous(), p2.contiguous(), s.contiguous()
        if arr[0].n0 != p1.shape[1] or arr[0].n1 != p2.shape[1] or s.shape[1] != p1.shape[1]:
            raise _lib.HplError('lattice was built for %d / %d points, got %d / %d' % (arr[0].n0, arr[0].n1, p1.shape[1], p2.shape[1]))
        need = self._lib.hpl_plan_workspace_bytes(self.handle, arr, n)
        if need < 0:
            raise _lib.HplError('hpl_plan_workspace_bytes: %s' % self._lib.hpl_last_error().decode())
        ws = self._ws.get('train')
        if ws is None or ws.numel() < need:
            # (pairs differ in their vertex counts: head room, so that the ~1.5 GB block is not re-allocated every few steps -- a
            # hipFree + hipMalloc of that size stalls the step for tens of milliseconds)
            self._ws.pop('train', None)
            del ws
            ws = self._ws['train'] = torch.empty(int(need * 1.3), dtype=torch.uint8, device=p1.device)
        self.refresh_weights()
        out = torch.empty((p1.shape[1], 3), dtype=torch.float32, device=p1.device)
        st = stream()
        side = self._side.cuda_stream if self._side is not None else None
        lo = 0
        n_ops = len(self.prog.ops)

        def run(lo, hi, join):
            check(self._lib.hpl_plan_run_range(self.handle, arr, n, ptr(p1), ptr(p2), ptr(s), ptr(out), ptr(self.loss), ws.data_ptr(),
                                               ws.numel(), st, side, lo, hi, join), 'hpl_plan_run_range')
        run(0, self.n_fwd, 0)
        torch.cuda.current_stream().wait_event(self._bwd_ready)          # the backward's weight images, the zeroed gradient arenas
        if not self.reducer._active():
            run(self.n_fwd, n_ops, 1)
        else:
            # several ranks: the program is issued in pieces that end with a bucket's un-layout, and the bucket's all-reduce starts
            # behind it -- on the side stream when there is one (the un-layout ran there): the main stream does not wait for a
            # weight gradient before the end of the step.  Every rank runs the same program: same order of collectives everywhere.
            lo = self.n_fwd
            for k, hi in enumerate(self.cuts):
                run(lo, hi, 0)
                if self._side is not None:
                    with torch.cuda.stream(self._side):
                        self.reducer.launch_flat(self.bucket_order[k])
                else:
                    self.reducer.launch_flat(self.bucket_order[k])
                lo = hi
            run(lo, n_ops, 1)
        self._keep = (p1, p2, s, lat)
        return out.t().unsqueeze(0), self.loss

    def finish(self):
        """Wait for the all-reduces (several ranks) and divide: after this the .grad views hold the step's gradients."""
        self.reducer.finish_flat()

    def reduce_fallback(self):
        """The all-reduce of a step this rank ran through autograd because the native program refused its lattice (step() returned
        None): the gradients sit in the same flat arena, and the buckets go out in THIS plan's order -- `bucket_order`, what a rank
        on the native program issues --, not in the reducer's index order: ranks on different paths in the same step still post the
        same collectives in the same slots (a bucket of another size in a slot hangs or corrupts the job)."""
        for b in self.bucket_order:
            self.reducer.launch_flat(b)
        self.reducer.finish_flat()

    def adam_step(self, opt):
        """optimizer.step() (main.py:216) for a torch.optim.Adam over exactly this plan's parameters (one group, weight_decay 0, no
        amsgrad / maximize: main.py:138-140) as ONE launch (hpl_adam_flat) over the flat parameter / gradient arrays and two flat
        moment arrays.  `opt` stays the owner of the state: its per-parameter 'exp_avg' / 'exp_avg_sq' become views of the flat
        moments (values adopted if it was loaded from a checkpoint), 'step' one shared device scalar -- opt.state_dict() keeps the
        reference's checkpoint format (main.py:183-189).  Do not mix with opt.step() (the shared scalar would be bumped per
        parameter).  Returns False, having done nothing, for an optimiser it does not cover."""
        if isinstance(self._adam, list):
            st = opt.state.get(self.params[0])
            if not st or st.get('step') is not self._adam[2]:      # opt.load_state_dict() since: adopt the loaded state
                self._adam = None
        if self._adam is None:
            g = opt.param_groups
            ok = (isinstance(opt, torch.optim.Adam) and not isinstance(opt, torch.optim.AdamW) and len(g) == 1 and
                  len(g[0]['params']) == len(self.params) and all(a is b for a, b in zip(g[0]['params'], self.params)) and
                  g[0].get('weight_decay', 0) == 0 and not g[0].get('amsgrad') and not g[0].get('maximize') and
                  not isinstance(g[0]['lr'], torch.Tensor))
            if not ok:
                self._adam = False
            else:
                m, v = torch.zeros_like(self.pflat), torch.zeros_like(self.pflat)
                t = 0
                for p in self.params:
                    st = opt.state.get(p)
                    o = self._goff[id(p)]
                    if st:                                    # loaded from a checkpoint (or stepped before): adopt
                        m[o:o + p.numel()].copy_(st['exp_avg'].reshape(-1))
                        v[o:o + p.numel()].copy_(st['exp_avg_sq'].reshape(-1))
                        t = max(t, int(float(st['step'])))
                step_t = torch.full((), float(t), dtype=torch.float32, device=self.pflat.device)
                for p in self.params:
                    o = self._goff[id(p)]
                    opt.state[p] = {'step': step_t, 'exp_avg': m[o:o + p.numel()].view(p.shape),
                                    'exp_avg_sq': v[o:o + p.numel()].view(p.shape)}
                self._adam = [m, v, step_t, t, getattr(torch._C._autograd, '_unsafe_set_version_counter', None)]
        if self._adam is False:
            return False
        m, v, step_t, t, bump = self._adam
        g = opt.param_groups[0]
        t += 1
        self._adam[3] = t
        b1, b2 = g['betas']
        check(_lib.load().hpl_adam_flat(self.pflat.data_ptr(), self.gflat.data_ptr(), m.data_ptr(), v.data_ptr(), self.pflat.numel(),
                                        float(g['lr']), b1, b2, g['eps'], t, stream()), 'hpl_adam_flat')
        step_t.add_(1.0)
        # the launch wrote the parameters behind autograd's back: their version counters are what every weight-image cache keys on
        if bump is not None:
            try:
                bump(tuple(self.params), tuple(p._version + 1 for p in self.params))
            except TypeError:
                # older torch: _unsafe_set_version_counter(Tensor, int), one tensor per call
                try:
                    for p in self.params:
                        bump(p, p._version + 1)
                except TypeError:
                    bump = self._adam[4] = None          # no usable signature: the epoch every cache also keys on
        if bump is None:
            ops.invalidate_weight_cache()
        return True
