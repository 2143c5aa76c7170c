#!/usr/bin/env python
"""Benchmark of the bilateral-convolution hot path on MI355X (contract: see the task brief).

A step = one pass of the hot path over one synthetic point-cloud pair that is already resident
in HBM: permutohedral lattice construction on the device (7 levels) + the full HPLFlowNet
forward (21 BCL + 5 CorrBCL calls + Conv1d stacks), N=8192, bs=1 -- BASELINE.json's metric
configuration ("point-pairs/sec + EPE3D, N=8192").  Pairs shard over GPUs as independent samples
(one process per GPU, no data-path collective; inference needs none -- SURVEY.md §8 e), so
`--gpus N` is weak scaling: every rank runs K steps on its own pairs, value = N*K / max-rank time.

Printed JSON (rank 0, one line): the contract fields plus
  roofline      dominant kernel (the wide gather-GEMM launches: fp16 MFMA with scaled fp16-pair operands by default, bf16 triples with HPL_MATH=bf16x3,
                fp32 MFMA with HPL_MATH=f32): executed MFMA flops per launch / HIP-event launch time, as a fraction of
                the matrix pipe's peak.  With several forward streams the launches inside the timed loop share the GPU
                with other pairs' kernels, so the headline figures come from the single-stream pass right after the
                timed loop (`measured` says which); `in_loop` keeps the timed-region figures;
  kernels       per-kernel-class breakdown (gather-GEMM classes by MFMA roofline, splat / slice by
                HBM roofline with the algorithmic bytes of SURVEY.md §8 d2);
  train         BASELINE config 4 on this GPU: 8 training steps on the same pairs (fwd + bwd + all-reduce + Adam);
  cpu_baseline  the CPU oracle ("port": C lattice + the faster of the numpy/BLAS and torch-CPU layer ports) timed on this
                host on a bounded sample of the same workload (rank 0, N=1 only); it doubles as the EPE3D parity check;
  exact_bf16x3  (default run, N=1) the same workload with the exact operand form of rounds 3-4 -- bf16 triples, six partial
                products instead of the three of the scaled fp16 pairs `value` is measured with -- from a subprocess with
                HPL_MATH=bf16x3 after this run's timed regions: both arithmetics in one line.
"""
import argparse
import json
import os
import subprocess
import sys
import threading
import time
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

MFMA_F32_PEAK_TFLOPS = 157.3     # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 64 FLOP/clk/SIMD
MFMA_BF16_PEAK_TFLOPS = 2516.6   # MI355X_MICROARCH.md: ~2.5 PF dense = 1024 SIMDs x 1024 FLOP/clk (32x32x16 in 32 cycles) x 2.4 GHz
SIMDS, PEAK_CLOCK_HZ = 1024, 2.4e9
#: the split-operand path (csrc/gconv3.hip) spends 3 fp16 MFMA flops (scaled fp16 pairs, the default) or 6 bf16 MFMA flops
#: (exact bf16 triples, HPL_MATH=bf16x3) per fp32 flop it stands for; both MFMAs run at the same rate (32 cycles per 32x32x16)
def split_products():
    from hplflownet_amd import ops
    return 3 if ops.SPLIT_PLANES == 2 else 6


def split_words():
    from hplflownet_amd import ops
    if ops.SPLIT_PLANES == 2:
        return ('fp16', 'scaled fp16 pairs (2 x fp16 split operands) on the fp16 MFMA',
                'fp32 operands as scaled fp16 pairs (x s = hi + lo to 2^-22, s the power of two of the matrix\'s largest magnitude); the 3 partial '
                'products hi*hi + hi*lo + lo*hi are accumulated in fp32 on v_mfma_f32_32x32x16_f16: error vs float64 not larger than the '
                'fp32-MFMA kernel\'s (tests/test_gpu_split3.py); HPL_MATH=bf16x3 selects exact bf16 triples (6 products), HPL_MATH=f32 the fp32-MFMA kernels',
                'f32 (wide layers: fp32 operands as scaled fp16 pairs, 3 partial products, fp32 accumulate)')
    return ('bf16', '3 x bf16 split operands on the bf16 MFMA',
            'fp32 operands as exact sums of three bf16 terms (round-to-nearest splits); the 6 partial products with '
            'i + j <= 2 are accumulated in fp32 on v_mfma_f32_32x32x16_bf16: error vs float64 not larger than the '
            'fp32-MFMA kernel\'s (tests/test_gpu_split3.py); HPL_MATH=f32 selects the fp32-MFMA kernels',
            'f32 (wide layers: fp32 operands as exact 3 x bf16 splits, fp32 accumulate)')
HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E 8 TB/s (6.29 TB/s measured copy)


def split3_takes(M, N, C, F, scat=False):
    """Mirror of hpl_gc::launch_split3 (csrc/gconv3.hip): which launches run on the bf16 MFMA with split operands."""
    from hplflownet_amd import ops
    if not (ops.SPLIT3 and not scat and C >= 32 and C % 4 == 0 and N >= 256 and F <= 15 and M >= 1024):
        return False
    tiles = -(-M // 128) * -(-N // 256)
    fills = tiles >= 128
    if F == 1:
        return M >= 8192 or fills
    if M >= 16384 or fills:
        return True
    # mid-size stencils: only split over K into one round of workgroups (partial tiles in the split-K workspace)
    splitk = min(8, 256 // max(1, tiles), (-(-F * C // 32)) // 16)
    return splitk >= 2 and N % 256 == 0 and M * N <= (8 << 20) and splitk * M * N * 4 <= (64 << 20)


def gconv_class(M, N, K=1 << 20):
    """Mirror of the tile selection in csrc/gconv.hip (hpl_gconv_forward)."""
    t128, t64 = (M + 127) // 128, (M + 63) // 64
    if N > 64:
        tn = (N + 127) // 128
        return '64x128' if t64 * tn >= 512 else '64x64'
    if N > 32:
        return '64x64'
    return '128x32' if t128 >= 512 else '64x32'


DOMINANT_F32 = 'gconv_64x128_g'
DOMINANT_SPLIT3 = 'gconv3_128x256_g'


class KernelTimers(object):
    """HIP events around every launch of the hot kernels (torch's current stream is the stream
    the C ABI launches on), aggregated per kernel class."""

    def __init__(self, ops):
        self.ops = ops
        self.records = []
        self.enabled = False
        self.only = None           # set of class names to time (None = all); events cost host time
        self.exec_fractions = False   # compute the executed / algorithmic ratio of every gathered launch (detail pass)
        # replay > 0 (detail pass): every timed launch is issued `replay` more times back to back behind its isolated run, one event
        # pair around the run of replays -> kernel_us.  An event pair around ONE launch also times the dispatch of an isolated
        # kernel (5-10 us on this GPU: more than the kernel itself for the 5-25 us launches); back-to-back launches of a stream
        # leave the kernel's own duration, which is what rocprofv3 --kernel-trace reports (profiles/*_step_timeline.txt).
        self.replay = 0
        self._orig = (ops.gconv_raw, ops.splat_raw, ops.slice_raw)
        timers = self

        def wrap(fn, describe):
            def inner(*a, **k):
                if not timers.enabled:
                    return fn(*a, **k)
                if timers.only is not None and describe(*a, **k)[0] not in timers.only:
                    return fn(*a, **k)
                s = torch.cuda.Event(enable_timing=True)
                e = torch.cuda.Event(enable_timing=True)
                s.record()
                out = fn(*a, **k)
                e.record()
                rep = None
                if timers.replay > 0 and k.get('scat') is None:      # (a scatter launch accumulates: not idempotent)
                    s2 = torch.cuda.Event(enable_timing=True)
                    e2 = torch.cuda.Event(enable_timing=True)
                    s2.record()
                    for _ in range(timers.replay):
                        fn(*a, **k)
                    e2.record()
                    rep = (s2, e2, timers.replay)
                timers.records.append((describe(*a, **k), s, e, rep))
                return out
            return inner

        ef_cache = {}

        def exec_frac(nbr, C, F, perm, rows):
            """executed / algorithmic multiply-adds of a gathered launch: the kernels skip the MFMAs of a block of `rows` output
            rows for the 32-wide contraction slices whose taps none of its rows has (host mirror: needed_slice_fraction)"""
            if nbr is None or F == 1 or C < 32 or not timers.exec_fractions:
                return 1.0            # (exec_fractions is on in the untimed per-launch detail pass only: the mirror syncs the device)
            key = (nbr.data_ptr(), C, F, perm.data_ptr() if perm is not None else 0, rows)
            if key not in ef_cache:
                ef_cache[key] = needed_slice_fraction(types.SimpleNamespace(t=nbr, perm=perm), C, BM=rows)
            return ef_cache[key]

        def d_gconv(A, nbr, M, C, F, Wt, N, **k):
            # suffix: g = gathered (15-tap stencil, template F_LDS=15), d = dense (F_LDS=1); gconv3_* = the split-operand
            # kernel (bf16 MFMA); flops = the fp32 multiply-adds the launch stands for (2*M*F*C*N)
            if k.get('Wt3') is not None and split3_takes(M, N, C, F, k.get('scat') is not None):
                return ('gconv3_128x%d_%s%s' % (256 if -(-N // 256) * 256 * 100 <= -(-N // 128) * 128 * 108 else 128, 'g' if F > 1 else 'd', '' if M >= 16384 else '_mid'),
                        2.0 * M * F * C * N, 0.0, exec_frac(nbr, C, F, k.get('row_perm'), 64))
            return ('gconv_%s_%s' % (gconv_class(M, N, F * C), 'g' if F > 1 else 'd'), 2.0 * M * F * C * N, 0.0,
                    exec_frac(nbr, C, F, k.get('row_perm'), 32))

        # the HBM-bound gathers by lattice size: levels 0-2 of the N=8192 frustum move 10-140 MB per launch, the deeper
        # levels < 1 MB (pure launch latency) -- one class each, so that the big ones are not averaged away
        def d_splat(feat, csr, H, use_norm=True, out=None):
            N, C = feat.shape
            return ('splat' if H >= 8192 else 'splat_deep', 0.0, 4.0 * C * N + 32.0 * N + 4.0 * (C + 1) * (H + 1), 1.0)

        def d_slice(Y, bary, off, N, vscale=None, bias=None, out=None):
            H, C = Y.shape
            return ('slice' if H >= 8192 else 'slice_deep', 0.0, 4.0 * C * H + 32.0 * N + 4.0 * C * N, 1.0)

        ops.gconv_raw = wrap(ops.gconv_raw, d_gconv)
        ops.splat_raw = wrap(ops.splat_raw, d_splat)
        ops.slice_raw = wrap(ops.slice_raw, d_slice)

    def summary(self, steps):
        """Every `frac` of the gather-GEMM classes is the SAME quantity as roofline.frac: matrix-pipe busy time / launch time
        = executed MFMA flops (absent-neighbour blocks skipped: host mirror of the kernels' slice lists; split classes: 6 bf16
        MFMA flops per executed fp32 multiply-add) / launch time / the pipe's peak.  The rate of the multiply-adds the
        REFERENCE performs stays under f32_equivalent_algorithmic_tflops."""
        if not self.records:
            return {}
        agg = {}
        # the launches of a step come in the same order in every step: the fastest run of each POSITION, averaged over a class's
        # positions, is the like-for-like partner of a best-of-N figure (the copy line below)
        per_step = len(self.records) // max(1, steps) if len(self.records) % max(1, steps) == 0 else 0
        best = {}
        rep_ms = {}
        for idx, ((name, flops, nbytes, ef), s, e, rep) in enumerate(self.records):
            if rep is not None:
                r = rep_ms.setdefault(name, [0, 0.0])
                r[0] += 1
                r[1] += rep[0].elapsed_time(rep[1]) / rep[2]
            a = agg.setdefault(name, [0, 0.0, 0.0, 0.0, 0.0])
            a[0] += 1
            ms_ = s.elapsed_time(e)
            a[1] += ms_                       # ms
            a[2] += flops
            a[3] += nbytes
            a[4] += flops * ef
            if per_step:
                k = (name, idx % per_step)
                best[k] = min(best.get(k, ms_), ms_)
        out = {}
        for name, (cnt, ms, flops, nbytes, flops_ex) in sorted(agg.items()):
            d = {'launches_per_step': cnt / float(steps), 'avg_launch_us': 1e3 * ms / cnt,
                 'ms_per_step': ms / steps}
            mine = [v for (nm_, _), v in best.items() if nm_ == name]
            if mine:
                d['best_launch_us'] = 1e3 * sum(mine) / len(mine)
            if flops:
                split = name.startswith('gconv3_')
                ef = flops_ex / flops
                alg = flops / (ms * 1e-3) / 1e12                       # fp32 multiply-adds of the reference per second
                peak = MFMA_BF16_PEAK_TFLOPS if split else MFMA_F32_PEAK_TFLOPS
                executed = alg * ef * (split_products() if split else 1)   # MFMA flops actually issued per second
                d.update(bound='mfma', achieved=executed, peak=peak, unit='TFLOP/s (executed MFMA flops: %s)' % (split_words()[0] if split else 'fp32'),
                         executed_fraction=ef, f32_equivalent_algorithmic_tflops=alg, gflop_per_step=flops / steps / 1e9,
                         path=split_words()[1] if split else 'fp32 MFMA')
            else:
                d.update(bound='hbm', achieved=nbytes / (ms * 1e-3) / 1e9, peak=HBM_PEAK_GBS, unit='GB/s',
                         mbytes_per_step=nbytes / steps / 1e6)
            d['frac'] = d['achieved'] / d['peak']
            if name in rep_ms and rep_ms[name][0] == cnt:
                # back-to-back replays: the kernel's own duration (see KernelTimers.replay)
                k_ms = rep_ms[name][1]
                d['kernel_us'] = 1e3 * k_ms / cnt
                d['frac_isolated_launch'] = d['frac']
                d['achieved_isolated_launch'] = d['achieved']
                d['achieved'] = d['achieved'] * ms / k_ms
                d['frac'] = d['achieved'] / d['peak']
                d['timing'] = ('frac / achieved: from kernel_us = the launch repeated back to back on its stream (HIP events around the run, / repeats) '
                               '= what rocprofv3 --kernel-trace reports per launch; avg_launch_us / *_isolated_launch: HIP events around ONE launch, '
                               'which also times the dispatch latency of an isolated kernel')
            out[name] = d
        return out


def mfma_ceiling(dev):
    """Sustained v_mfma_f32_32x32x2_f32 rate with no memory traffic on this device (TFLOP/s),
    best of three 10-ms bursts (the first burst after an idle period runs at a lower clock)."""
    from hplflownet_amd import _lib
    L = _lib.load_diag()                  # libhplbcl_diag.so (include/hpl_diag.h): not part of the product library
    out = torch.empty(1024 * 256, device=dev)
    best = 0.0
    for _ in range(3):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        L.hpl_mfma_probe(out.data_ptr(), 1024, 1500, _lib.stream())
        e.record()
        torch.cuda.synchronize()
        best = max(best, 1024 * 4.0 * 1500 * 64 * 4096 / (s.elapsed_time(e) * 1e-3) / 1e12)
    return best


def needed_slice_fraction(tbl, C, BM=64, BK=32):
    """Fraction of (tile, 32-wide contraction slice) pairs the gather-GEMM executes for neighbour
    table `tbl` (NbrTable) after tap-mask row sorting: mirrors the slice list built per tile in
    csrc/gconv.hip (a slice is skipped when its taps are absent for all BM rows of the tile)."""
    nbr = tbl.t if tbl.perm is None else tbl.t[:, tbl.perm.long()]
    F, M = nbr.shape
    nt = (M + BM - 1) // BM
    valid = torch.zeros((F, nt * BM), dtype=torch.bool, device=nbr.device)
    valid[:, :M] = nbr >= 0
    tile_tap = valid.view(F, nt, BM).any(dim=2)                       # (F, tiles)
    nk = (F * C + BK - 1) // BK
    kt = torch.arange(nk, device=nbr.device)
    f_lo = (kt * BK) // C
    f_hi = torch.clamp((kt * BK + BK - 1) // C, max=F - 1)
    need = tile_tap[f_lo] | tile_tap[f_hi]                            # (nk, tiles); C >= 32: <= 2 taps per slice
    return float(need.float().mean().item())


def _lattice_worker(job):
    from oracle import lattice_oracle
    pc1, pc2, sfm = job
    lattice_oracle.generate_data(pc1, pc2, sfm)
    return 1


def lattice_worker_rates(sample, sfm, workers=(8, 16)):
    """Lattice builds per second of the C oracle with P worker PROCESSES, one build per worker at a time -- how the
    reference parallelises this stage (DataLoader workers: configs/train_ours.yaml:61 `workers: 16`,
    test_ours_*.yaml `workers: 8`; each Numba build is single-threaded)."""
    import multiprocessing as mp
    out = {}
    pc1, pc2, _ = sample
    ctx = mp.get_context('fork')
    for P in workers:
        try:
            with ctx.Pool(P) as pool:
                pool.map(_lattice_worker, [(pc1, pc2, sfm)] * P)            # warm: library load, page-in
                t0 = time.time()
                n = sum(pool.map(_lattice_worker, [(pc1, pc2, sfm)] * (4 * P)))
                out[str(P)] = n / (time.time() - t0)
        except Exception as e:      # (no fork, no /dev/shm ...): report, do not fail the bench
            out[str(P)] = 'failed: %s' % e
    return out


def cpu_model():
    try:
        for line in open('/proc/cpuinfo'):
            if line.startswith('model name'):
                return line.split(':', 1)[1].strip()
    except Exception:
        pass
    return 'unknown'


def cpu_baseline(samples, sfm, state_dict, shallow=False):
    """The CPU oracle on a bounded sample of the same workload (`samples`: [(pc1, pc2, sf)], about 10-30 s of CPU
    work); returns (dict, flow of the first pair, its EPE3D)."""
    from oracle import bcl_oracle, lattice_oracle
    try:
        from threadpoolctl import threadpool_info
        threads = max([p.get('num_threads', 1) for p in threadpool_info()] + [1])
    except Exception:
        threads = os.cpu_count()
    t_lat = t_fwd = 0.0
    flow0 = None
    for pc1, pc2, _ in samples:
        t0 = time.time()
        gd = lattice_oracle.generate_data(pc1, pc2, sfm)
        t1 = time.time()
        flow = bcl_oracle.hplflownet_forward(state_dict, pc1.T, pc2.T, gd, shallow=shallow)
        t2 = time.time()
        t_lat, t_fwd = t_lat + (t1 - t0), t_fwd + (t2 - t1)
        flow0 = flow if flow0 is None else flow0
    n = len(samples)
    # second port of the layer half: torch-CPU ops in fp32 (oracle/torch_oracle.py; MKL-DNN convs / index ops as the
    # reference's own CPU path uses, models/bilateralNN.py:219), intra-op threads = the same core count (SURVEY.md 8 d3 ii).
    # ONE pair: the faster of the two ports is the baseline.
    t_torch = None
    try:
        from oracle import torch_oracle
        old_threads = torch.get_num_threads()
        torch.set_num_threads(int(threads))
        pc1, pc2, _ = samples[0]
        with torch.no_grad():
            sd_t = torch_oracle.parameters(state_dict, torch.float32, requires_grad=False)
            gd_t = torch_oracle.lattice(gd, torch.float32) if len(samples) == 1 else torch_oracle.lattice(
                lattice_oracle.generate_data(pc1, pc2, sfm), torch.float32)
            t0 = time.time()
            torch_oracle.hplflownet_forward(sd_t, torch.from_numpy(pc1.T.copy()), torch.from_numpy(pc2.T.copy()), gd_t, shallow=shallow)
            t_torch = time.time() - t0
        torch.set_num_threads(old_threads)
    except Exception as e_:          # report, do not fail the bench
        t_torch = 'failed: %s' % e_
    fwd_numpy = t_fwd / n
    fwd_best = min(fwd_numpy, t_torch) if isinstance(t_torch, float) else fwd_numpy
    d = {'value': 1.0 / (t_lat / n + fwd_best), 'unit': 'point-pairs/s', 'cores': int(threads), 'kind': 'port',
         'sample': '%d pairs, N=%d, %d-level lattice build (C oracle, 1 thread) + %s forward timed with both CPU ports '
                   '(numpy/BLAS oracle on %d pairs, torch-CPU fp32 oracle on 1 pair; threads = cores), the faster one counted'
                   % (n, samples[0][0].shape[0], len(sfm), 'HPLFlowNetShallow' if shallow else 'full HPLFlowNet', n),
         'lattice_s': t_lat / n, 'forward_s': fwd_best, 'forward_s_numpy': fwd_numpy, 'forward_s_torch': t_torch,
         'host_cpus': os.cpu_count(), 'cpu_model': cpu_model(),
         # the lattice stage alone with P worker processes, as the reference's DataLoader runs it
         'lattice_pairs_per_s_by_workers': dict({'1': n / t_lat}, **lattice_worker_rates(samples[0], sfm))}
    return d, flow0, bcl_oracle.epe3d(flow0, samples[0][2].T)


def train_probe(H, arch, margs, state, pairs, sfs, gen, steps=8, warmup=3):
    """BASELINE config 4 on this GPU, inside the default command: `steps` training steps (device lattice build + forward +
    EPE3D loss + backward + gradient all-reduce + Adam lr 1e-4, the reference's loop main.py:203-217) on the pairs of the
    inference run; a fresh model with the same weights.  -> dict for the JSON line."""
    from hplflownet_amd import ops, parallel
    dev = pairs[0][0].device
    targs = types.SimpleNamespace(**dict(vars(margs), evaluate=False))
    model = getattr(H, arch)(targs)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in state.items()})
    model = model.to(dev).train()
    native = os.environ.get('HPL_NATIVE_TRAIN', '1') != '0'          # one native program per step (train_plan.TrainPlan); 0: autograd
    opt = torch.optim.Adam(model.parameters(), lr=1e-4, fused=True)
    if native:
        from hplflownet_amd.train_plan import TrainPlan
        reducer = parallel.GradAllReducer(model.parameters(), overlap=False)
        tplan = TrainPlan(model, reducer=reducer)
    else:
        ops.enable_weight_bank(True)
        reducer = parallel.GradAllReducer(model.parameters())
    loss = None
    from hplflownet_amd.lattice import LatticePipeline
    side = torch.cuda.Stream(device=dev, priority=-1)
    main = torch.cuda.current_stream(dev)
    # as `bench.py --train`: the lattice of the next pair is built on a second stream while this pair trains; exactly one
    # build and one optimiser step per timed step
    # the native (fused) builder on a producer thread: the tables of the training path (tap lists, symmetry verdicts) are added
    # there too, off the thread that issues the ~900 launches of the step
    pipe = LatticePipeline(gen, lambda i: pairs[i % len(pairs)], 0, warmup + steps, depth=2, stream=side, for_training=True,
                           native=True, threaded=True)
    keep = []

    def one():
        (i, _), lat, ev = pipe.get()
        main.wait_event(ev)
        p1, p2 = pairs[i % len(pairs)]
        r = tplan.step(p1, p2, sfs[i % len(pairs)], lat) if native else None
        if r is not None:
            tplan.finish()
            if not tplan.adam_step(opt):
                opt.step()
            ls = r[1]
        else:
            if native:
                tplan.gflat.zero_()
            flow = model(p1[None], p2[None], lat)                 # (refreshes the weight bank: one batched re-layout per step)
            ls = torch.norm(flow - sfs[i % len(pairs)][None], p=2, dim=1).mean()
            if not native:
                opt.zero_grad(set_to_none=True)
            ls.backward()
            reducer()
            if not (native and tplan.adam_step(opt)):
                opt.step()
        fin = torch.cuda.Event()
        fin.record(main)
        keep.append((lat, fin))                # side-stream allocations stay alive until the step that used them has RUN
        while len(keep) > 2:
            keep[0][1].synchronize()
            keep.pop(0)
        return ls
    try:
        for _ in range(warmup):
            one()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            loss = one()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) * 1e3 / steps
        out = {'ms_per_step': ms, 'steps': steps, 'warmup': warmup, 'pairs_per_s': 1e3 / ms, 'loss_last_step': float(loss.detach().reshape(-1)[0]),
               'issue': ('one native program per step (train_plan.TrainPlan: %d forward + %d backward ops, weight gradients on a side '
                         'stream)' % (tplan.n_fwd, len(tplan.prog.ops) - tplan.n_fwd)) if native else 'python autograd, launch by launch',
               'step': 'device lattice build (second stream, next pair) + forward + EPE3D loss + backward + gradient all-reduce '
                       '(world size 1 here) + Adam, one pair per GPU (BASELINE config 4)'}
    except Exception as e_:
        out = {'failed': str(e_)}
    ops.enable_weight_bank(False)
    if native:
        del tplan
    del model, opt, reducer
    torch.cuda.empty_cache()
    return out


def smi_sample(index):
    """One rocm-smi reading of GPU `index`: package power, its limit, shader clock (None where rocm-smi has no answer)."""
    try:
        out = subprocess.run(['rocm-smi', '-d', str(index), '--showpower', '--showmaxpower', '--showclocks', '--json'],
                             stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=10).stdout.decode()
        card = next(iter(json.loads(out).values()))
    except Exception:
        return None
    def num(pred):
        for k, v in card.items():
            if pred(k.lower()):
                try:
                    return float(str(v).strip('()').lower().replace('mhz', '').replace('w', ''))
                except ValueError:
                    return None
        return None
    return {'package_w': num(lambda k: 'package power' in k and 'max' not in k),
            'limit_w': num(lambda k: 'max' in k and 'power' in k),
            'sclk_mhz': num(lambda k: k.startswith('sclk clock speed'))}


def source_stamp():
    """What a PMC profile under profiles/ must have been taken with to describe THIS run's kernels: the kernel sources
    and the switches that select kernels / tiles."""
    import hashlib
    h = hashlib.sha256()
    for f in ('gconv.hip', 'gconv3.hip', 'gconv_common.h', 'executor.hip', 'row_order.hip', 'splat_slice.hip'):
        h.update(open(os.path.join(ROOT, 'hplflownet_amd', 'csrc', f), 'rb').read())
    for k in ('HPL_MATH', 'HPL_SPLIT3_EPILOGUE', 'HPL_GCONV_EPILOGUE', 'HPL_RANGE_GUARD'):
        h.update(('%s=%s;' % (k, os.environ.get(k, ''))).encode())
    return h.hexdigest()


def host_report(host, steps, threaded=False):
    """Host wall time per step inside the timed loop.  lattice_build_ms: the main thread inside pipe.get() (the enqueue of
    a build when it drives the builds itself; with the producer thread: waiting for the queue); forward_enqueue_ms: inside
    the forward call (incl. forward_wait_ms, waiting for a free workspace slot = the GPU is the limiter);
    lattice_wait_ms: the MAIN thread blocked because no finished lattice was there; lattice_producer_idle_ms: the producer
    thread waiting for the one count read-back of the pair it builds ahead (idle by design: it runs pairs ahead of the
    consumer); busy_ms: what the main thread would need per step if the GPU were infinitely fast."""
    from hplflownet_amd import lattice as lat_mod, plan as plan_mod
    d = {k: v / steps for k, v in host.items()}
    spin = lat_mod.WAIT['s'] * 1e3 / steps
    d['lattice_wait_ms'] = d['lattice_build_ms'] if threaded else spin
    d['lattice_producer_idle_ms'] = spin if threaded else 0.0
    d['lattice_producer_busy_ms'] = lat_mod.BUSY['s'] * 1e3 / steps if threaded else 0.0      # CPU time of the producer thread per step
    d['forward_wait_ms'] = plan_mod.WAIT['s'] * 1e3 / steps
    d['busy_ms'] = d['lattice_build_ms'] + d['forward_enqueue_ms'] - d['lattice_wait_ms'] - d['forward_wait_ms']
    return d


#: the contract line must stay short: the driver's parser dropped the 20-KB line of round 5 (BENCH_r05.json: parsed null).
#: Everything else goes to the detail file written beside it before the line is printed.
LINE_LIMIT = 6144
DETAIL_FILE = 'bench_detail.json'


def _r(x, sig=6):
    """floats to `sig` significant digits (the detail file keeps the full values)."""
    if isinstance(x, bool) or x is None or isinstance(x, (int, str)):
        return x
    if isinstance(x, float):
        if x != x or x in (float('inf'), float('-inf')):
            return None
        return float('%.*g' % (sig, x))
    if isinstance(x, (list, tuple)):
        return [_r(v, sig) for v in x]
    if isinstance(x, dict):
        return {k: _r(v, sig) for k, v in x.items()}
    return x


def _pick(d, *keys):
    d = d or {}
    return {k: d[k] for k in keys if k in d and d[k] is not None}


def compact_line(full, detail_name=DETAIL_FILE):
    """The ONE stdout line of the contract, made from the full record: contract fields, a short `config`, `roofline` and
    `cpu_baseline` as the brief defines them, and the handful of secondary numbers the reviews quote.  Per-class tables, copy
    lines, notes, power, host timings stay in the detail file (`detail`)."""
    cfg = full.get('config') or {}
    roof = full.get('roofline') or {}
    out = {k: full.get(k) for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better',
                                    'scaling', 'vs_baseline', 'dtype', 'data')}
    out['config'] = _pick(cfg, 'workload', 'num_points', 'step_includes_lattice_build', 'forward_streams', 'lattices_under_construction',
                          'sharding', 'vertices_per_level_pc1', 'exact_fallback_launches', 'launches_per_forward')
    ld = cfg.get('lattice_driver') or {}
    if ld:
        out['config']['lattice_launches_per_pair'] = ld.get('launches_per_pair')
    rl = _pick(roof, 'bound', 'kernel', 'achieved', 'peak', 'unit', 'frac', 'avg_launch_us', 'traffic', 'launches_per_step',
               'mfma_busy_cycles_per_launch', 'executed_fraction', 'executed_source', 'measured', 'frac_at_measured_clock',
               'pmc_note', 'traffic_note')
    for k in ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic'):        # the brief's keys are always there (null: not measured)
        rl.setdefault(k, roof.get(k))
    if isinstance(rl.get('kernel'), str):
        rl['kernel'] = rl['kernel'].split(';')[0][:160]
    for k in ('executed_source', 'measured', 'pmc_note', 'traffic_note'):
        if isinstance(rl.get(k), str):
            rl[k] = rl[k][:120]
    if roof.get('whole_step'):
        rl['whole_step'] = _pick(roof['whole_step'], 'frac')
    if roof.get('in_loop'):
        rl['in_loop'] = _pick(roof['in_loop'], 'frac', 'avg_launch_us')
    out['roofline'] = rl
    if full.get('cpu_baseline'):
        cb = _pick(full['cpu_baseline'], 'value', 'unit', 'cores', 'kind', 'cpu_model', 'lattice_s', 'forward_s')
        if isinstance(full['cpu_baseline'].get('sample'), str):
            cb['sample'] = full['cpu_baseline']['sample'][:200]
        out['cpu_baseline'] = cb
    if full.get('epe3d'):
        out['epe3d'] = _pick(full['epe3d'], 'gpu', 'cpu_oracle', 'abs_delta', 'max_abs_flow_diff')
    if full.get('speedup_vs_cpu_baseline') is not None:
        out['speedup_vs_cpu_baseline'] = full['speedup_vs_cpu_baseline']
    if full.get('exact_bf16x3'):
        out['exact_bf16x3'] = _pick(full['exact_bf16x3'], 'value', 'ms_per_step', 'steps', 'steady', 'epe3d_abs_delta_vs_cpu_oracle',
                                    'roofline_frac', 'error')
    if full.get('train'):
        out['train'] = _pick(full['train'], 'ms_per_step', 'steps', 'warmup', 'pairs_per_s', 'dispatches_per_step')
    rk = full.get('ranks') or {}
    out['ranks'] = _pick(rk, 'world', 'backend', 'ranks_seen', 'ms_per_step_by_rank', 'host_busy_ms_by_rank')
    ks = {}
    for nm, d in (full.get('kernels') or {}).items():
        if nm in ('slice', 'splat', 'slice_deep', 'splat_deep') or nm == full.get('dominant_class'):
            e = _pick(d, 'frac', 'kernel_us', 'launches_per_step', 'trace_frac', 'trace_us')
            c = (d.get('copy_of_same_bytes') or {}).get('kernel_us_vs_copy_back_to_back')
            if c is not None:
                e['vs_copy_back_to_back'] = c
            ks[nm] = e
    out['kernels'] = ks
    if full.get('steady'):
        out['steady'] = _pick(full['steady'], 'steps', 'value', 'ms_per_step')
    if full.get('forward_only'):
        out['forward_only'] = _pick(full['forward_only'], 'steps', 'pairs_per_s', 'ms_per_step')
    if full.get('single_pair_latency_ms'):
        out['single_pair_latency_ms'] = _pick(full['single_pair_latency_ms'], 'lattice_build_ms', 'forward_ms')
    if full.get('pipelined_output_check'):
        out['pipelined_output_check'] = _pick(full['pipelined_output_check'], 'max_abs_diff', 'max_abs')
    if full.get('device_memory_mb'):
        out['device_memory_mb'] = _pick(full['device_memory_mb'], 'max_allocated')
    if full.get('power'):
        out['power'] = _pick(full['power'], 'package_w', 'limit_w', 'sclk_mhz')
    out['detail'] = detail_name
    out = _r(out)
    line = json.dumps(out, separators=(',', ':'))
    if len(line) >= LINE_LIMIT:                      # never print a line the driver cannot read: shed the optional blocks
        for k in ('power', 'device_memory_mb', 'pipelined_output_check', 'single_pair_latency_ms', 'forward_only', 'kernels', 'steady'):
            out.pop(k, None)
            line = json.dumps(out, separators=(',', ':'))
            if len(line) < LINE_LIMIT:
                break
    return line


def emit(full, detail_path):
    """Write the full record beside the script (or where --detail says), then print the short contract line as the LAST stdout line."""
    name = None
    if detail_path:
        try:
            with open(detail_path, 'w') as f:
                json.dump(full, f)
            name = os.path.basename(detail_path)
        except OSError as e:                          # a read-only tree must not cost the line
            print('bench.py: could not write %s: %s' % (detail_path, e), file=sys.stderr)
    line = compact_line(full, name)
    sys.stdout.flush()
    print(line, flush=True)
    return line


def spawn_ranks(n):
    """`python bench.py --gpus N` without a torchrun environment: start N ranks of this same command line, one per visible
    GPU (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* as torchrun would set them, rendezvous on 127.0.0.1 at a free port), pass
    rank 0's stdout through (the ONE JSON line) and return the first non-zero exit code.  Fewer than N visible GPUs is an
    error, not a smaller job -- except under HPL_BENCH_SHARE_GPU=1, the functional test of this path on a 1-GPU box (all
    ranks on device 0, gloo instead of RCCL, which refuses two ranks on one device)."""
    import socket
    share = os.environ.get('HPL_BENCH_SHARE_GPU') == '1'
    visible = torch.cuda.device_count()
    if visible < n and not share:
        print('bench.py --gpus %d: only %d GPU(s) visible; refusing to run a smaller job under that name' % (n, visible), file=sys.stderr)
        return 2
    sock = socket.socket()
    sock.bind(('127.0.0.1', 0))
    port = sock.getsockname()[1]
    sock.close()
    base = dict(os.environ, WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
                HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
    if share:
        base['HPL_DIST_BACKEND'] = 'gloo'
    cmd = [sys.executable, os.path.abspath(__file__)] + sys.argv[1:]
    procs = []
    for r in range(n):
        env = dict(base, RANK=str(r), LOCAL_RANK='0' if share else str(r))
        procs.append(subprocess.Popen(cmd, env=env, stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    try:
        import time as _t
        live = list(procs)
        while live:
            for p in list(live):
                code = p.poll()
                if code is None:
                    continue
                live.remove(p)
                if code != 0 and rc == 0:
                    rc = code
                    for q in live:              # a dead rank leaves the others in a collective: end the job
                        q.terminate()
            _t.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()                        # (exactly the processes started above)
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--points', type=int, default=8192)
    ap.add_argument('--pool', type=int, default=4, help='distinct resident pairs cycled through the steps')
    ap.add_argument('--no-lattice', action='store_true', help='exclude the device lattice build from the step')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-train-probe', action='store_true', help='skip the training steps reported under "train"')
    ap.add_argument('--no-overlap', action='store_true',
                    help='build each lattice on the main stream instead of a second stream overlapping the previous forward')
    ap.add_argument('--streams', type=int, default=3,
                    help='HIP streams the forwards of consecutive pairs alternate over (pairs in flight)')
    ap.add_argument('--lattice-streams', type=int, default=1,
                    help='HIP streams the lattice builds of consecutive pairs alternate over')
    ap.add_argument('--lattice-depth', type=int, default=3,
                    help='pairs whose lattice is under construction at once on the lattice stream (1: block on every '
                         'read-back of vertex counts; measured at N=8192: 2: 272, 3: 292, 4: 292 pairs/s)')
    ap.add_argument('--arch', default='HPLFlowNet', choices=['HPLFlowNet', 'HPLFlowNetShallow'],
                    help='HPLFlowNetShallow + --points 4096 is BASELINE config 2')
    ap.add_argument('--data', default='frustum', choices=['frustum', 'surface'],
                    help='frustum: the uniform FT3D-like frustum of SURVEY.md 8(d1) (the headline workload); surface: points on '
                         'smooth patches, the dense extreme (few lattice vertices per point)')
    ap.add_argument('--no-lattice-thread', dest='lattice_thread', action='store_false',
                    help='drive the native lattice builds from the main thread instead of a producer host thread (the builder '
                         'spends its time in C calls made with the GIL released, so the forward enqueue overlaps it on a second '
                         'core: N=8192 292 -> 299 pairs/s, shallow model N=4096 896 -> 942)')
    ap.add_argument('--python-lattice', action='store_true',
                    help='drive the lattice build stage by stage from Python instead of the native builder')
    ap.add_argument('--python-forward', action='store_true',
                    help='issue the forward launch by launch from Python instead of one native hpl_plan_run per pair')
    ap.add_argument('--train', action='store_true',
                    help='time a training step (fwd + bwd + gradient all-reduce + Adam) instead of inference')
    ap.add_argument('--detail', default=os.path.join(ROOT, DETAIL_FILE),
                    help='where rank 0 writes the full record (per-class kernel tables, notes, power, host timings); the ONE stdout '
                         'line carries the contract fields and stays below %d bytes; empty: no file' % LINE_LIMIT)
    a = ap.parse_args()

    if a.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        # a bare `python bench.py --gpus N` (no torchrun environment): this process becomes the launcher of N ranks
        raise SystemExit(spawn_ranks(a.gpus))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a GPU: the HIP path has no CPU fallback')
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    if world > 1:
        # N ranks share the node's cores: the host side of a step is one Python thread per rank, keep the
        # CPU thread pools (torch intra-op, BLAS) from claiming every core N times over
        torch.set_num_threads(max(1, (os.cpu_count() or 8) // (2 * world)))
    from hplflownet_amd import parallel
    local_world = int(os.environ.get('LOCAL_WORLD_SIZE', world))
    pin = parallel.pin_host_threads(local_rank, local_world) if world > 1 else {'pinned': False}
    parallel.init_distributed(backend='nccl', device=dev)      # RCCL; inference uses it for barrier/max only
    if a.gpus != world:
        raise SystemExit('bench.py: --gpus %d but WORLD_SIZE=%d: the line would report the wrong job size' % (a.gpus, world))

    import hplflownet_amd as H
    from hplflownet_amd import ops
    from hplflownet_amd.synthetic import SCALES_FILTER_MAP, fill_module_, surface_pair, synthetic_pair

    full = a.arch == 'HPLFlowNet'
    sfm = SCALES_FILTER_MAP if full else SCALES_FILTER_MAP[:5]
    margs = types.SimpleNamespace(dim=3, scales_filter_map=sfm, evaluate=True, use_leaky=True,
                                  bcn_use_bias=True, bcn_use_norm=True, last_relu=False, DEVICE='cuda')
    model = getattr(H, a.arch)(margs)
    fill_module_(model, 1.0, 'hash')                      # random-init weights of the named architecture
    state = {k: v.numpy().copy() for k, v in model.state_dict().items()}
    model = model.to(dev).eval()
    gen = H.GenerateDataUnsymmetric(margs, device=dev, wide_up=model.lattice_hint())

    make_pair = surface_pair if a.data == 'surface' else synthetic_pair
    pairs_np = [make_pair(a.points, s) for s in parallel.sample_seeds(rank, world, a.pool)]
    pairs = [(torch.from_numpy(p1.T.copy()).to(dev), torch.from_numpy(p2.T.copy()).to(dev)) for p1, p2, _ in pairs_np]
    fixed_lat = [gen.build(p1, p2) for p1, p2 in pairs] if a.no_lattice else None
    timers = KernelTimers(ops)
    ceiling = mfma_ceiling(dev)
    # inference runs each forward as ONE native call (hplflownet_amd.plan / csrc/executor.hip); --python-forward
    # keeps the per-launch Python path (what round 1 measured).  Training is autograd, i.e. the Python path.
    native = bool(model.native_forward) and not a.python_forward and not a.train
    model.native_forward = native

    def step(i):
        p1, p2 = pairs[i % a.pool]
        lat = fixed_lat[i % a.pool] if a.no_lattice else gen.build(p1, p2)
        return model(p1[None], p2[None], lat)

    def sync_all():
        parallel.barrier()
        torch.cuda.synchronize()

    if a.train:
        # BASELINE config 4: one pair per GPU, identical weights, mean loss over ranks ==
        # all-reduce(mean) of the gradients (77.2 MB fp32 over xGMI), Adam lr 1e-4 (main.py:138-140)
        model.train()
        native_train = os.environ.get('HPL_NATIVE_TRAIN', '1') != '0'
        parallel.broadcast_parameters(model)
        opt = torch.optim.Adam(model.parameters(), lr=1e-4, fused=True)
        sfs = [torch.from_numpy(sf.T.copy()).to(dev) for _, _, sf in pairs_np]
        if native_train:
            # forward + loss + backward as ONE native program per step; gradients in a flat arena the bucketed all-reduce runs on
            from hplflownet_amd.train_plan import TrainPlan
            reducer = parallel.GradAllReducer(model.parameters(), overlap=False)
            tplan = TrainPlan(model, reducer=reducer)
        else:
            ops.enable_weight_bank(True)     # one batched weight re-layout per step
            reducer = parallel.GradAllReducer(model.parameters())

        def compute(i, lat):
            p1, p2 = pairs[i % a.pool]
            r = tplan.step(p1, p2, sfs[i % a.pool], lat) if native_train else None
            if r is not None:
                tplan.finish()
                if not tplan.adam_step(opt):
                    opt.step()
                return r[0]
            if native_train:
                tplan.gflat.zero_()
            flow = model(p1[None], p2[None], lat)
            loss = torch.norm(flow - sfs[i % a.pool][None], p=2, dim=1).mean()      # EPE3DLoss, main.py:213
            if not native_train:
                opt.zero_grad(set_to_none=True)
            loss.backward()
            if native_train:
                tplan.reduce_fallback()      # same bucket order as the ranks that ran the native program
            else:
                reducer()
            if not (native_train and tplan.adam_step(opt)):
                opt.step()
            return flow

        def step(i):                                           # noqa: F811
            p1, p2 = pairs[i % a.pool]
            with torch.no_grad():
                lat = fixed_lat[i % a.pool] if a.no_lattice else gen.build(p1, p2)
            return compute(i, lat)
    else:
        def compute(i, lat):
            p1, p2 = pairs[i % a.pool]
            return model(p1[None], p2[None], lat)

    overlap = not (a.no_lattice or a.no_overlap)
    # Which streams get the high-priority hardware queues.  Rounds 2-4: the lattice stream (a build is a chain of 34 short launches:
    # behind 700-us tiles it starved).  Round 5, fp16 pairs: the forwards of the full model -- their wide launches are short enough
    # for the lattice chain to slip through, and the forward streams no longer share the four normal-priority queues: N = 8 192
    # 442-447 -> 458-462 pairs/s (steady 443 -> 466), N = 2 048 867 -> 878; no priorities at all: 355.  The shallow model is bound by
    # its lattice builds (forward alone 1 740 pairs/s) and keeps the lattice stream in front: 1 176 vs 1 130.
    prio = 'forward' if (full and not a.train) else 'lattice'       # (all streams high: 450 / 443; 4 forward streams: forward alone 528, with the lattice 316)
    side = [torch.cuda.Stream(device=dev, priority=-1 if prio == 'lattice' else 0) for _ in range(max(1, a.lattice_streams))] \
        if overlap else None
    # forwards of consecutive pairs alternate over a.streams HIP streams: the launch-bound deep levels of
    # one pair run in the shadow of the big GEMMs of another (measured: 1: 144, 2: 163, 3: 173, 4: 160 pairs/s)
    n_fwd = max(1, a.streams) if not a.train else 1
    fwd_streams = [torch.cuda.Stream(device=dev, priority=-1 if prio == 'forward' else 0) for _ in range(n_fwd)] \
        if overlap else None

    def run_pipelined(first, count, fixed=None):
        """count steps; the lattices of the next pairs are built on a second HIP stream while the forward
        of pair i runs on the main streams (the reference overlaps the same two stages with DataLoader
        worker processes, main.py:85-92); up to --lattice-depth pairs are under construction at once so
        that the host never blocks on the per-level vertex counts.  Exactly `count` lattice builds and
        `count` forwards, the first build starting inside this call."""
        import collections
        from hplflownet_amd.lattice import LatticePipeline

        nat_lat = (native or a.train) and not a.python_lattice      # (training: native lattice, Python autograd forward)
        pipe = None if fixed is not None else LatticePipeline(
            gen, lambda i: pairs[i % a.pool], first, count, depth=a.lattice_depth, stream=side, for_training=a.train,
            native=nat_lat, threaded=nat_lat and a.lattice_thread)
        done_ev = torch.cuda.Event()
        done_ev.record()
        nxt = [first]

        def build():
            if fixed is not None:                  # (forward-only rate: the same loop on lattices that exist already)
                i = nxt[0]
                nxt[0] += 1
                return i, fixed[i % a.pool], done_ev
            t = time.perf_counter()
            (i, _), lat, ev = pipe.get()
            host['lattice_build_ms'] += (time.perf_counter() - t) * 1e3
            return i, lat, ev
        keep = collections.deque()
        out = None
        for _ in range(count):
            i, lat, ev = build()
            main = fwd_streams[i % n_fwd]
            main.wait_event(ev)
            t = time.perf_counter()
            with torch.cuda.stream(main):
                out = compute(i, lat)
            host['forward_enqueue_ms'] += (time.perf_counter() - t) * 1e3
            fin = torch.cuda.Event()
            fin.record(main)
            keep.append((lat, out, fin))          # side-stream allocations stay alive until their forward is done
            while len(keep) > 1 + n_fwd:
                keep[0][2].synchronize()
                keep.popleft()
        return out

    # which gather-GEMM class dominates this model / size: one untimed step with every launch timed
    dominant = DOMINANT_SPLIT3 if ops.SPLIT3 else DOMINANT_F32
    if not full:
        timers.enabled, timers.only = True, None
        model.native_forward = False         # the per-launch HIP events wrap the Python ops
        with torch.set_grad_enabled(a.train):
            step(0)
        torch.cuda.synchronize()
        model.native_forward = native
        timers.enabled = False
        pre = {k: v for k, v in timers.summary(1).items() if k.startswith('gconv')}
        dominant = max(pre, key=lambda k: pre[k]['ms_per_step'])
        timers.records = []
    host = {'lattice_build_ms': 0.0, 'forward_enqueue_ms': 0.0}     # host wall time inside the timed loop
    with torch.set_grad_enabled(a.train):
        # one-time costs (weight images, allocator pools of every stream, lazy module loads) are paid by
        # PREWARM untimed steps of our own, so that --warmup 0 still measures the steady state
        PREWARM = 3
        if overlap:
            run_pipelined(0, PREWARM + a.warmup)
        else:
            for i in range(PREWARM + a.warmup):
                step(i)
        sync_all()
        timers.enabled = True
        timers.only = {dominant}
        host = dict.fromkeys(host, 0.0)
        from hplflownet_amd import lattice as _lat_mod, plan as _plan_mod
        _lat_mod.WAIT['s'] = _plan_mod.WAIT['s'] = _lat_mod.BUSY['s'] = 0.0
        plan = model.forward_plan() if native else None
        if plan is not None:
            from hplflownet_amd.plan import TAG_WIDE_BLUR
            plan.profile(TAG_WIDE_BLUR)       # HIP events around the wide stencil convs, on the stream they run on
        t0 = time.perf_counter()
        if overlap:
            y = run_pipelined(PREWARM + a.warmup, a.steps)
        else:
            for i in range(a.steps):
                y = step(i)
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        rank_elapsed = elapsed
        timers.enabled = False
        native_prof = None
        if plan is not None:
            plan.profile(-1)
            native_prof = plan.profile_read()
        parallel.barrier()
        elapsed = parallel.max_over_ranks(elapsed, device=dev)
        host_line = host_report(host, a.steps, threaded=(native or a.train) and not a.python_lattice and a.lattice_thread) if overlap else None      # (the regions below run the same loop: snapshot first)
        # every rank's own step time and host busy time (rank 0 prints them: a slow rank, or a host that cannot feed 8 GPUs, shows here)
        rank_stats = parallel.gather_floats([rank_elapsed * 1e3 / a.steps, (host_line or {}).get('busy_ms', 0.0),
                                             float(pin.get('numa_node') if pin.get('numa_node') is not None else -1)], device=dev)
        # The contract times EXACTLY --steps steps; a short region (the driver passes 20: 60 ms) is at the mercy of one
        # scheduling hiccup, so a second region of >= 1 s of the same loop is timed right behind it and reported beside it
        steady = None
        if overlap and not a.train and elapsed < 1.0:
            n2 = max(a.steps, int(1.2 * a.steps / max(elapsed, 1e-3)))
            sync_all()
            t1 = time.perf_counter()
            run_pipelined(PREWARM + a.warmup + a.steps, n2)
            torch.cuda.synchronize()
            e2 = parallel.max_over_ranks(time.perf_counter() - t1, device=dev)
            steady = {'steps': n2, 'value': world * n2 / e2, 'ms_per_step': 1e3 * e2 / n2,
                      'note': 'the same pipelined loop over >= 1 s, timed right behind the contract region of %d steps' % a.steps}
        # forward-only rate: the same streams and loop on lattices that already exist (what the lattice build costs the step)
        fwd_only = None
        if overlap and not a.train:
            fixed = [gen.build_native(p1, p2) if native and not a.python_lattice else gen.build(p1, p2) for p1, p2 in pairs]
            n3 = max(20, min(a.steps, 200))
            run_pipelined(0, 6, fixed=fixed)
            sync_all()
            t1 = time.perf_counter()
            run_pipelined(0, n3, fixed=fixed)
            torch.cuda.synchronize()
            e3 = parallel.max_over_ranks(time.perf_counter() - t1, device=dev)
            fwd_only = {'steps': n3, 'pairs_per_s': world * n3 / e3, 'ms_per_step': 1e3 * e3 / n3}
            del fixed

        # what the board draws while the loop runs (rocm-smi on rank 0's GPU, sampled from a thread during >= 3 s of the same
        # loop on every rank): the wide launches are bound by the clock the chip sustains under matrix load, DESIGN_HISTORY.md 4.8
        power = None
        if overlap and not a.train and not os.environ.get('HPL_BENCH_NO_POWER'):
            per = max(elapsed / a.steps, 1e-4)
            n4 = max(a.steps, int(3.5 / per))
            samples, stop = [], threading.Event()
            def _watch():
                time.sleep(0.6)
                while not stop.is_set() and len(samples) < 4:
                    smp = smi_sample(local_rank)
                    if smp is None:
                        return
                    samples.append(smp)
            th = threading.Thread(target=_watch, daemon=True) if rank == 0 else None
            sync_all()
            t1 = time.perf_counter()
            if th is not None:
                th.start()
            run_pipelined(PREWARM + a.warmup, n4)
            torch.cuda.synchronize()
            e4 = time.perf_counter() - t1
            stop.set()
            if th is not None:
                th.join(timeout=15)
            e4 = parallel.max_over_ranks(e4, device=dev)
            busy = [x for x in samples if x.get('package_w')]
            if busy:
                med = lambda k: sorted(x[k] for x in busy if x.get(k) is not None)[len([x for x in busy if x.get(k) is not None]) // 2] if any(x.get(k) is not None for x in busy) else None
                power = {'package_w': med('package_w'), 'limit_w': med('limit_w'), 'sclk_mhz': med('sclk_mhz'), 'samples': len(busy),
                         'pairs_per_s': world * n4 / e4, 'steps': n4,
                         'note': 'rocm-smi on rank 0, sampled while the same pipelined loop runs on every rank'}

    # the pipelined loop's last output against a plain single-stream forward of the same pair (inference)
    pipe_check = None
    if overlap and not a.train:
        with torch.no_grad():
            ref = step(PREWARM + a.warmup + a.steps - 1)
        torch.cuda.synchronize()
        pipe_check = {'max_abs_diff': float((y - ref).abs().max()), 'max_abs': float(ref.abs().max())}
    dom = timers.summary(a.steps).get(dominant, {})
    if native_prof is not None and native_prof[0] > 0 and full:
        # the native run brackets exactly the launches of the dominant class (bcn1_ / bcn2_ blur convs, two
        # tap-group passes each): the same record the Python-path timers produce
        lat0 = gen.build(*pairs[0])
        gf = sum(2.0 * lat0.levels[L].H[0] * 15 * c * o for L, c, o in ((0, 580, 1024), (1, 324, 512))) / 1e9
        n_l, ms = native_prof
        # launches per step: one per tap group of the two wide convs (the executor stops recording after 32 768
        # launches, so a very long run is covered by its first steps only)
        lps = float(sum(len(lat0.levels[L].blur[0].groups() or [1]) for L in (0, 1)))
        covered = n_l / lps
        dom = {'launches_per_step': lps, 'avg_launch_us': 1e3 * ms / n_l, 'ms_per_step': ms / covered,
               'gflop_per_step': gf, 'steps_covered': covered}
    # per-kernel detail: a separate, untimed, non-overlapped pass (an event pair around each of the
    # ~130 launches costs ~1.5 ms of host time per step, which the timed loop does not pay)
    detail_steps = min(a.steps, 10)
    # one pair at a time on one stream, as a bs = 1 caller sees it: lattice build, then the forward (native path)
    latency = None
    if not a.train and native and not a.no_lattice:
        lb = fw = 0.0
        with torch.no_grad():
            for i in range(6):
                p1, p2 = pairs[i % a.pool]
                torch.cuda.synchronize()
                t0_ = time.perf_counter()
                lat_ = gen.build_native(p1, p2) if not a.python_lattice else gen.build(p1, p2)
                torch.cuda.synchronize()
                t1_ = time.perf_counter()
                model(p1[None], p2[None], lat_)
                torch.cuda.synchronize()
                if i >= 2:
                    lb, fw = lb + (t1_ - t0_), fw + (time.perf_counter() - t1_)
        latency = {'lattice_build_ms': lb / 4 * 1e3, 'forward_ms': fw / 4 * 1e3,
                   'note': 'host wall clock around one pair alone on the GPU, single stream (no overlap with other pairs)'}
    timers.records = []
    timers.only = None
    timers.enabled = True
    timers.exec_fractions = True
    timers.replay = 8
    clk = torch.zeros(4, dtype=torch.int64, device=dev)
    ops.CLOCK_PROBE = clk                    # the wide row-ordered launches stamp their first workgroup's clocks
    model.native_forward = False             # launch by launch, an event pair around each
    with torch.set_grad_enabled(a.train):
        for i in range(detail_steps):
            step(i)
    torch.cuda.synchronize()
    model.native_forward = native
    timers.enabled = False
    timers.exec_fractions = False
    timers.replay = 0
    ops.CLOCK_PROBE = None
    cyc, ticks = clk.tolist()[:2]
    clock_ghz = cyc / float(ticks) * 0.1 if ticks > 0 else None
    kernels = timers.summary(detail_steps)
    # The HBM-bound gathers move 10-60 MB per launch: at 8 TB/s that is 2-8 us, the order of what ANY dependent launch costs
    # on this GPU.  Beside the fraction of the 8 TB/s peak, the line therefore carries what a plain device-to-device copy of
    # the SAME number of bytes per launch reaches, timed the same way (HIP events around one launch, best of 30).
    for nm, d in kernels.items():
        if d.get('bound') != 'hbm' or not d.get('launches_per_step'):
            continue
        nbytes = int(d['mbytes_per_step'] * 1e6 / d['launches_per_step'])
        src = torch.empty(max(16, nbytes // 2), dtype=torch.uint8, device=dev)
        dst = torch.empty_like(src)
        best, tot = None, 0.0
        for _ in range(30):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            dst.copy_(src)
            e1.record()
            e1.synchronize()
            t_ = e0.elapsed_time(e1) * 1e3
            best = t_ if best is None else min(best, t_)
            tot += t_
        # the copy back to back as well (8 per event pair, best of 8 runs): the partner of kernel_us
        b2b = None
        for _ in range(8):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _r in range(8):
                dst.copy_(src)
            e1.record()
            e1.synchronize()
            t_ = e0.elapsed_time(e1) * 1e3 / 8.0
            b2b = t_ if b2b is None else min(b2b, t_)
        # kernel_vs_copy: the kernel's AVERAGE launch against the copy's BEST (rounds 3-4's figure, kept for continuity);
        # like for like: best against best, average against average
        d['copy_of_same_bytes'] = {'bytes_per_launch': nbytes, 'us': best, 'us_avg': tot / 30.0, 'GB/s': nbytes / (best * 1e-6) / 1e9,
                                   'us_back_to_back': b2b, 'kernel_us_vs_copy_back_to_back': (b2b / d['kernel_us']) if d.get('kernel_us') else None,
                                   'kernel_vs_copy': (nbytes / (best * 1e-6)) and d.get('achieved_isolated_launch', d['achieved']) / (nbytes / (best * 1e-6) / 1e9),
                                   'kernel_avg_vs_copy_avg': (tot / 30.0) / d['avg_launch_us'],
                                   'kernel_best_vs_copy_best': (best / d['best_launch_us']) if d.get('best_launch_us') else None}
        del src, dst
    if rank == 0:
        split3 = bool(ops.SPLIT3) and full
        lat0 = gen.build(*pairs[0])
        # ---------------------------------------------------------------- roofline of the dominant kernel
        # The unit is MATRIX-PIPE TIME.  SQ_VALU_MFMA_BUSY_CYCLES counts the cycles a SIMD's matrix pipe is occupied: 64 per
        # v_mfma_f32_32x32x2_f32 (4096 flop), 32 per v_mfma_f32_32x32x16_bf16 (32768 flop) -- at 2.4 GHz on 1024 SIMDs that is
        # the 157.3 TF fp32 / 2.5 PF bf16 datasheet peaks.  achieved = executed MFMA flops per launch / launch duration;
        # frac = achieved / peak = busy cycles / (1024 SIMDs x duration x 2.4 GHz), a fraction <= 1 whatever the operand type.
        # The executed work per launch is MEASURED (rocprofv3 --pmc on this command, tools/pmc_mfma.py -> profiles/mfma_pmc.json,
        # used only when its stamp matches the kernel sources and tile configuration of this run) or, failing that, mirrored
        # on the host from the lattice tables (the kernel skips the MFMAs of a wave's 64 rows -- 32 in the fp32 kernel -- for slices whose taps they lack).
        wide = [(0, 580, 1024), (1, 324, 512)] if full else []
        alg_gf = [2.0 * lat0.levels[L].H[0] * 15 * c * o / 1e9 for L, c, o in wide]          # fp32 multiply-adds, GF
        ex = kernels.get(dominant, {})
        roofline = {'bound': 'mfma', 'unit': 'TFLOP/s', 'traffic': None}
        if full:
            # rows whose MFMAs are skipped together: a wave's 64 rows in the ping-pong split-operand kernel (its compute phase
            # is one block of 24 MFMAs), a 32-row MFMA block in the fp32 kernel
            skip_rows = 64 if split3 else 32

            def executed(tbl, c):
                groups = tbl.groups()
                if not groups:
                    return needed_slice_fraction(tbl, c, BM=skip_rows)
                F = tbl.t.shape[0]
                return sum((f1 - f0) * needed_slice_fraction(types.SimpleNamespace(t=tbl.t[f0:f1], perm=p), c, BM=skip_rows)
                           for f0, f1, p in groups) / F
            fr = [executed(lat0.levels[L].blur[0], c) for L, c, _ in wide]
            mirror_gf = sum(a_ * b_ for a_, b_ in zip(alg_gf, fr))                              # executed fp32-equivalent GF / step
            lps = float(sum(len(lat0.levels[L].blur[0].groups() or [1]) for L, _, _ in wide))
            flop_per_busy = 1024.0 if split3 else 64.0
            peak = MFMA_BF16_PEAK_TFLOPS if split3 else MFMA_F32_PEAK_TFLOPS
            products = split_products() if split3 else 1
            busy_per_launch = mirror_gf * 1e9 * products / flop_per_busy / lps                # matrix-pipe cycles per launch
            src = 'host mirror of the kernel\'s slice lists and %d-row block masks (bench.needed_slice_fraction)' % skip_rows
            stamp = source_stamp()
            pmc_path = os.path.join(ROOT, 'profiles', 'mfma_pmc.json')
            pmc_note = None
            if os.path.exists(pmc_path) and a.points == 8192 and a.data == 'frustum' and not a.train:
                try:
                    pj = json.load(open(pmc_path))
                    if pj.get('stamp') != stamp:
                        pmc_note = 'profiles/mfma_pmc.json ignored: it was taken with other kernel sources / configuration'
                    elif abs(pj['dominant_launches_per_step'] - lps) > 1e-6:
                        pmc_note = 'profiles/mfma_pmc.json ignored: another tap-group schedule'
                    elif abs(pj['dominant_busy_cycles_per_launch'] / busy_per_launch - 1.0) > 0.05:
                        pmc_note = ('profiles/mfma_pmc.json ignored: its executed work (%.4g pipe cycles per launch) and the host '
                                    'mirror (%.4g) disagree by more than 5 %%' % (pj['dominant_busy_cycles_per_launch'], busy_per_launch))
                    else:
                        roofline['executed_host_mirror_busy_cycles'] = busy_per_launch
                        busy_per_launch = pj['dominant_busy_cycles_per_launch']
                        src = 'rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES (profiles/mfma_pmc.json, stamp %s)' % stamp[:12]
                        if pj.get('busy_cycles_per_step'):
                            # the whole step against the same roofline: matrix-pipe cycles of ALL kernels of one pair / step time
                            t_step = elapsed / a.steps / world
                            roofline['whole_step'] = {
                                'mfma_busy_cycles': pj['busy_cycles_per_step'],
                                'frac': pj['busy_cycles_per_step'] / (SIMDS * PEAK_CLOCK_HZ * elapsed / a.steps),
                                'note': 'matrix-pipe cycles of all kernels of one pair (PMC) / (1024 SIMDs x ms_per_step x 2.4 GHz), per GPU'}
                            del t_step
                except Exception as e_:
                    pmc_note = 'profiles/mfma_pmc.json unreadable: %s' % e_
            if pmc_note:
                roofline['pmc_note'] = pmc_note

            def rates(avg_launch_us):
                t_ = avg_launch_us * 1e-6
                ach = busy_per_launch * flop_per_busy / t_ / 1e12
                return {'achieved': ach, 'frac': busy_per_launch / (SIMDS * PEAK_CLOCK_HZ * t_), 'avg_launch_us': avg_launch_us,
                        'f32_equivalent_tflops': ach / products,
                        'f32_equivalent_algorithmic_tflops': sum(alg_gf) / lps * 1e9 / t_ / 1e12}
            roofline.update(peak=peak, launches_per_step=lps, gflop_per_step_algorithmic_f32=sum(alg_gf),
                            executed_fraction=mirror_gf / sum(alg_gf), executed_source=src,
                            mfma_busy_cycles_per_launch=busy_per_launch,
                            kernel=('k_gconv3w<8,4,%d> (gather-GEMM on the %s MFMA, %s, %d partial '
                                    'products accumulated in fp32; 128x256 tiles, 8 waves in two ping-pong rows; blur convs of bcn1_/bcn2_ as two '
                                    'tap-group passes each)' % (ops.SPLIT_PLANES, split_words()[0], 'every fp32 operand a scaled fp16 pair' if ops.SPLIT_PLANES == 2
                                                                else 'every fp32 operand split exactly into 3 bf16 terms', split_products()))
                            if split3 else 'k_gconv<64,128,2,4,true,8,COMPACT> (fp32-MFMA gather-GEMM, 64x128 tiles, 8 waves, 3 workgroups per CU)')
            # Kernel quality is what the kernel does alone on the GPU: with several forward streams the launches inside the timed
            # loop share the CUs with kernels of other pairs.  Headline = the single-stream pass right after the timed loop (same
            # process, HIP events around the same launches; also what rocprofv3 sees, because kernel tracing serialises the
            # streams); in_loop = the same launches bracketed by the native executor inside the timed loop.
            if ex.get('avg_launch_us'):
                roofline.update(rates(ex['avg_launch_us']))
                roofline['measured'] = 'single-stream pass of %d steps right after the timed loop (HIP events around each launch)' % detail_steps
                if dom.get('avg_launch_us'):
                    roofline['in_loop'] = rates(dom['avg_launch_us'])
                    roofline['in_loop']['note'] = 'inside the timed loop, where kernels of %d pairs share the GPU' % n_fwd
            elif dom.get('avg_launch_us'):
                roofline.update(rates(dom['avg_launch_us']))
                roofline['measured'] = 'HIP events around the launches inside the timed loop'
            if split3:
                roofline['f32_mfma_peak'] = MFMA_F32_PEAK_TFLOPS
                roofline['arithmetic'] = split_words()[2]
        else:
            # other models / sizes: the dominant class of the per-class table, algorithmic = executed (no skipping counted)
            roofline.update(peak=ex.get('peak'), achieved=ex.get('achieved'), frac=ex.get('frac'), avg_launch_us=ex.get('avg_launch_us'),
                            launches_per_step=ex.get('launches_per_step'), kernel='class %s of the per-class table' % dominant,
                            measured='single-stream pass after the timed loop', executed_source='algorithmic = executed')
        roofline['measured_f32_mfma_ceiling'] = ceiling
        if clock_ghz:
            # the chip clocks to its power budget: sampled workgroups of the dominant launches accumulate shader cycles and
            # 100 MHz wall ticks (hpl_gconv_desc.clock_probe)
            roofline['shader_clock_ghz'] = clock_ghz
            if roofline.get('frac'):
                roofline['frac_at_measured_clock'] = roofline['frac'] * PEAK_CLOCK_HZ / (clock_ghz * 1e9)
        roofline['note'] = ('achieved = executed MFMA flops per launch / launch duration on the stream it runs on; frac = the matrix pipe\'s busy '
                            'cycles / (1024 SIMDs x duration x 2.4 GHz); f32_equivalent_* = the fp32 multiply-adds that work stands for '
                            '(executed: absent-neighbour products skipped; algorithmic: 2*H*15*C_in*C_out as the reference multiplies them)')
        prof = os.path.join(ROOT, 'profiles', 'pmc_traffic.json')
        if os.path.exists(prof) and full:
            try:
                tj = json.load(open(prof))
                if tj.get('stamp') == source_stamp():
                    roofline['traffic'] = tj.get('dominant_bytes_per_launch')
                    for nm in ('splat', 'slice'):
                        if nm in kernels and tj.get('k_%s_bytes_per_launch_all_levels' % nm):
                            kernels[nm]['traffic_all_levels_avg'] = tj['k_%s_bytes_per_launch_all_levels' % nm]
                else:
                    roofline['traffic_note'] = 'profiles/pmc_traffic.json ignored: taken with other kernel sources / configuration'
            except Exception:
                pass
        # the HBM-bound gathers by their kernel-trace durations inside a forward (profiles/trace_hbm.json, tools/trace_hbm.py: the
        # forward's buffers are cold, bench's own replay of a launch on one buffer set is not) -- under the stamp rule of the PMC files
        tpath = os.path.join(ROOT, 'profiles', 'trace_hbm.json')
        if os.path.exists(tpath) and full and a.points == 8192 and a.data == 'frustum' and not a.train:
            try:
                tj = json.load(open(tpath))
                if tj.get('stamp') == source_stamp():
                    for nm, e in (tj.get('classes') or {}).items():
                        kd = kernels.get(nm)
                        if kd and kd.get('mbytes_per_step') and kd.get('launches_per_step') and e.get('us_per_launch'):
                            per = kd['mbytes_per_step'] * 1e6 / kd['launches_per_step']
                            kd['trace_us'] = e['us_per_launch']
                            kd['trace_frac'] = per / (e['us_per_launch'] * 1e-6) / 1e9 / HBM_PEAK_GBS
            except Exception:
                pass
        line = {'metric': 'point-pairs/sec + EPE3D, N=8192 FlyingThings3D, 1/2/4/8 MI355X',
                'value': world * a.steps / elapsed, 'unit': 'point-pairs/s', 'n_gpus': world, 'steps': a.steps,
                'warmup': a.warmup, 'ms_per_step': 1e3 * elapsed / a.steps, 'higher_is_better': True,
                'scaling': 'weak', 'vs_baseline': None,
                'dtype': split_words()[3] if ops.SPLIT3 else 'f32', 'data': 'synthetic' if a.data == 'frustum' else 'synthetic (surface patches)',
                'config': {'workload': ('full HPLFlowNet %s (7 levels, 19.3M params, random init), ' if full else 'HPLFlowNetShallow %s (5 levels, random init), ') % ('training step (fwd+bwd+grad all-reduce+Adam)' if a.train else 'inference') +
                                       ('FT3D-like synthetic pair' if a.data == 'frustum' else 'synthetic pair of surface patches') + ', N=%d, bs=1 per GPU' % a.points,
                           'num_points': a.points, 'step_includes_lattice_build': not a.no_lattice,
                           'lattice_overlapped_on_second_stream': bool(overlap), 'high_priority_streams': prio if overlap else None,
                           'lattices_under_construction': a.lattice_depth if overlap else 1,
                           'forward_streams': n_fwd if overlap else 1,
                           'forward_issue': 'one native hpl_plan_run per pair' if native else 'python, launch by launch',
                           'lattice_issue': ('native builder (hpl_lattice_*)' + (' on a producer thread' if a.lattice_thread else ''))
                           if (native and not a.python_lattice and overlap) else 'python, stage by stage',
                           'lattice_driver': (lambda nb: {'fused': bool(nb.fused), 'launches_per_pair': getattr(nb, 'launches', None) if nb.fused else '~260 (staged: one count read-back per level)',
                                                          'count_readbacks_per_pair': 1 if nb.fused else len(sfm), 'staged_fallbacks': nb.fallbacks,
                                                          'vertex_bounds_per_cloud': list(nb.bounds[:len(sfm)]) if nb.fused else None})(gen.native_builder())
                           if (native and not a.python_lattice) else None,
                           # launches of ALL forwards of this command that took the second (residual) pass of the fp16-pair form's range
                           # guard (hpl_gconv_desc.a_guard: an operand with a row 2^18 below its largest magnitude); 0 on this workload
                           'exact_fallback_launches': (plan.guard_trips() if (plan is not None and ops.SPLIT_PLANES == 2 and ops.SPLIT3) else None),
                           'sharding': 'independent pairs per GPU, no data-path collective',
                           'vertices_per_level_pc1': [lv.H[0] for lv in gen.build(*pairs[0]).levels]},
                'roofline': roofline, 'kernels': kernels,
                'host_ms_per_step': host_line, 'steady': steady, 'forward_only': fwd_only, 'power': power,
                'ranks': {'world': world, 'backend': (torch.distributed.get_backend() if world > 1 else None),
                          'ranks_seen': len(rank_stats), 'ms_per_step_by_rank': [r[0] for r in rank_stats],
                          'host_busy_ms_by_rank': [r[1] for r in rank_stats], 'numa_node_by_rank': [int(r[2]) for r in rank_stats],
                          'host_threads_pinned': pin,
                          'note': 'multi-rank runs over RCCL on > 1 GPU have not been possible on the builder\'s 1-GPU boxes: unmeasured until the driver\'s SCALE run'},
                'pipelined_output_check': pipe_check, 'single_pair_latency_ms': latency,
                'device_memory_mb': {'max_allocated': torch.cuda.max_memory_allocated(dev) / 2 ** 20,
                                     'reserved': torch.cuda.memory_reserved(dev) / 2 ** 20}}
        if world == 1 and not a.train and not a.no_train_probe and full:
            sfs_ = [torch.from_numpy(sf.T.copy()).to(dev) for _, _, sf in pairs_np]
            line['train'] = train_probe(H, a.arch, margs, state, pairs, sfs_, gen)
        if world == 1 and not a.no_cpu_baseline and not a.train:
            p1, p2, sf = pairs_np[0]
            base, flow_cpu, epe_cpu = cpu_baseline(pairs_np[:2], sfm, state, shallow=not full)
            with torch.no_grad():
                y0 = step(0)
            flow_gpu = y0[0].cpu().numpy()
            epe_gpu = float(np.sqrt(((flow_gpu - sf.T) ** 2).sum(0)).mean())
            line['cpu_baseline'] = base
            line['epe3d'] = {'gpu': epe_gpu, 'cpu_oracle': epe_cpu, 'abs_delta': abs(epe_gpu - epe_cpu),
                             'max_abs_flow_diff': float(np.abs(flow_gpu - flow_cpu).max()),
                             'note': 'random-init weights: parity number, not accuracy'}
            line['speedup_vs_cpu_baseline'] = line['value'] / base['value']
        if (world == 1 and full and not a.train and not a.no_cpu_baseline and ops.SPLIT_PLANES == 2 and ops.SPLIT3
                and not os.environ.get('HPL_BENCH_SUBRUN') and a.points == 8192 and a.data == 'frustum'):
            # The same workload with the EXACT operand form of rounds 3-4 (bf16 triples, six partial products), in a process of its own
            # (the mode is read once per process), on the now idle GPU: the headline's arithmetic is the scaled fp16 pair (operands to
            # 2^-22, sums measured closer to float64 than the fp32 MFMA's) -- a reader who wants fp32 operands carried exactly finds
            # that number here, from the same run.
            try:
                torch.cuda.synchronize()
                env = {k: v for k, v in os.environ.items()
                       if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'LOCAL_WORLD_SIZE', 'GROUP_RANK', 'MASTER_ADDR', 'MASTER_PORT',
                                    'TORCHELASTIC_RUN_ID', 'TORCHELASTIC_RESTART_COUNT', 'TORCHELASTIC_MAX_RESTARTS')}      # (a plain single process)
                env.update(HPL_MATH='bf16x3', HPL_BENCH_SUBRUN='1', HPL_BENCH_NO_POWER='1')
                out = subprocess.run([sys.executable, os.path.abspath(__file__), '--steps', '100', '--warmup', '10', '--no-train-probe',
                                      '--detail', (a.detail[:-5] + '_bf16x3.json') if a.detail.endswith('.json') else ''],
                                     env=env, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, timeout=420).stdout
                sub = json.loads([l for l in out.splitlines() if l.startswith('{')][-1])
                line['exact_bf16x3'] = {'value': sub['value'], 'unit': sub['unit'], 'steps': sub['steps'], 'ms_per_step': sub['ms_per_step'],
                                        'steady': (sub.get('steady') or {}).get('value'), 'dtype': sub['dtype'],
                                        'epe3d_abs_delta_vs_cpu_oracle': (sub.get('epe3d') or {}).get('abs_delta'),
                                        'roofline_frac': sub['roofline'].get('frac'), 'roofline_avg_launch_us': sub['roofline'].get('avg_launch_us'),
                                        'note': 'python bench.py --steps 100 with HPL_MATH=bf16x3 in a subprocess after the timed regions of this run'}
            except Exception as e_:
                line['exact_bf16x3'] = {'error': repr(e_)[:200]}
        if line['ranks']['ranks_seen'] != line['n_gpus'] or line['n_gpus'] != a.gpus:
            raise SystemExit('bench.py: %d of %d ranks reported (--gpus %d): not printing a line for a job of another size'
                             % (line['ranks']['ranks_seen'], line['n_gpus'], a.gpus))
        line['dominant_class'] = dominant
        emit(line, a.detail)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
