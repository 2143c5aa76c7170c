"""Train / evaluate driver for the HIP hot path.

Own counterpart of the reference's loops (/root/reference/main.py:95-286) and checkpoint helpers
(/root/reference/main_utils.py:54-64), reduced to what drives the bilateral layers:

* loss `EPE3DLoss` = mean_n ||flow_n - sf_n||_2 (main.py:213, models/epe3d_loss.py:9-10);
* Adam(lr=1e-4, weight_decay=0) over all parameters (main.py:138-140).  The reference's
  `adjust_learning_rate` computes a decayed rate and then resets every group to `args.lr`
  (main_utils.py:14-30), i.e. the rate is constant; so it is here;
* checkpoint dict {'epoch' (next start epoch), 'arch', 'state_dict', 'min_loss', 'optimizer'}
  written as checkpoint.pth.tar, copied to checkpoint_<epoch>.pth.tar when epoch % 10 == 1 and to
  model_best.pth.tar when best (main.py:183-189, main_utils.py:54-64).  The reference saves the
  state_dict of a DataParallel wrapper, so keys carry a 'module.' prefix; it is written (and
  accepted on load, see flownet.load_reference_checkpoint) so files are interchangeable;
* metrics EPE3D / Acc3D strict / Acc3D relax / outliers (evaluation_utils.py:4-19), on the device.

What is new relative to the reference: the permutohedral lattice of pair i+1 is built on the GPU
on a second HIP stream while pair i trains (the reference builds it in DataLoader worker
processes on the CPU), and with world_size > 1 every rank trains its own pair and the gradients
are averaged with one bucketed all-reduce over RCCL (parallel.GradAllReducer) -- mean loss over
the global batch, as `.mean()` over a batch would give.

On real data (`--dataset FlyingThings3DSubset|KITTI --data-root DIR`) the training set goes through
`data.Augmentation` with the published settings (configs/train_ours.yaml:40-57, main.py:56-63) and is
visited in a fresh random order every epoch (DataLoader shuffle=True, main.py:67); validation goes
through `data.ProcessData` in order (main.py:76-90).  `--init xavier` gives the reference's start
(xavier-normal weights, zero biases: main_utils.py:33-47, main.py:100-101).

    python -m hplflownet_amd.engine --arch HPLFlowNet --points 8192 --pairs 8 --epochs 1 --ckpt-dir /tmp/ck
    python -m hplflownet_amd.engine --evaluate --resume /tmp/ck/model_best.pth.tar --pairs 4
"""
import argparse
import collections
import os
import shutil
from types import SimpleNamespace

import numpy as np
import torch

from . import data as data_mod
from . import ops, parallel
from .flownet import HPLFlowNet, HPLFlowNetShallow, load_reference_checkpoint
from .lattice import GenerateDataUnsymmetric, LatticePipeline
from .synthetic import SCALES_FILTER_MAP, fill_module_, synthetic_pair

ARCHS = {'HPLFlowNet': (HPLFlowNet, 7), 'HPLFlowNetShallow': (HPLFlowNetShallow, 5)}


def epe3d_loss(flow, sf):
    """flow, sf: (B, 3, N) -> scalar mean end-point error."""
    return torch.norm(flow - sf, p=2, dim=1).mean()


def flow_metrics(pred, gt):
    """pred, gt: (N, 3) tensors -> dict(EPE3D, Acc3DS, Acc3DR, Outliers) (evaluation_utils.py:4-19)."""
    err = torch.norm(gt - pred, dim=-1)
    rel = err / (torch.norm(gt, dim=-1) + 1e-4)
    return {'EPE3D': float(err.mean()),
            'Acc3DS': float(((err < 0.05) | (rel < 0.05)).float().mean()),
            'Acc3DR': float(((err < 0.1) | (rel < 0.1)).float().mean()),
            'Outliers': float(((err > 0.3) | (rel > 0.1)).float().mean())}


# configs/train_ours.yaml:40-57
DATA_PROCESS = {'DEPTH_THRESHOLD': 35., 'NO_CORR': True}
AUG_TOGETHER = {'degree_range': 0.1745329252, 'shift_range': 1., 'scale_low': 0.95, 'scale_high': 1.05,
                'jitter_sigma': 0.01, 'jitter_clip': 0.00}
AUG_PC2 = {'degree_range': 0., 'shift_range': 0.3, 'jitter_sigma': 0.01, 'jitter_clip': 0.00}


def init_weights_(model, init_type='xavier', gain=1.0):
    """Every Conv*/Linear weight drawn as the reference does, biases zeroed (main_utils.py:33-47)."""
    fn = {'normal': lambda w: torch.nn.init.normal_(w, 0.0, gain),
          'xavier': lambda w: torch.nn.init.xavier_normal_(w, gain=gain),
          'kaiming': lambda w: torch.nn.init.kaiming_normal_(w, a=0, mode='fan_in'),
          'orthogonal': lambda w: torch.nn.init.orthogonal_(w, gain=gain)}
    if init_type not in fn:
        raise NotImplementedError('initialization method [%s] is not implemented' % init_type)
    with torch.no_grad():        # in-place through the parameter itself: bumps its version, which the cached
        for m in model.modules():    # weight images of the inference path key on (ops.invalidate_weight_cache)
            name = m.__class__.__name__
            if hasattr(m, 'weight') and ('Conv' in name or 'Linear' in name):
                fn[init_type](m.weight)
                if getattr(m, 'bias', None) is not None:
                    m.bias.zero_()
    ops.invalidate_weight_cache()
    return model


def model_args(nscales, device='cuda', evaluate=False):
    """The reference's config keys the layers read (configs/train_ours.yaml)."""
    return SimpleNamespace(dim=3, scales_filter_map=SCALES_FILTER_MAP[:nscales], use_leaky=True, bcn_use_bias=True,
                           bcn_use_norm=True, last_relu=False, DEVICE=device, evaluate=evaluate)


class SyntheticPairs(object):
    """`count` seeded FT3D-like pairs (synthetic.synthetic_pair), resident on the device as (3, N)."""

    def __init__(self, count, num_points, device, first_seed=0):
        self.items = []
        for s in range(first_seed, first_seed + count):
            pc1, pc2, sf = synthetic_pair(num_points, s)
            self.items.append(tuple(torch.from_numpy(np.ascontiguousarray(a.T)).to(device) for a in (pc1, pc2, sf)))

    def __len__(self):
        return len(self.items)

    def __getitem__(self, i):
        return self.items[i]


class Trainer(object):
    def __init__(self, arch='HPLFlowNet', device='cuda', lr=1e-4, seed=0, distributed=False, rank=0, init='hash',
                 native_step=None):
        cls, nsc = ARCHS[arch]
        self.rank = rank
        self.arch, self.device = arch, torch.device(device)
        self.args = model_args(nsc, evaluate=False)
        torch.manual_seed(seed)
        self.model = cls(self.args)
        if init == 'hash':
            fill_module_(self.model, 1.0, 'hash')       # deterministic He-uniform start (no dataset, no RNG state)
        else:
            init_weights_(self.model, init)
        self.shuffle = np.random.RandomState(seed + 7919 * rank)
        self.model.to(self.device)
        self.gen = GenerateDataUnsymmetric(self.args, device=self.device, wide_up=self.model.lattice_hint())
        if self.device.type == 'cuda':
            ops.enable_weight_bank()          # one batched weight re-layout per training step
        # (fused: the same update, two launches instead of sixteen)
        self.opt = torch.optim.Adam([p for p in self.model.parameters() if p.requires_grad], lr=lr, weight_decay=0,
                                    fused=self.device.type == 'cuda')
        self.reducer = None
        if distributed:
            parallel.broadcast_parameters(self.model)
            self.reducer = parallel.GradAllReducer(self.model.parameters())
        self.epoch = 0
        self.min_loss = None
        self._side = torch.cuda.Stream(device=self.device, priority=-1) if self.device.type == 'cuda' else None
        # the training step as one native program (train_plan.TrainPlan: forward + loss + backward enqueued by a handful of C calls,
        # gradients in one flat arena the all-reduce runs on); HPL_NATIVE_TRAIN=0 / native_step=False keep the autograd path
        if native_step is None:
            native_step = os.environ.get('HPL_NATIVE_TRAIN', '1') != '0'
        self.native_step = bool(native_step) and self.device.type == 'cuda'
        self.tplan = None
        self.native_steps = 0

    def train_step(self, pc1, pc2, sf, lat):
        """One optimiser step on one pair (main.py:203-217) -> the loss (device tensor, not synchronised)."""
        if self.native_step and self.tplan is None:
            from .train_plan import TrainPlan
            if self.reducer is None:
                self.reducer = parallel.GradAllReducer(self.model.parameters(), overlap=False)
            self.tplan = TrainPlan(self.model, reducer=self.reducer)
        if self.tplan is not None:
            r = self.tplan.step(pc1, pc2, sf, lat)
            if r is not None:
                self.tplan.finish()
                if not self.tplan.adam_step(self.opt):         # one launch over the flat arrays (hpl_adam_flat)
                    self.opt.step()
                self.native_steps += 1
                return r[1][0].clone()
            self.tplan.gflat.zero_()                 # a lattice the native backward refuses: autograd adds into the same arena
        flow = self.model(pc1[None], pc2[None], lat)
        loss = epe3d_loss(flow, sf[None])
        if self.tplan is None:
            self.opt.zero_grad(set_to_none=True)
        loss.backward()
        if self.tplan is not None:
            self.tplan.reduce_fallback()             # same bucket order as the ranks that ran the native program
        elif self.reducer is not None:
            self.reducer()
        if self.tplan is None or not self.tplan.adam_step(self.opt):
            self.opt.step()
        return loss.detach()

    # ------------------------------------------------------------------ lattice pipeline
    def _lattices(self, data, order, training, depth=2):
        """Yield (sample, lattice) for `order`.  Each sample is fetched once (readers may sample randomly);
        the lattices of the next `depth` samples are under construction on the side stream while the
        current one is consumed, and the host never blocks on their vertex-count read-backs."""
        main = torch.cuda.current_stream(self.device)
        # the native (fused) builder drives both; in training it also adds the tables of the backward (tap lists, symmetry
        # verdicts) -- on a producer thread, off the thread that issues the step's launches
        pipe = LatticePipeline(self.gen, lambda k: data[order[k]], 0, len(order), depth=depth, stream=self._side,
                               for_training=training, native=self.gen.native_supported(),
                               threaded=training and self.gen.native_supported())
        keep = collections.deque()
        try:
            for _ in range(len(order)):
                (_, sample), lat, ev = pipe.get()
                main.wait_event(ev)
                yield sample, lat
                fin = torch.cuda.Event()
                fin.record(main)
                keep.append((lat, sample, fin))         # side-stream memory stays alive until its consumer is done
                while len(keep) > 2:
                    keep.popleft()[2].synchronize()
        finally:
            pipe.close()                                # (a consumer that stops early must not leave the producer thread behind)
            for _, _, fin in keep:
                fin.synchronize()

    # ------------------------------------------------------------------ loops
    def train_epoch(self, data, order=None):
        self.model.train()
        order = list(range(len(data))) if order is None else list(order)
        total = torch.zeros((), device=self.device)
        for (pc1, pc2, sf), lat in self._lattices(data, order, True):
            total += self.train_step(pc1, pc2, sf, lat)
        self.epoch += 1
        tot = parallel.sum_over_ranks([float(total), float(len(order))], device=self.device)
        return tot[0] / max(1.0, tot[1])             # mean loss over the global batch stream (all ranks)

    @torch.no_grad()
    def validate(self, data):
        self.model.eval()
        agg = collections.OrderedDict()
        for (pc1, pc2, sf), lat in self._lattices(data, list(range(len(data))), False):
            flow = self.model(pc1[None], pc2[None], lat)
            for k, v in flow_metrics(flow[0].t(), sf.t()).items():
                agg[k] = agg.get(k, 0.0) + v
        # every rank evaluated its own shard (shards may differ in length by one): sums and the sample count are
        # added over the ranks, so all ranks return the metrics of the WHOLE split (and agree on `best` in fit())
        keys = list(agg) if agg else ['EPE3D', 'Acc3DS', 'Acc3DR', 'Outliers']
        tot = parallel.sum_over_ranks([agg.get(k, 0.0) for k in keys] + [float(len(data))], device=self.device)
        self.val_samples = int(tot[-1])         # over all ranks: 0 = no rank had a validation sample
        return {k: v / max(1.0, tot[-1]) for k, v in zip(keys, tot[:-1])}

    # ------------------------------------------------------------------ checkpoints
    def state(self):
        # (the native training step keeps parameters and Adam moments as views of flat arrays -- train_plan.TrainPlan --: a
        # checkpoint holds tensors of their own, as the reference's does)
        sd = collections.OrderedDict(('module.' + k, v.detach().clone()) for k, v in self.model.state_dict().items())
        osd = self.opt.state_dict()
        osd['state'] = {i: {k: (v.detach().clone() if torch.is_tensor(v) else v) for k, v in st.items()} for i, st in osd['state'].items()}
        return {'epoch': self.epoch, 'arch': self.arch, 'state_dict': sd, 'min_loss': self.min_loss,
                'optimizer': osd}

    def save_checkpoint(self, ckpt_dir, is_best, filename='checkpoint.pth.tar'):
        os.makedirs(ckpt_dir, exist_ok=True)
        path = os.path.join(ckpt_dir, filename)
        st = self.state()
        torch.save(st, path)
        if st['epoch'] % 10 == 1:
            shutil.copyfile(path, os.path.join(ckpt_dir, 'checkpoint_%d.pth.tar' % st['epoch']))
        if is_best:
            shutil.copyfile(path, os.path.join(ckpt_dir, 'model_best.pth.tar'))
        return path

    def resume(self, path, load_optimizer=True):
        ck = torch.load(path, map_location='cpu')
        load_reference_checkpoint(self.model, ck, strict=True)
        if isinstance(ck, dict):
            self.epoch = int(ck.get('epoch', 0))
            self.min_loss = ck.get('min_loss')
            if load_optimizer and 'optimizer' in ck:
                self.opt.load_state_dict(ck['optimizer'])
        return ck

    def fit(self, train_data, val_data, epochs, ckpt_dir=None, log=print, shuffle=False):
        for _ in range(self.epoch, epochs):
            tr = self.train_epoch(train_data, self.shuffle.permutation(len(train_data)) if shuffle else None)
            # validate() ends in a collective: every rank calls it whenever the split exists, also with an empty shard
            # (fewer validation samples than ranks); the decision to fall back to the train loss is taken on the
            # globally reduced sample count so that all ranks agree
            val = tr
            if val_data is not None:
                res = self.validate(val_data)
                if self.val_samples > 0:
                    val = res['EPE3D']
            best = self.min_loss is None or val < self.min_loss
            if best:
                self.min_loss = val
            log('epoch %d  train EPE3D %.5f  val EPE3D %.5f%s' % (self.epoch, tr, val, '  (best)' if best else ''))
            if ckpt_dir and self.rank == 0:
                self.save_checkpoint(ckpt_dir, best)
        return self.min_loss


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument('--arch', default='HPLFlowNet', choices=sorted(ARCHS))
    ap.add_argument('--points', type=int, default=8192)
    ap.add_argument('--pairs', type=int, default=None,
                    help='pairs per epoch and rank: synthetic data default 8; real datasets default 0 = the whole '
                         'split (a positive value caps the shard and is logged)')
    ap.add_argument('--val-pairs', type=int, default=None,
                    help='validation pairs per rank while training: synthetic default 2, real datasets default 0 = all')
    ap.add_argument('--epochs', type=int, default=1)
    ap.add_argument('--lr', type=float, default=1e-4)
    ap.add_argument('--ckpt-dir', default=None)
    ap.add_argument('--resume', default=None)
    ap.add_argument('--evaluate', action='store_true')
    ap.add_argument('--dataset', default='synthetic', choices=['synthetic', 'FlyingThings3DSubset', 'KITTI'])
    ap.add_argument('--data-root', default=None)
    ap.add_argument('--init', default='hash', choices=['hash', 'xavier', 'normal', 'kaiming', 'orthogonal'])
    a = ap.parse_args(argv)
    if a.pairs is None:
        a.pairs = 8 if a.dataset == 'synthetic' else 0
    if a.val_pairs is None:
        a.val_pairs = 2 if a.dataset == 'synthetic' else 0
    rank, world, local_rank = parallel.init_distributed()
    dev = torch.device('cuda', local_rank)
    torch.cuda.set_device(dev)
    tr = Trainer(a.arch, dev, lr=a.lr, distributed=world > 1, rank=rank, init=a.init)
    if a.resume:
        tr.resume(a.resume, load_optimizer=not a.evaluate)
    if a.dataset != 'synthetic':
        return _real_data(a, tr, dev, rank, world)
    if a.evaluate:
        res = tr.validate(SyntheticPairs(a.pairs, a.points, dev, first_seed=1000 + rank * a.pairs))
        if rank == 0:
            print(' '.join('%s %.4f' % kv for kv in res.items()))
        return res
    train = SyntheticPairs(a.pairs, a.points, dev, first_seed=rank * a.pairs)
    val = SyntheticPairs(a.val_pairs, a.points, dev, first_seed=1000)
    return tr.fit(train, val, a.epochs, a.ckpt_dir, log=print if rank == 0 else (lambda *_: None))


class _Shard(object):
    """Every world-th sample of a reader, starting at rank (independent pairs per GPU).

    equal=True (training): every rank gets ceil(len / world) samples, the short ranks wrapping around to the
    start of the reader (what torch's DistributedSampler does) -- train_epoch issues one gradient all-reduce per
    step, so ranks with different step counts would leave each other blocked in a collective.  equal=False
    (validation: no per-step collective, sums are reduced at the end): the plain strided split, no duplicates."""

    def __init__(self, reader, rank, world, limit=0, equal=False):
        self.reader = reader
        n = len(reader)
        if equal and n > 0:
            per = (n + world - 1) // world
            self.ids = [(rank + i * world) % n for i in range(per)]
        else:
            self.ids = list(range(rank, n, world))
        if limit > 0:
            self.ids = self.ids[:limit]

    def __len__(self):
        return len(self.ids)

    def __getitem__(self, i):
        return self.reader[self.ids[i]]


def _real_data(a, tr, dev, rank, world):
    log = print if rank == 0 else (lambda *_: None)
    # the published evaluation protocol (configs/test_ours_KITTI.yaml:9,36-37, test_ours_FlyingThings3D.yaml:9,35-36):
    # NO_CORR True (the two clouds are sampled independently) and allow_less_points True -- a frame with fewer than
    # num_points valid points is evaluated on what it has, not replaced by another frame; training
    # (configs/train_ours.yaml:6) rejects such frames
    if a.dataset == 'KITTI':                        # evaluation only in the reference
        val = data_mod.KITTI(data_mod.ProcessData(DATA_PROCESS, a.points, True, seed=0), a.data_root, device=dev)
        train = None
    else:
        val = data_mod.FlyingThings3DSubset(False, data_mod.ProcessData(DATA_PROCESS, a.points, bool(a.evaluate), seed=0),
                                            a.data_root, device=dev)
        train = None if a.evaluate else data_mod.FlyingThings3DSubset(
            True, data_mod.Augmentation(AUG_TOGETHER, AUG_PC2, DATA_PROCESS, a.points, False, seed=1 + rank),
            a.data_root, device=dev)
    for ds in (train, val):
        msg = ds.check_counts() if ds is not None else None
        if msg:
            log('warning: ' + msg)
    cap = a.val_pairs if train is not None else a.pairs
    if cap > 0:
        log('note: evaluating the first %d samples of each rank\'s shard only (--%s)' % (cap, 'val-pairs' if train is not None else 'pairs'))
    if train is not None and a.pairs > 0:
        log('note: training on the first %d samples of each rank\'s shard only (--pairs)' % a.pairs)
    val = _Shard(val, rank, world, cap)
    if train is None:
        res = tr.validate(val)
        log(' '.join('%s %.4f' % kv for kv in res.items()))
        return res
    return tr.fit(_Shard(train, rank, world, a.pairs, equal=True), val, a.epochs, a.ckpt_dir, log=log, shuffle=True)


if __name__ == '__main__':
    main()
