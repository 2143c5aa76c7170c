"""Multi-GPU plumbing: one process per GPU, `torch.distributed` (backend "nccl" = RCCL over xGMI
on ROCm; "gloo" in the CPU tests).

The hot path shards by sample: every pair builds its own lattices and the layers are B = 1
(reference README.md:57), so inference needs **no data-path collective** -- ranks only meet at
barriers and for the max-over-ranks timing.  Training adds one collective: the all-reduce of the
19.3 M-parameter gradient (77.2 MB fp32).  The reference has no counterpart (it wraps the model
in `torch.nn.DataParallel`, main.py:104, degenerate at batch size 1); this is new functionality
required by BASELINE config 4.

xGMI is point-to-point (7 links x ~153 GB/s per GPU): a ring all-reduce pushes 2*(7/8)*77.2 MB
through one link per GPU (~0.9 ms), far below a training step, so gradients are reduced in a few
large flat buckets (default 32 MB -> 3 collectives) launched as soon as backward has finished,
rather than many small per-tensor calls.
"""
import os

import torch
import torch.distributed as dist


def init_distributed(backend=None, device=None):
    """Initialise the default process group from the torchrun environment (RANK, WORLD_SIZE,
    LOCAL_RANK, MASTER_ADDR/PORT).  Returns (rank, world, local_rank); no-op for WORLD_SIZE=1."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        backend = backend or ('nccl' if torch.cuda.is_available() else 'gloo')
        kw = {}
        if backend == 'nccl' and device is not None:
            kw['device_id'] = device
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, world, local_rank


def sample_seeds(rank, world, per_rank, base=0):
    """Disjoint sample ids for this rank: rank r owns base + r, base + r + world, ... (the usual
    strided DistributedSampler split; pairs are independent, so any partition is valid)."""
    return [base + rank + i * world for i in range(per_rank)]


def max_over_ranks(value, device='cpu'):
    """max of a python float over all ranks (the job's step time is the slowest rank's)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def barrier():
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()


class GradAllReducer(object):
    """Bucketed flat all-reduce (mean) of parameter gradients.

    Parameters are packed in reverse registration order (the order backward produces them) into
    flat buckets of at most `bucket_bytes`; each bucket is one asynchronous all-reduce.  Missing
    gradients count as zeros so that all ranks issue identical collectives."""

    def __init__(self, params, bucket_bytes=32 << 20):
        self.params = [p for p in params if p.requires_grad]
        self.buckets = []
        cur, size = [], 0
        for p in reversed(self.params):
            nbytes = p.numel() * p.element_size()
            if cur and size + nbytes > bucket_bytes:
                self.buckets.append(cur)
                cur, size = [], 0
            cur.append(p)
            size += nbytes
        if cur:
            self.buckets.append(cur)
        self._flat = [None] * len(self.buckets)

    def __call__(self):
        if not dist.is_initialized() or dist.get_world_size() == 1:
            return
        world = dist.get_world_size()
        work = []
        for i, bucket in enumerate(self.buckets):
            n = sum(p.numel() for p in bucket)
            flat = self._flat[i]
            if flat is None or flat.device != bucket[0].device:
                flat = self._flat[i] = torch.empty(n, dtype=bucket[0].dtype, device=bucket[0].device)
            o = 0
            for p in bucket:
                k = p.numel()
                if p.grad is None:
                    flat[o:o + k].zero_()
                else:
                    flat[o:o + k].copy_(p.grad.reshape(-1))
                o += k
            work.append((dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=True), flat, bucket))
        for w, flat, bucket in work:
            w.wait()
            flat.div_(world)
            o = 0
            for p in bucket:
                k = p.numel()
                if p.grad is None:
                    p.grad = flat[o:o + k].reshape(p.shape).clone()
                else:
                    p.grad.copy_(flat[o:o + k].reshape(p.shape))
                o += k


def broadcast_parameters(module, src=0):
    """Make every rank start from rank `src`'s parameters and buffers."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return
    for t in list(module.parameters()) + list(module.buffers()):
        dist.broadcast(t.data, src=src)
