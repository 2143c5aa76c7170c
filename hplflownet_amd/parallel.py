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
large flat buckets (default 32 MB -> 3 collectives) rather than many small per-tensor calls; a bucket's
all-reduce starts from a gradient hook as soon as its last gradient exists, under the rest of backward.
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
        # HPL_DIST_BACKEND: override (e.g. gloo for a functional run of several ranks on ONE GPU, which RCCL refuses)
        backend = os.environ.get('HPL_DIST_BACKEND') or backend or ('nccl' if torch.cuda.is_available() else 'gloo')
        kw = {}
        if backend == 'nccl' and device is not None:
            kw['device_id'] = device
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, world, local_rank


def _cpulist(text):
    cpus = []
    for part in text.strip().split(','):
        if not part:
            continue
        a, _, b = part.partition('-')
        cpus += list(range(int(a), int(b or a) + 1))
    return cpus


def gpu_numa_node(device_index):
    """NUMA node the GPU hangs off (sysfs of its PCI function), or None if the platform does not say."""
    try:
        pr = torch.cuda.get_device_properties(device_index)
        bdf = '%04x:%02x:%02x.0' % (getattr(pr, 'pci_domain_id', 0), pr.pci_bus_id, pr.pci_device_id)
        node = int(open('/sys/bus/pci/devices/%s/numa_node' % bdf).read())
        return node if node >= 0 else None
    except Exception:
        return None


def pin_host_threads(local_rank, local_world, device_index=None):
    """Keep this rank's host threads (the forward-enqueue thread and the lattice producer thread) on the cores next to its
    GPU: the cores of the GPU's NUMA node, divided among the ranks whose GPUs share that node; without NUMA information an
    even split of the visible cores by local rank.  One rank needs two cores (bench.py reports its busy time per step);
    what matters at 8 ranks is that no rank's threads migrate across sockets or pile up on another rank's cores.
    Returns a dict describing what was done (for the bench line)."""
    info = {'numa_node': None, 'cpus': None, 'pinned': False}
    # Off unless HPL_PIN=1: the slot arithmetic below assumes the GPUs are spread evenly and contiguously over the NUMA nodes, which
    # has never been checked on a real 8-GPU node (the builder's boxes have one GPU) -- a wrong guess would put two ranks on one
    # core slice silently, in the very run it is meant to help.
    if not hasattr(os, 'sched_setaffinity') or os.environ.get('HPL_PIN') != '1':
        return info
    try:
        allowed = sorted(os.sched_getaffinity(0))
        node = gpu_numa_node(local_rank if device_index is None else device_index) if torch.cuda.is_available() else None
        cpus, share, slot = allowed, max(1, local_world), local_rank
        if node is not None:
            on_node = [c for c in _cpulist(open('/sys/devices/system/node/node%d/cpulist' % node).read()) if c in allowed]
            if on_node:
                # ranks on the same node: assume GPUs are spread evenly over the nodes (8 GPUs / 2 sockets on MI355X hosts)
                nodes = len([d for d in os.listdir('/sys/devices/system/node') if d.startswith('node') and d[4:].isdigit()]) or 1
                share = max(1, -(-local_world // nodes))
                slot = local_rank % share
                cpus = on_node
        per = max(2, len(cpus) // share)
        mine = cpus[slot * per:(slot + 1) * per] or cpus
        os.sched_setaffinity(0, mine)
        info.update(numa_node=node, cpus='%d-%d (%d)' % (mine[0], mine[-1], len(mine)), pinned=True)
    except Exception as e:           # containers without sysfs / restricted affinity: report, never fail
        info['error'] = str(e)
    return info


def gather_floats(values, device='cpu'):
    """every rank's list of python floats -> [[rank 0's], [rank 1's], ...] on every rank"""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return [[float(v) for v in values]]
    t = torch.tensor([float(v) for v in values], dtype=torch.float64, device=device)
    out = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [[float(v) for v in o.tolist()] for o in out]


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


def sum_over_ranks(values, device='cpu'):
    """element-wise sum of a list of python floats over all ranks (validation sums and sample counts)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return [float(v) for v in values]
    t = torch.tensor([float(v) for v in values], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return [float(v) for v in t.tolist()]


def barrier():
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()


class GradAllReducer(object):
    """Bucketed flat all-reduce (mean) of parameter gradients, overlapped with the backward pass.

    Parameters are packed in reverse registration order (the order backward produces them) into flat
    buckets of at most `bucket_bytes`.  With `overlap` every parameter carries a post-accumulate hook;
    the bucket whose last gradient has just been accumulated is packed (one multi-tensor copy) and its
    asynchronous all-reduce starts while backward is still computing the earlier layers.  Calling the
    reducer after `backward()` launches whatever is left (buckets holding parameters that received no
    gradient: those count as zeros, so all ranks issue identical collectives in identical order), waits,
    divides by the world size and writes the means back into `.grad`.  Buckets always go out in index
    order -- a bucket that completes early waits for its predecessors -- so ranks cannot interleave them
    differently."""

    def __init__(self, params, bucket_bytes=32 << 20, overlap=True, single_rank=False):
        self.params = [p for p in params if p.requires_grad]
        #: run the collectives in a process group of ONE rank too (a 1-GPU box exercising the RCCL path:
        #: hooks, packing, asynchronous all-reduce, write-back); off by default -- nothing to reduce
        self.single_rank = bool(single_rank)
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
        self._views = [None] * len(self.buckets)
        self._ready = [0] * len(self.buckets)
        self._work = [None] * len(self.buckets)
        self._next = 0                                  # first bucket not launched yet
        self._bucket_of = {}
        self.overlap = bool(overlap) and hasattr(torch.Tensor, 'register_post_accumulate_grad_hook')
        if self.overlap:
            for b, bucket in enumerate(self.buckets):
                for p in bucket:
                    self._bucket_of[id(p)] = b
                    p.register_post_accumulate_grad_hook(self._on_grad)

    def _active(self):
        return dist.is_initialized() and (dist.get_world_size() > 1 or self.single_rank)

    # ---- gradients that already live in ONE flat arena in bucket order (train_plan.TrainPlan: the parameters' .grad are views of
    # it): a bucket's collective runs on its slice, no packing or write-back copies
    def adopt_flat(self, gflat, ranges):
        assert len(ranges) == len(self.buckets)
        self._flat = [gflat[a:b] for a, b in ranges]
        self._views = [[p.grad for p in bucket] for bucket in self.buckets]
        self._flat_launched = []

    def launch_flat(self, b):
        """Start the all-reduce of bucket b (its gradients are complete on the current stream).  Every rank runs the same
        program, so every rank calls this in the same order."""
        if not self._active():
            return
        self._work[b] = dist.all_reduce(self._flat[b], op=dist.ReduceOp.SUM, async_op=True)
        self._flat_launched.append(b)

    def finish_flat(self):
        if not self._active():
            return
        world = dist.get_world_size()
        for b in self._flat_launched:
            self._work[b].wait()
            self._flat[b].div_(world)
            self._work[b] = None
        self._flat_launched = []

    def _on_grad(self, p):
        if not self._active():
            return
        b = self._bucket_of[id(p)]
        self._ready[b] += 1
        while self._next < len(self.buckets) and self._ready[self._next] == len(self.buckets[self._next]):
            self._launch(self._next)

    def _launch(self, b):
        bucket = self.buckets[b]
        flat = self._flat[b]
        if flat is None or flat.device != bucket[0].device:
            n = sum(p.numel() for p in bucket)
            flat = self._flat[b] = torch.empty(n, dtype=bucket[0].dtype, device=bucket[0].device)
            views, o = [], 0
            for p in bucket:
                views.append(flat[o:o + p.numel()].view(p.shape))
                o += p.numel()
            self._views[b] = views
        have = [(v, p.grad) for v, p in zip(self._views[b], bucket) if p.grad is not None]
        if len(have) < len(bucket):
            flat.zero_()
        if have:
            torch._foreach_copy_([v for v, _ in have], [g for _, g in have])
        self._work[b] = dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=True)
        self._next = b + 1

    def __call__(self):
        if not self._active():
            return
        while self._next < len(self.buckets):           # no overlap, or parameters without a gradient
            self._launch(self._next)
        world = dist.get_world_size()
        for b, bucket in enumerate(self.buckets):
            self._work[b].wait()
            self._flat[b].div_(world)
            missing = [(p, v) for p, v in zip(bucket, self._views[b]) if p.grad is None]
            have = [(p.grad, v) for p, v in zip(bucket, self._views[b]) if p.grad is not None]
            if have:
                torch._foreach_copy_([g for g, _ in have], [v for _, v in have])
            for p, v in missing:
                p.grad = v.clone()
            self._work[b] = None
            self._ready[b] = 0
        self._next = 0


def broadcast_parameters(module, src=0):
    """Make every rank start from rank `src`'s parameters and buffers."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return
    for t in list(module.parameters()) + list(module.buffers()):
        dist.broadcast(t.data, src=src)
