"""CPU, world_size 2, gloo: the multi-process plumbing of hplflownet_amd.parallel -- sample
sharding, max-over-ranks timing, bucketed gradient all-reduce == single-process gradient of the
mean loss, parameter broadcast.  (The HIP kernels need a GPU; the collectives do not.)"""
import os
import socket
import sys
import tempfile

import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _make_model(seed):
    torch.manual_seed(seed)
    return torch.nn.Sequential(torch.nn.Conv1d(3, 16, 1), torch.nn.LeakyReLU(0.1), torch.nn.Conv1d(16, 16, 1),
                               torch.nn.LeakyReLU(0.1), torch.nn.Conv1d(16, 3, 1))


def _sample(i):
    g = torch.Generator().manual_seed(100 + i)
    return torch.randn(1, 3, 64, generator=g), torch.randn(1, 3, 64, generator=g)


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1',
                      MASTER_PORT=str(port))
    from hplflownet_amd import parallel as P
    r, w, lr = P.init_distributed(backend='gloo')
    assert (r, w, lr) == (rank, world, rank)
    seeds = P.sample_seeds(rank, world, 3)
    model = _make_model(seed=rank)                 # deliberately different per rank ...
    P.broadcast_parameters(model, src=0)           # ... until broadcast
    red = P.GradAllReducer(model.parameters(), bucket_bytes=1024)   # small buckets -> several collectives
    assert len(red.buckets) > 1 and red.overlap
    x, t = _sample(seeds[0])
    loss = torch.norm(model(x) - t, p=2, dim=1).mean()
    loss.backward()
    launched_in_backward = red._next                # buckets whose all-reduce started from the gradient hooks
    red()
    # the same step without overlap (everything launched after backward) and with a parameter that gets no
    # gradient on this rank only (counts as zeros; collectives stay aligned): identical means
    twin = _make_model(seed=rank)
    twin.load_state_dict(model.state_dict())
    extra = torch.nn.Parameter(torch.ones(5))
    red2 = P.GradAllReducer(list(twin.parameters()) + [extra], bucket_bytes=1024, overlap=False)
    lt = torch.norm(twin(x) - t, p=2, dim=1).mean() + (extra.sum() * 0.5 if rank == 0 else 0.0)
    lt.backward()
    red2()
    same = all(torch.equal(a.grad, b.grad) for a, b in zip(model.parameters(), twin.parameters()))
    # second step through the hooks: counters were reset
    for q in model.parameters():
        q.grad = None
    x2, t2 = _sample(seeds[1])
    torch.norm(model(x2) - t2, p=2, dim=1).mean().backward()
    second_launched = red._next
    red()
    grads2 = [q.grad.clone() for q in model.parameters()]
    for q in model.parameters():
        q.grad = None
    x, t = _sample(seeds[0])
    torch.norm(model(x) - t, p=2, dim=1).mean().backward()
    red()
    P.barrier()
    tmax = P.max_over_ranks(1.0 + rank)
    gathered = P.gather_floats([10.0 + rank, -1.0 * rank])
    os.environ['HPL_PIN'] = '1'                      # (off by default until validated on an 8-GPU node)
    pin = P.pin_host_threads(rank, world)            # (no GPU here: an even split of the visible cores by local rank)
    affinity = sorted(os.sched_getaffinity(0))
    torch.save({'seeds': seeds, 'gathered': gathered, 'pin': pin, 'affinity': affinity, 'grads': [p.grad.clone() for p in model.parameters()], 'same': same,
                'extra': extra.grad.clone(), 'launched': (launched_in_backward, second_launched, len(red.buckets)),
                'grads2': grads2,
                'params': [p.detach().clone() for p in model.parameters()], 'tmax': tmax},
               os.path.join(out_dir, 'r%d.pt' % rank))
    torch.distributed.destroy_process_group()


def test_two_rank_gloo():
    world = 2
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker, args=(world, _free_port(), d), nprocs=world, join=True)
        res = [torch.load(os.path.join(d, 'r%d.pt' % r)) for r in range(world)]
    assert res[0]['seeds'] == [0, 2, 4] and res[1]['seeds'] == [1, 3, 5]            # disjoint shards
    assert res[0]['tmax'] == res[1]['tmax'] == 2.0                                   # slowest rank
    assert res[0]['gathered'] == res[1]['gathered'] == [[10.0, 0.0], [11.0, -1.0]]   # per-rank stats of the bench line
    if all(r['pin']['pinned'] for r in res):                                         # the two ranks' threads on disjoint cores
        assert len(res[0]['affinity']) >= 1 and not (set(res[0]['affinity']) & set(res[1]['affinity']))
    for a, b in zip(res[0]['params'], res[1]['params']):
        assert torch.equal(a, b)                                                     # broadcast
    for a, b in zip(res[0]['grads'], res[1]['grads']):
        assert torch.equal(a, b)                                                     # identical after all-reduce
    for r in res:
        assert r['same']                                                             # overlap == no overlap
        assert torch.equal(r['extra'], torch.full((5,), 0.25))                      # (0.5 + missing -> 0) / 2
        assert r['launched'][0] == r['launched'][1] == r['launched'][2]              # all started inside backward
    for a, b in zip(res[0]['grads2'], res[1]['grads2']):
        assert torch.equal(a, b)
    # reference: one process, mean of the two per-sample losses
    model = _make_model(seed=0)
    loss = sum(torch.norm(model(_sample(s)[0]) - _sample(s)[1], p=2, dim=1).mean() for s in (0, 1)) / 2
    loss.backward()
    for p, g in zip(model.parameters(), res[0]['grads']):
        assert torch.allclose(p.grad, g, atol=1e-6, rtol=1e-5)


def test_cpulist_and_single_rank_helpers():
    sys.path.insert(0, ROOT)
    from hplflownet_amd import parallel as P
    assert P._cpulist('0-3,8,10-11\n') == [0, 1, 2, 3, 8, 10, 11] and P._cpulist('') == []
    assert P.gather_floats([1.5, 2]) == [[1.5, 2.0]]
    before = sorted(os.sched_getaffinity(0))
    try:
        assert not P.pin_host_threads(1, 4)['pinned']                # default: off
        os.environ['HPL_PIN'] = '1'
        info = P.pin_host_threads(1, 4)
        os.environ.pop('HPL_PIN')
        if info['pinned']:
            now = sorted(os.sched_getaffinity(0))
            assert set(now) <= set(before) and len(now) >= min(2, len(before))
    finally:
        os.sched_setaffinity(0, before)


def test_bucket_order_of_the_full_model_is_rank_independent():
    """The failure mode that only shows over RCCL: ranks issuing their bucket all-reduces in different orders.  For the
    FULL model (150 state entries, 19.3 M parameters): the bucket partition is a pure function of the parameter list
    (two independently built models agree name by name, whatever the values), every parameter is in exactly one bucket,
    buckets follow the reverse registration order, and -- whatever order the gradient hooks fire in on a rank -- bucket
    b's collective is never launched before bucket b-1's."""
    import random
    import types
    sys.path.insert(0, ROOT)
    from hplflownet_amd import parallel as P
    from hplflownet_amd.flownet import HPLFlowNet
    from hplflownet_amd.synthetic import SCALES_FILTER_MAP
    args = types.SimpleNamespace(dim=3, scales_filter_map=SCALES_FILTER_MAP, evaluate=False, use_leaky=True, bcn_use_bias=True,
                                 bcn_use_norm=True, last_relu=False, DEVICE='cpu')
    layouts = []
    for seed in (0, 1):
        torch.manual_seed(seed)
        model = HPLFlowNet(args)
        names = {id(p): n for n, p in model.named_parameters()}
        red = P.GradAllReducer(model.parameters())
        layouts.append([[(names[id(p)], tuple(p.shape)) for p in b] for b in red.buckets])
    assert layouts[0] == layouts[1]
    flat = [n for b in layouts[0] for n, _ in b]
    assert len(flat) == len(set(flat)) == len(list(model.named_parameters()))
    assert flat == [n for n, _ in reversed(list(model.named_parameters()))]
    assert len(layouts[0]) >= 3 and all(sum(torch.Size(s).numel() for _, s in b) * 4 <= (32 << 20) or len(b) == 1 for b in layouts[0])
    # hooks in a scrambled order (another rank's autograd may finish its leaves differently): launches stay in index order
    launched = []
    red._active = lambda: True
    red._launch = lambda b: (launched.append(b), setattr(red, '_next', b + 1))
    order = list(red.params)
    random.Random(5).shuffle(order)
    for p in order:
        red._on_grad(p)
    assert launched == list(range(len(red.buckets)))


def test_bench_refuses_a_job_it_cannot_place():
    """`python bench.py --gpus N` without a torchrun environment spawns its own ranks -- and with fewer than N visible GPUs
    (none here) it must fail loudly instead of reporting a one-rank job as N GPUs."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'HPL_BENCH_SHARE_GPU')}
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '1'], env=env, cwd=ROOT,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert p.returncode == 2 and 'visible' in p.stderr and not [ln for ln in p.stdout.splitlines() if ln.startswith('{')]


def _flat_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    from hplflownet_amd import parallel as P
    P.init_distributed(backend='gloo')
    model = _make_model(seed=0)
    red = P.GradAllReducer(model.parameters(), bucket_bytes=1024, overlap=False)
    # the layout train_plan.TrainPlan uses: one flat arena in bucket order, .grad = views of it
    flat = torch.zeros(sum(p.numel() for p in model.parameters()))
    ranges, o = [], 0
    for b in red.buckets:
        o0 = o
        for p in b:
            p.grad = flat[o:o + p.numel()].view(p.shape)
            o += p.numel()
        ranges.append((o0, o))
    red.adopt_flat(flat, ranges)
    x, t = _sample(rank)
    loss = torch.norm(model(x) - t, p=2, dim=1).mean()
    grads = torch.autograd.grad(loss, list(model.parameters()))
    for p, g in zip(model.parameters(), grads):
        p.grad.copy_(g)                                # (what the native backward does: write into the arena)
    for b in reversed(range(len(red.buckets))):        # any order, as long as every rank uses the same one
        red.launch_flat(b)
    red.finish_flat()
    torch.save({'grads': [p.grad.clone() for p in model.parameters()], 'views': all(p.grad.data_ptr() >= flat.data_ptr() for p in model.parameters())},
               os.path.join(out_dir, 'f%d.pt' % rank))
    torch.distributed.destroy_process_group()


def test_two_rank_flat_arena_all_reduce():
    """GradAllReducer.adopt_flat / launch_flat / finish_flat (the native training step's gradients live in one arena): the mean of
    the two ranks' gradients, in place, no packing copies."""
    world = 2
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_flat_worker, args=(world, _free_port(), d), nprocs=world, join=True)
        res = [torch.load(os.path.join(d, 'f%d.pt' % r)) for r in range(world)]
    assert res[0]['views'] and res[1]['views']
    model = _make_model(seed=0)
    loss = sum(torch.norm(model(_sample(s)[0]) - _sample(s)[1], p=2, dim=1).mean() for s in (0, 1)) / 2
    loss.backward()
    for p, a, b in zip(model.parameters(), res[0]['grads'], res[1]['grads']):
        assert torch.equal(a, b) and torch.allclose(p.grad, a, atol=1e-6, rtol=1e-5)
