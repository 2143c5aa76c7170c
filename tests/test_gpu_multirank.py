"""GPU: the N > 1 code path of bench.py on the one GPU this box has -- two ranks (two processes), both on cuda:0, `gloo`
for the barrier / max-over-ranks timing and, in training, for the gradient all-reduce (RCCL refuses two ranks on one
device; the 8-GPU run over RCCL is the driver's).  Checks what the contract asks of a multi-rank run: every rank exits
0, rank 0 prints ONE JSON line with the whole-job throughput over both ranks, rank 1 prints no result line."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from test_bench_cpu import check_contract_line  # noqa: E402

#: a multi-rank line carries the contract fields, `roofline` and `ranks`; the single-GPU extras (cpu_baseline, epe3d, exact_bf16x3,
#: the train probe, power) are rank-0-at-N=1 work and must NOT be in it (they would sit inside the driver's clock at N = 2, 4, 8)
MULTI = ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline',
         'dtype', 'data', 'config', 'roofline', 'ranks')


def _one_line(out0, n_gpus, train=False):
    lines = [ln for ln in out0.strip().splitlines() if ln.startswith('{')]
    assert len(lines) == 1
    assert out0.strip().splitlines()[-1] == lines[0]                     # the LAST stdout line
    d = check_contract_line(lines[0], n_gpus=n_gpus, required=MULTI)
    for k in ('cpu_baseline', 'epe3d', 'exact_bf16x3', 'train'):
        assert k not in d, k
    return d


def _run_two_ranks(extra, timeout=420, world=2):
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), WORLD_SIZE=str(world), LOCAL_RANK='0',
               HPL_DIST_BACKEND='gloo', HSA_ENABLE_IPC_MODE_LEGACY='0')
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', str(world), '--no-cpu-baseline'] + extra
    procs = [subprocess.Popen(cmd, cwd=ROOT, env=dict(env, RANK=str(r)), stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                              text=True) for r in reversed(range(world))]
    outs = []
    try:
        for p in procs:
            outs.append(p.communicate(timeout=timeout))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()                                   # the exact processes started above
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, se[-2000:]
    return outs[-1][0], outs[-2][0]                         # stdout of rank 0, of rank 1


@pytest.mark.gpu
def test_two_ranks_inference_on_one_gpu():
    out0, out1 = _run_two_ranks(['--steps', '6', '--warmup', '1', '--points', '2048'])
    assert not [ln for ln in out1.splitlines() if ln.startswith('{')]        # only rank 0 reports (gloo logs a line)
    d = _one_line(out0, 2)
    assert d['n_gpus'] == 2 and d['steps'] == 6 and d['scaling'] == 'weak' and d['value'] > 0
    # whole-job throughput: both ranks' pairs over the slowest rank's time
    assert abs(d['value'] - 2 * 1e3 / d['ms_per_step']) < 1e-5 * d['value']      # (the contract line carries 6 significant digits)
    assert d['pipelined_output_check']['max_abs_diff'] == 0.0


@pytest.mark.gpu
def test_two_ranks_training_step_on_one_gpu():
    """bench.py --train with two ranks: one pair per rank, GradAllReducer hooks + bucketed all-reduce (gloo here), Adam."""
    out0, out1 = _run_two_ranks(['--train', '--steps', '2', '--warmup', '1', '--points', '2048'])
    assert not [ln for ln in out1.splitlines() if ln.startswith('{')]        # only rank 0 reports (gloo logs a line)
    d = _one_line(out0, 2)
    assert d['n_gpus'] == 2 and d['value'] > 0 and d['config']['workload'].lower().find('train') >= 0


@pytest.mark.gpu
def test_eight_ranks_dry_run_on_one_gpu():
    """The rank count the driver launches (WORLD_SIZE = 8), all on the one GPU over gloo: eight host threads, eight lattice
    pipelines, the seed partition and -- in training -- the bucket order of the gradient all-reduce at that size.  Small
    clouds (N = 1024, 3 steps): this checks the code path, not the speed.  Nothing here runs over RCCL with > 1 rank."""
    out0, out1 = _run_two_ranks(['--steps', '3', '--warmup', '1', '--points', '1024', '--no-train-probe'], timeout=600, world=8)
    assert not [ln for ln in out1.splitlines() if ln.startswith('{')]
    d = _one_line(out0, 8)
    assert d['n_gpus'] == 8 and d['steps'] == 3 and d['value'] > 0
    assert abs(d['value'] - 8 * 1e3 / d['ms_per_step']) < 1e-5 * d['value']      # (the contract line carries 6 significant digits)
    # every rank reported, the job's step time is the slowest rank's, and no rank's host threads are what limits it: the
    # host time a step needs (enqueue of one forward + one lattice, minus the time spent waiting for the GPU) stays far
    # below the step, with eight ranks' threads running side by side on this host
    rk = d['ranks']
    assert rk['ranks_seen'] == 8 and len(rk['ms_per_step_by_rank']) == 8 and rk['backend'] == 'gloo'
    assert abs(max(rk['ms_per_step_by_rank']) - d['ms_per_step']) < 1e-5 * d['ms_per_step']
    assert max(rk['host_busy_ms_by_rank']) < d['ms_per_step']
    # absolute: half of ONE GPU's N=8192 step (2.9 ms) with 8 ranks' threads on the host -- for the typical rank: a region of three steps
    # is at the mercy of one scheduling hiccup of the shared box (round 6: one rank of eight at 3.4 ms, the others at 0.7-1.2)
    assert sorted(rk['host_busy_ms_by_rank'])[len(rk['host_busy_ms_by_rank']) // 2] < 1.5
    assert d['pipelined_output_check']['max_abs_diff'] == 0.0
    out0, _ = _run_two_ranks(['--train', '--steps', '3', '--warmup', '1', '--points', '1024'], timeout=600, world=8)
    d = _one_line(out0, 8)
    assert d['n_gpus'] == 8 and d['value'] > 0 and 'train' in d['config']['workload'].lower()


@pytest.mark.gpu
def test_self_spawned_ranks_on_one_gpu():
    """`python bench.py --gpus 2` with NO torchrun environment: bench.py launches its two ranks itself (free rendezvous port,
    rank 0's stdout passed through).  HPL_BENCH_SHARE_GPU=1 puts both on cuda:0 over gloo (RCCL refuses that); without it the
    launcher refuses a box with fewer GPUs than ranks (tests/test_parallel_cpu.py)."""
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_PORT')}
    env['HPL_BENCH_SHARE_GPU'] = '1'
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '4', '--warmup', '1', '--points', '1024',
                        '--no-cpu-baseline', '--no-train-probe'], env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    d = _one_line(p.stdout, 2)
    assert d['n_gpus'] == 2 and d['ranks']['ranks_seen'] == 2 and d['ranks']['backend'] == 'gloo' and d['value'] > 0
    env.pop('HPL_BENCH_SHARE_GPU')
    import torch
    if torch.cuda.device_count() >= 2:
        return                                                   # (a multi-GPU box can place the job: nothing to refuse)
    q = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '1'], env=env, cwd=ROOT,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert q.returncode == 2 and 'visible' in q.stderr           # one GPU on this box: a 2-GPU job is refused, not shrunk
