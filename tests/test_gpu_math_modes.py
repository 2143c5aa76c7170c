"""GPU: the suite runs with the default arithmetic (wide layers on the fp16 MFMA with scaled fp16-pair operands,
csrc/gconv3.hip).  HPL_MATH=bf16x3 selects the exact bf16 triples of rounds 3-4 (twice the MFMA work), HPL_MATH=f32 keeps
every launch on the fp32 MFMA -- the A/B switches of the bench numbers and the paths of the kernel instances the default no
longer reaches.  The switch is read once per process, so each mode gets its own interpreter: benchmark-size parity
(configs 3 / 5: reference fixture + oracle), the native executor against the Python path, the backward kernels and the
native training step."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
@pytest.mark.parametrize('mode', ['f32', 'bf16x3'])
def test_other_math_modes_pass_the_benchmark_size_parity_tests(mode):
    env = dict(os.environ, HPL_MATH=mode)
    cmd = [sys.executable, '-m', 'pytest', '-x', '-q', '-m', 'gpu',
           'tests/test_gpu_bench_size.py::test_config3_full_n8192_vs_reference_and_oracle',
           'tests/test_gpu_bench_size.py::test_weight_gradient_and_mirrored_data_gradient_at_bench_size_vs_float64',
           'tests/test_gpu_plan.py::test_native_plan_equals_python_path',
           'tests/test_gpu_split3.py::test_split3_is_deterministic_and_order_independent']
    if mode == 'bf16x3':
        cmd += ['tests/test_gpu_wgrad3.py', 'tests/test_gpu_train_plan.py::test_native_step_matches_autograd']
    p = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-3000:]
    assert ' passed' in p.stdout and 'failed' not in p.stdout.splitlines()[-1]


@pytest.mark.gpu
def test_default_mode_runs_the_split_kernel_on_the_wide_layers():
    """The product path of the default mode really is the split-operand kernel: a wide tap-group launch with a split image
    gives bits that differ from the fp32-MFMA result of the same launch (and both are within fp32 rounding of each other)."""
    import torch
    from hplflownet_amd import ops
    if not ops.SPLIT3:
        pytest.skip('HPL_MATH=f32 in the environment')
    torch.manual_seed(0)
    M, C, F, N = 16384, 64, 1, 256
    A = torch.randn(M, C, device='cuda')
    Wt = torch.randn(C, N, device='cuda')
    y3 = ops.gconv_raw(A, None, M, C, F, Wt, N, Wt3=ops.weight_split3(Wt), split_k=False)
    y1 = ops.gconv_raw(A, None, M, C, F, Wt, N, split_k=False)
    assert not torch.equal(y3, y1)
    assert float((y3 - y1).abs().max()) < 1e-4 * float(y1.abs().max())
