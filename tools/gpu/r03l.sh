cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for M in split3 f32; do HPL_MATH=$M timeout 1500 python -m pytest tests/test_gpu_bench_size.py -x -q -s -k "config4" 2>&1 | grep "gradient parity\|passed\|failed"; done
