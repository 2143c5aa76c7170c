cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_split3.py tests/test_gpu_bench_size.py tests/test_gpu_train_loop.py tests/test_gpu_autograd.py -x -q 2>&1 | grep "passed\|failed\|Error" | tail -4
