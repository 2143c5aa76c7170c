cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_math_modes.py -x -q 2>&1 | grep "passed\|failed\|Error" | tail -3
