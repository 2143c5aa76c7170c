cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_autograd.py tests/test_gpu_plan.py tests/test_gpu_kernels.py -x -q 2>&1 | grep "passed\|failed" | tail -2
export ROUNDS=4 CASES="bcn1_ g,bcn2_ g"
for i in 1 2; do
HPL_LIB=$PWD/hplflownet_amd/libhplbcl_base.so python tools/bench_split3.py 2>&1 | grep "split3 " | sed 's/^/base /' | cut -c1-170
python tools/bench_split3.py 2>&1 | grep "split3 " | sed 's/^/gray /' | cut -c1-170
done
