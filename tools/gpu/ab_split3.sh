cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_split3.py -q -s > gpurun_out/r03c_split3_test.txt 2>&1; grep "max|err|\|passed\|failed\|^E " gpurun_out/r03c_split3_test.txt | cut -c1-250 | tail -20
timeout 600 python tools/bench_split3.py > gpurun_out/r03c_split3_bench.txt 2>&1; tail -7 gpurun_out/r03c_split3_bench.txt
HPL_SPLIT3_BN=128 timeout 600 python tools/bench_split3.py > gpurun_out/r03c_split3_bench128.txt 2>&1; tail -7 gpurun_out/r03c_split3_bench128.txt
