cd $GRAFT_REPO_ROOT
O=gpurun_out/r06b; mkdir -p $O
export TMPDIR=/tmp
python -m pytest tests/test_gpu_split3.py tests/test_gpu_kernels.py tests/test_gpu_train_plan.py -x -q -s > $O/pytest_first.txt 2>&1; echo "first rc=$?"; grep -E "row-wise|passed|failed|Error" $O/pytest_first.txt | tail -12
python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest_gpu.txt
CASES="bcn1_ g,bcn2_ g,1x1" REPS=10 python tools/bench_split3.py > $O/split3_after_guard.txt 2>&1; cat $O/split3_after_guard.txt
python bench.py --gpus 1 --steps 20 --warmup 5 --detail $O/bench_driver_cmd_detail.json > $O/bench_driver_cmd.txt 2> $O/bench_err.txt; echo "bench rc=$?"; tail -c 3400 $O/bench_driver_cmd.txt; tail -5 $O/bench_err.txt
