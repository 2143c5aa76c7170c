cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for i in 1 2 3; do (timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider) > gpurun_out/flaky_$i.log 2>&1; grep -E "passed|failed" gpurun_out/flaky_$i.log | tail -1; grep -E "^FAILED" gpurun_out/flaky_$i.log | head; done
