# round 3, run b: first light of the split-operand kernel (correctness + A/B timing) and the FETCH A/B of the row order
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_split3.py -x -q -s > gpurun_out/r03b_split3_test.txt 2>&1; tail -15 gpurun_out/r03b_split3_test.txt
timeout 600 python tools/bench_split3.py > gpurun_out/r03b_split3_bench.txt 2>&1; cat gpurun_out/r03b_split3_bench.txt | tail -12
HPL_SPLIT3_BN=256 timeout 600 python tools/bench_split3.py > gpurun_out/r03b_split3_bench256.txt 2>&1; cat gpurun_out/r03b_split3_bench256.txt | tail -8
for RO in 1 0; do
  HPL_ROW_ORDER=$RO rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/pmc_f$RO -o f -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-overlap > /dev/null 2>&1
  HPL_ROW_ORDER=$RO rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/pmc_w$RO -o w -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-overlap > /dev/null 2>&1
  cp profiles/pmc_traffic.json gpurun_out/r03b_traffic_ro$RO.json
  python tools/pmc_traffic.py $(ls gpurun_out/pmc_f$RO/*/f_results.db gpurun_out/pmc_f$RO/f_results.db 2>/dev/null | head -1) $(ls gpurun_out/pmc_w$RO/*/w_results.db gpurun_out/pmc_w$RO/w_results.db 2>/dev/null | head -1) gpurun_out/r03b_traffic_ro$RO.json > gpurun_out/r03b_traffic_ro$RO.txt
  rm -rf gpurun_out/pmc_f$RO gpurun_out/pmc_w$RO
  echo RO=$RO; head -1 gpurun_out/r03b_traffic_ro$RO.txt
done
