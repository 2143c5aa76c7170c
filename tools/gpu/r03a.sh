# round 3, run a: GPU tests on the new deterministic (mask, Morton) row order + A/B of the order: pairs/s, launch time, FETCH_SIZE
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r03a_pytest.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/r03a_pytest.txt
tail -3 gpurun_out/r03a_pytest.txt
for RO in 1 0; do
  HPL_ROW_ORDER=$RO python bench.py --no-cpu-baseline > gpurun_out/r03a_bench_ro$RO.json 2> gpurun_out/r03a_bench_ro$RO.err
  HPL_ROW_ORDER=$RO rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/pmc_f$RO -o f -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-overlap > /dev/null 2>&1
  HPL_ROW_ORDER=$RO rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/pmc_w$RO -o w -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-overlap > /dev/null 2>&1
  cp profiles/pmc_traffic.json gpurun_out/r03a_traffic_ro$RO.json
  python tools/pmc_traffic.py $(ls gpurun_out/pmc_f$RO/*/f_results.db gpurun_out/pmc_f$RO/f_results.db 2>/dev/null | head -1) $(ls gpurun_out/pmc_w$RO/*/w_results.db gpurun_out/pmc_w$RO/w_results.db 2>/dev/null | head -1) gpurun_out/r03a_traffic_ro$RO.json > gpurun_out/r03a_traffic_ro$RO.txt
  rm -rf gpurun_out/pmc_f$RO gpurun_out/pmc_w$RO
  python - <<PY
import json
d=json.load(open('gpurun_out/r03a_bench_ro$RO.json')); r=d['roofline']
print('RO=$RO', round(d['value'],1), d['ms_per_step'], {k:r.get(k) for k in ('frac','avg_launch_us','shader_clock_ghz')}, d['single_pair_latency_ms'], d['host_ms_per_step'])
PY
  head -1 gpurun_out/r03a_traffic_ro$RO.txt
done
