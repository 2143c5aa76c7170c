cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_split3.py tests/test_gpu_plan.py -x -q 2>&1 | tail -2
for MR in 8192 9500 16384; do
  HPL_SPLIT3_MIN_ROWS=$MR python bench.py --no-cpu-baseline --no-train-probe > gpurun_out/r03j_bench_$MR.json 2> gpurun_out/r03j_bench_$MR.err
  python - <<PY
import json
d=json.load(open('gpurun_out/r03j_bench_$MR.json')); r=d['roofline']
print('min_rows $MR', round(d['value'],1), round(d['ms_per_step'],3), d['single_pair_latency_ms']['forward_ms'], d['single_pair_latency_ms']['lattice_build_ms'], {k:(round(v['ms_per_step'],3)) for k,v in d['kernels'].items() if 'mid' in k or k in ('gconv_64x64_g','gconv_64x64_d')})
PY
done
