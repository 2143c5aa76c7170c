cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/r03e_pytest.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/r03e_pytest.txt
tail -5 gpurun_out/r03e_pytest.txt
for MATH in split3 f32; do
  HPL_MATH=$MATH python bench.py --no-cpu-baseline > gpurun_out/r03e_bench_$MATH.json 2> gpurun_out/r03e_bench_$MATH.err
  python - <<PY
import json
d=json.load(open('gpurun_out/r03e_bench_$MATH.json')); r=d['roofline']
print('$MATH', round(d['value'],1), round(d['ms_per_step'],3), {k:r.get(k) for k in ('avg_launch_us','shader_clock_ghz')}, d['single_pair_latency_ms']['forward_ms'], d['pipelined_output_check'])
PY
done
