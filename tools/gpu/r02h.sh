cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(python -m pytest tests/test_gpu_kernels.py tests/test_gpu_plan.py -q -x) 2>&1 | tail -3
for v in "HPL_TAP_GROUPS=2" "HPL_TAP_GROUPS=3"; do
  echo "== $v"
  env $v python bench.py --steps 200 --no-cpu-baseline 2>/dev/null > gpurun_out/r02h_$v.json
  python - <<PY
import json
d=json.load(open('gpurun_out/r02h_$v.json')); r=d['roofline']
print(round(d['value'],1), d['host_ms_per_step'], 'frac',r.get('frac'), 'us',r.get('avg_launch_us'), 'launches', r.get('launches_per_step'), 'clk', r.get('shader_clock_ghz'), 'frac@clk', r.get('frac_at_measured_clock'), 'exec', r.get('executed_fraction'))
PY
done
python bench.py --steps 100 --arch HPLFlowNetShallow --points 4096 --no-cpu-baseline > gpurun_out/r02h_shallow.json 2>gpurun_out/r02h_shallow.err; tail -2 gpurun_out/r02h_shallow.err
python - <<PY
import json
d=json.load(open('gpurun_out/r02h_shallow.json')); r=d['roofline']
print('shallow', round(d['value'],1), d['host_ms_per_step'], r.get('frac'), r.get('kernel'))
PY
