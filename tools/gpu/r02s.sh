cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
(timeout 1500 python -m pytest tests -m gpu -q -x) > gpurun_out/r02s_tests.log 2>&1; tail -4 gpurun_out/r02s_tests.log
python bench.py --gpus 1 --steps 20 --warmup 3 > gpurun_out/r02s_bench_driver_like.json 2>/dev/null
python - <<'PY'
import json
d=json.load(open('gpurun_out/r02s_bench_driver_like.json')); r=d['roofline']
print(round(d['value'],1), d['ms_per_step'], d['host_ms_per_step'], {k:r.get(k) for k in ('frac','achieved','avg_launch_us','shader_clock_ghz','frac_at_measured_clock','traffic')}, d['epe3d'], d['cpu_baseline']['value'])
PY
