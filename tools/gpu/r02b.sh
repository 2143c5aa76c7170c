set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(python -m pytest tests/test_gpu_plan.py tests/test_gpu_train_loop.py tests/test_gpu_layers.py tests/test_gpu_autograd.py -q -x --durations=5) > gpurun_out/r02b_tests.log 2>&1
tail -25 gpurun_out/r02b_tests.log
python tools/host_time.py > gpurun_out/r02b_host.txt 2>&1
cat gpurun_out/r02b_host.txt
python bench.py --steps 100 > gpurun_out/r02b_bench.json 2> gpurun_out/r02b_bench.err; tail -3 gpurun_out/r02b_bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r02b_bench.json')); print(d['value'], d['host_ms_per_step'], d['roofline']['frac'], d['pipelined_output_check'])
PY
python bench.py --steps 100 --data surface --no-cpu-baseline > gpurun_out/r02b_surface.json 2>/dev/null
python bench.py --steps 100 --arch HPLFlowNetShallow --points 4096 --no-cpu-baseline > gpurun_out/r02b_shallow.json 2>/dev/null
python - <<'PY'
import json
for f in ('r02b_surface','r02b_shallow'):
    d=json.load(open('gpurun_out/%s.json'%f)); print(f, d['value'], d['host_ms_per_step'])
PY
