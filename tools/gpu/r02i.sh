cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(python -m pytest tests/test_gpu_plan.py -q -x) 2>&1 | tail -15
for args in "" "--data surface" "--arch HPLFlowNetShallow --points 4096"; do
  python bench.py --steps 200 --no-cpu-baseline $args 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('$args', round(d['value'],1), d['host_ms_per_step'], 'frac', r.get('frac'), 'us', r.get('avg_launch_us'), 'clk', r.get('shader_clock_ghz'), 'frac@clk', r.get('frac_at_measured_clock'))"
done
python bench.py --steps 200 --no-cpu-baseline --python-lattice 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('python-lattice', round(d['value'],1), d['host_ms_per_step'])"
