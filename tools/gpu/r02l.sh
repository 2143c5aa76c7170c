cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 1200 python -m pytest tests -m gpu -q -x --durations=5) > gpurun_out/r02l_tests.log 2>&1; tail -12 gpurun_out/r02l_tests.log
for v in 2 3; do
HPL_TAP_GROUPS=$v python bench.py --steps 200 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('bench G=$v', round(d['value'],1), d['host_ms_per_step'], 'us', r.get('avg_launch_us'), r.get('launches_per_step'))"
done
