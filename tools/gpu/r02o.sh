cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_plan.py -q -x) 2>&1 | tail -3
for i in 1 2; do
python bench.py --steps 200 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('bench', round(d['value'],1), d['host_ms_per_step'], 'us', r.get('avg_launch_us'), 'frac', r.get('frac'))"
done
python bench.py --steps 200 --no-cpu-baseline --data surface 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('surface', round(d['value'],1), d['host_ms_per_step'])"
python bench.py --steps 200 --no-cpu-baseline --arch HPLFlowNetShallow --points 4096 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('shallow', round(d['value'],1), d['host_ms_per_step'])"
