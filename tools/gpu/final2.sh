cd $GRAFT_REPO_ROOT
(timeout 1500 python -m pytest tests -m gpu -q -x) > gpurun_out/r02z_tests.log 2>&1; grep -E "passed|failed" gpurun_out/r02z_tests.log | tail -1
bash tools/gpu/profile_round.sh r02z
for g in 2 3; do HPL_TAP_GROUPS=$g python bench.py --steps 200 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('G=$g', round(d['value'],1), 'us', round(r.get('avg_launch_us'),1), r.get('launches_per_step'))"; done
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r02z_bench_driver_cmd.json 2>/dev/null
python -c "
import json
d=json.load(open('gpurun_out/r02z_bench_driver_cmd.json')); print('driver cmd', round(d['value'],1), d['ms_per_step'])"
