cd $GRAFT_REPO_ROOT
run() { python bench.py --no-cpu-baseline --steps 200 "$@" | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), round(d['host_ms_per_step']['busy_ms'],2))"; }
for Q in 4 8; do
for S in 3 4 6; do
echo "== GPU_MAX_HW_QUEUES=$Q streams=$S: frustum, surface, shallow4096"
GPU_MAX_HW_QUEUES=$Q run --streams $S
GPU_MAX_HW_QUEUES=$Q run --streams $S --data surface
GPU_MAX_HW_QUEUES=$Q run --streams $S --arch HPLFlowNetShallow --points 4096
done
done
