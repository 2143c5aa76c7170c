cd $GRAFT_REPO_ROOT
run() { python bench.py --no-cpu-baseline --steps 200 "$@" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), round(d['host_ms_per_step']['busy_ms'],2))"; }
for X in 0 1; do echo "HPL_X_SKIP=$X surface / shallow4096 / frustum"
HPL_X_SKIP=$X run --data surface
HPL_X_SKIP=$X run --arch HPLFlowNetShallow --points 4096
HPL_X_SKIP=$X run
done
