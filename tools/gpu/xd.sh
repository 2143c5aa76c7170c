cd $GRAFT_REPO_ROOT
run() { python bench.py --no-cpu-baseline --steps 200 "$@" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), round(d['host_ms_per_step']['busy_ms'],2))"; }
for X in 0 6 0 6; do echo "HPL_X_DUMMY=$X (x7 tap orders per pair) frustum / surface"
HPL_X_DUMMY=$X run
HPL_X_DUMMY=$X run --data surface
done
