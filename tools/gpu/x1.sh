cd $GRAFT_REPO_ROOT
run() { python bench.py --no-cpu-baseline --steps 200 "$@" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1))"; }
for X in 0 256 0 256 64; do echo "HPL_X_NO1X1=$X frustum / surface"
HPL_X_NO1X1=$X run
HPL_X_NO1X1=$X run --data surface
done
HPL_X_NO1X1=256 python tools/chain_run.py frustum | tail -1
