cd $GRAFT_REPO_ROOT
for V in 1 0 1 0; do
HPL_WG3_DENSE=$V python tools/bench_dense.py 2>&1 | tail -1 | cut -c1-330
done
run() { python bench.py --no-cpu-baseline --steps 200 "$@" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), d['pipelined_output_check'])"; }
for V in 1 0 1 0; do echo "HPL_WG3_DENSE=$V frustum/surface"; HPL_WG3_DENSE=$V run; HPL_WG3_DENSE=$V run --data surface; done
