cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/tests.log 2>&1; grep -n "passed\|failed\|Error\|error" gpurun_out/tests.log | tail -8
run() { python bench.py --no-cpu-baseline --steps 200 "$@" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), round(d['host_ms_per_step']['busy_ms'],2), d['pipelined_output_check'])"; }
for i in 1 2; do
run --data surface
run --arch HPLFlowNetShallow --points 4096
run
done
python tools/chain_run.py frustum | tail -1
python tools/chain_run.py surface | tail -1
