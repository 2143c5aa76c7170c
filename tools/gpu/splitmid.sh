cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
for V in 0 1; do
echo "== HPL_SPLIT_MID=$V"
for D in frustum surface; do
HPL_SPLIT_MID=$V python tools/chain_run.py $D | tail -1
HPL_SPLIT_MID=$V python bench.py --no-cpu-baseline --steps 200 --data $D | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$D', round(d['value'],1), d['host_ms_per_step']['busy_ms'])"
done
HPL_SPLIT_MID=$V python bench.py --no-cpu-baseline --steps 200 --arch HPLFlowNetShallow --points 4096 | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('shallow', round(d['value'],1))"
done
rocprofv3 --kernel-trace -d gpurun_out/ct_s -o k -- python tools/chain_run.py surface > /dev/null 2>&1
python tools/chain_trace.py $(ls gpurun_out/ct_s/*/k_results.db | head -1) gpurun_out/r02_chain_surface_split.txt | head -16
rm -rf gpurun_out/ct_s
