cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
for S in 2 3 4 5; do python bench.py --streams $S --no-cpu-baseline --no-train-probe > gpurun_out/r03o_s$S.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/r03o_s$S.json')); print('streams $S', round(d['value'],1), round(d['ms_per_step'],3), d['host_ms_per_step']['busy_ms'], d['device_memory_mb']['max_allocated'])"; done
python bench.py --streams 3 --lattice-depth 3 --no-cpu-baseline --no-train-probe > gpurun_out/r03o_d3.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/r03o_d3.json')); print('streams 3 depth 3', round(d['value'],1), round(d['ms_per_step'],3))"
python bench.py --data surface --no-cpu-baseline --no-train-probe > gpurun_out/r03o_surface.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/r03o_surface.json')); print('surface', round(d['value'],1), round(d['ms_per_step'],3))"
python bench.py --arch HPLFlowNetShallow --points 4096 --no-cpu-baseline --no-train-probe > gpurun_out/r03o_shallow.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/r03o_shallow.json')); print('shallow 4096', round(d['value'],1), round(d['ms_per_step'],3))"
