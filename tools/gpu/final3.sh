cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
(timeout 1500 python -m pytest tests -m gpu -q -x) > gpurun_out/r02z_tests.log 2>&1; grep -E "passed|failed" gpurun_out/r02z_tests.log | tail -1
bash tools/gpu/profile_round.sh r02z
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r02z_bench_driver_cmd.json 2>/dev/null
python -c "
import json
d=json.load(open('gpurun_out/r02z_bench_driver_cmd.json')); print('driver cmd', round(d['value'],1), d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('whole_step'))"
python bench.py --data surface > gpurun_out/r02z_surface_bench.json 2>/dev/null
python bench.py --arch HPLFlowNetShallow --points 4096 > gpurun_out/r02z_shallow_n4096_bench.json 2>/dev/null
python bench.py --arch HPLFlowNetShallow --points 4096 --lattice-thread > gpurun_out/r02z_shallow_n4096_thread_bench.json 2>/dev/null
python bench.py --train > gpurun_out/r02z_train_bench.json 2>/dev/null
for f in surface shallow_n4096 shallow_n4096_thread train; do python -c "
import json
d=json.loads(open('gpurun_out/r02z_${f}_bench.json').read().strip().splitlines()[-1]); print('$f', round(d['value'],1), d['unit'], round(d['ms_per_step'],3))"; done
