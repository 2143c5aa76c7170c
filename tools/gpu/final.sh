cd $GRAFT_REPO_ROOT
bash tools/gpu/profile_round.sh r02z
python bench.py --steps 200 --no-cpu-baseline --data surface > gpurun_out/r02z_surface_bench.json 2>/dev/null
python bench.py --steps 200 --no-cpu-baseline --arch HPLFlowNetShallow --points 4096 > gpurun_out/r02z_shallow_n4096_bench.json 2>/dev/null
python bench.py --steps 200 --no-cpu-baseline --arch HPLFlowNetShallow --points 4096 --lattice-thread > gpurun_out/r02z_shallow_n4096_thread_bench.json 2>/dev/null
python bench.py --train --steps 30 --warmup 5 > gpurun_out/r02z_train_bench.json 2>/dev/null
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r02z_bench_driver_cmd.json 2>/dev/null
python - <<'PY'
import json
for f in ('r02z_surface_bench','r02z_shallow_n4096_bench','r02z_shallow_n4096_thread_bench','r02z_train_bench','r02z_bench_driver_cmd'):
    d=json.load(open('gpurun_out/%s.json'%f)); print(f, round(d['value'],1), round(d['ms_per_step'],3), d.get('host_ms_per_step'))
PY
