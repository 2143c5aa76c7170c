set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
HPL_LIB=$PWD/hplflownet_amd/libhplbcl_timing.so python tools/tile_timing.py > gpurun_out/r02c_tile_timing.txt 2>&1
cat gpurun_out/r02c_tile_timing.txt
python bench.py --steps 100 --arch HPLFlowNetShallow --points 4096 --no-cpu-baseline > gpurun_out/r02c_shallow.json 2>gpurun_out/r02c_shallow.err; tail -3 gpurun_out/r02c_shallow.err
python bench.py --steps 200 --no-cpu-baseline > gpurun_out/r02c_bench.json 2>gpurun_out/r02c_bench.err; tail -3 gpurun_out/r02c_bench.err
python bench.py --steps 200 --no-cpu-baseline --streams 2 > gpurun_out/r02c_bench_s2.json 2>/dev/null
python bench.py --steps 200 --no-cpu-baseline --streams 4 > gpurun_out/r02c_bench_s4.json 2>/dev/null
python - <<'PY'
import json
for f in ('r02c_shallow','r02c_bench','r02c_bench_s2','r02c_bench_s4'):
    try:
        d=json.load(open('gpurun_out/%s.json'%f)); r=d['roofline']; print(f, round(d['value'],1), d['host_ms_per_step'], r.get('frac'), r.get('avg_launch_us'), r.get('in_loop'))
    except Exception as e: print(f, 'ERR', e)
PY
