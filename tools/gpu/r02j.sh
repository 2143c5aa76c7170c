cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_plan.py -q -x) 2>&1 | tail -3
for v in 1 0; do
  echo "== HPL_PERSISTENT=$v"
  HPL_PERSISTENT=$v timeout 300 python tools/bench_groups.py 2>&1 | grep -E "groups=(1|2) "
done
HPL_LIB=$PWD/hplflownet_amd/libhplbcl_timing.so python tools/pers_timing.py 2>&1 | grep -v amdgpu | grep -v queue
for v in 1 0; do
HPL_PERSISTENT=$v python bench.py --steps 200 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('bench pers=$v', round(d['value'],1), d['host_ms_per_step'], 'frac', r.get('frac'), 'us', r.get('avg_launch_us'), 'clk', r.get('shader_clock_ghz'), 'frac@clk', r.get('frac_at_measured_clock'))"
done
