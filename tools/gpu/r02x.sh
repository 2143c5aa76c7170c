cd $GRAFT_REPO_ROOT
HPL_WG3=1 timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_plan.py -q -x 2>&1 | tail -2
for v in 0 1 0 1; do echo "== HPL_WG3=$v"; HPL_WG3=$v python tools/bench_groups.py 2>&1 | grep -E "groups=2 "; done
HPL_WG3=1 HPL_LIB=$PWD/hplflownet_amd/libhplbcl_timing.so python tools/tile_timing.py 2>&1 | grep -E "blur|wall span|residency|cycles per slice|share of" | head -10
for v in 0 1; do
HPL_WG3=$v python bench.py --steps 200 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('bench wg3=$v', round(d['value'],1), 'us', r.get('avg_launch_us'), 'frac', r.get('frac'), 'clk', r.get('shader_clock_ghz'))"
done
