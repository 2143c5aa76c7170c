set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(python -m pytest tests/test_gpu_kernels.py tests/test_gpu_plan.py tests/test_gpu_autograd.py tests/test_gpu_layers.py -q -x) > gpurun_out/r02d_tests.log 2>&1
tail -15 gpurun_out/r02d_tests.log
for v in 0 1; do
  if [ $v = 1 ]; then export HPL_NO_TILES=1; fi
  echo "HPL_NO_TILES=$HPL_NO_TILES"
  python tools/host_time.py 2>&1 | grep -v amdgpu
  python bench.py --steps 200 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('bench', round(d['value'],1), d['host_ms_per_step'], r.get('frac'), r.get('avg_launch_us'), (r.get('in_loop') or {}).get('avg_launch_us'))"
done > gpurun_out/r02d_ab.txt 2>&1
cat gpurun_out/r02d_ab.txt
