# round 6: the guard's second pass inside the launch (a second body behind the first, tile by tile) against the second launch
# (hplflownet_amd/libhplbcl_launchguard.so: the build of the commit before) and against no guard
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06n; mkdir -p $O
export TMPDIR=/tmp
python -m pytest tests/test_gpu_split3.py tests/test_gpu_kernels.py -x -q > $O/pytest_first.txt 2>&1; echo "first rc=$?"; tail -3 $O/pytest_first.txt
for lib in hplflownet_amd/libhplbcl.so hplflownet_amd/libhplbcl_launchguard.so hplflownet_amd/libhplbcl.so hplflownet_amd/libhplbcl_launchguard.so; do
  HPL_LIB=$PWD/$lib CASES="bcn1_ g0,bcn1_ g1,bcn2_ g0,bcn2_ g1,1x1" REPS=10 python tools/bench_split3.py 2>&1 | grep -v amdgpu.ids | sed "s#^#$(basename $lib) #"
done > $O/split3_ab.txt; cat $O/split3_ab.txt
run() { python bench.py --steps 300 --warmup 10 --no-cpu-baseline --no-train-probe --detail '' 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), (d.get('steady') or {}).get('value'), (d.get('forward_only') or {}).get('pairs_per_s'), (d.get('single_pair_latency_ms') or {}).get('forward_ms'))"; }
for rep in 1 2 3; do
  echo "guard off:                 $(HPL_RANGE_GUARD=0 run)"
  echo "guard on, second body:     $(run)"
  echo "guard on, second launch:   $(HPL_LIB=$PWD/hplflownet_amd/libhplbcl_launchguard.so run)"
done > $O/guard_inkernel_ab.txt; cat $O/guard_inkernel_ab.txt
