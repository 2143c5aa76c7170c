# round 6: transposed accumulators, 16-byte epilogue accesses, against the build of the commit before (libhplbcl_before_epi.so)
# when there is no residual) against the build before (hplflownet_amd/libhplbcl_before_epi.so)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06t; mkdir -p $O
export TMPDIR=/tmp
python -m pytest tests/test_gpu_split3.py tests/test_gpu_kernels.py tests/test_gpu_layers.py tests/test_gpu_plan.py tests/test_gpu_bench_size.py tests/test_gpu_train_plan.py tests/test_variants.py -x -q > $O/pytest_first.txt 2>&1; echo "first rc=$?"; tail -3 $O/pytest_first.txt
HPL_LIB=$PWD/hplflownet_amd/libhplbcl_probe.so python tools/tile_phase_probe.py 2>&1 | grep -v amdgpu.ids | tee $O/tile_phase_probe.txt
for lib in libhplbcl.so libhplbcl_before_epi.so libhplbcl.so libhplbcl_before_epi.so; do
  HPL_LIB=$PWD/hplflownet_amd/$lib CASES="bcn1_ g0,bcn1_ g1,bcn2_ g0,bcn2_ g1,1x1" REPS=10 python tools/bench_split3.py 2>&1 | grep -v amdgpu.ids | sed "s#^#$lib #"
done > $O/split3_ab.txt; cat $O/split3_ab.txt
run() { python bench.py --steps 300 --warmup 10 --no-cpu-baseline --no-train-probe --detail '' 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), (d.get('steady') or {}).get('value'), (d.get('forward_only') or {}).get('pairs_per_s'), (d.get('single_pair_latency_ms') or {}).get('forward_ms'))"; }
for rep in 1 2 3; do
  echo "new epilogues:  $(run)"
  echo "before:         $(HPL_LIB=$PWD/hplflownet_amd/libhplbcl_before_epi.so run)"
done > $O/epilogue_ab.txt; cat $O/epilogue_ab.txt
