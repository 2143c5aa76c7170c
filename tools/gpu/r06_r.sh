# round 6: data gradients of the training program without the range guard (HPL_FLAG_NOGUARD); tile phase probe of the wide kernel
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06r; mkdir -p $O
export TMPDIR=/tmp
python -m pytest tests/test_gpu_train_plan.py tests/test_gpu_train_loop.py tests/test_gpu_autograd.py tests/test_gpu_train_ops.py tests/test_gpu_plan.py -x -q > $O/pytest_first.txt 2>&1; echo "first rc=$?"; tail -3 $O/pytest_first.txt
python -m pytest tests/test_gpu_bench_size.py tests/test_gpu_math_modes.py -x -q > $O/pytest_second.txt 2>&1; echo "second rc=$?"; tail -3 $O/pytest_second.txt
python bench.py --train --steps 40 --warmup 5 --no-cpu-baseline --detail $O/train_bench_detail.json > $O/train_bench.json 2>/dev/null; tail -c 1200 $O/train_bench.json; echo
PROBE_ONLY=native rocprofv3 --kernel-trace -d $O/prof_n -o n -- python tools/train_native_probe.py > $O/train_probe.txt 2>&1
python tools/train_timeline.py $(ls $O/prof_n/*/n_results.db $O/prof_n/n_results.db 2>/dev/null | head -1) > $O/train_timeline.txt; rm -rf $O/prof_n
head -1 $O/train_timeline.txt; tail -2 $O/train_timeline.txt; tail -3 $O/train_probe.txt
HPL_LIB=$PWD/hplflownet_amd/libhplbcl_probe.so python tools/tile_phase_probe.py 2>&1 | grep -v amdgpu.ids | tee $O/tile_phase_probe.txt
