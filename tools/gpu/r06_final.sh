# round 6, closing set: all GPU tests, the round's profile set (tools/gpu/profile_round.sh), point-count sweep, the training step
cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
export TMPDIR=/tmp
python -m pytest tests -m gpu -q > $O/r06_pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -3 $O/r06_pytest_gpu.txt
bash tools/gpu/profile_round.sh r06
python tools/point_sweep.py > $O/r06_point_count_sweep.txt 2>&1; cat $O/r06_point_count_sweep.txt
python bench.py --train --steps 40 --warmup 5 --no-cpu-baseline --detail $O/r06_train_bench_detail.json > $O/r06_train_bench.json 2>/dev/null; tail -c 600 $O/r06_train_bench.json; echo
rocprofv3 --kernel-trace --stats -d $O/prof_t -o t -- python bench.py --train --steps 60 --no-cpu-baseline --detail '' > /dev/null 2>&1
python tools/prof_summary.py $(ls $O/prof_t/*/t_results.db $O/prof_t/t_results.db 2>/dev/null | head -1) "python bench.py --train --steps 60 --no-cpu-baseline" > $O/r06_train_kernel_stats.txt; rm -rf $O/prof_t
PROBE_ONLY=native rocprofv3 --kernel-trace -d $O/prof_n -o n -- python tools/train_native_probe.py > $O/r06_train_probe.txt 2>&1
python tools/train_timeline.py $(ls $O/prof_n/*/n_results.db $O/prof_n/n_results.db 2>/dev/null | head -1) > $O/r06_train_timeline.txt; rm -rf $O/prof_n
head -1 $O/r06_train_timeline.txt; tail -2 $O/r06_train_timeline.txt
