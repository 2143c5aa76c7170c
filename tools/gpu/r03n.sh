cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_t -o k -- python bench.py --train --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r03n_train.json 2> /dev/null
python tools/prof_summary.py $(ls gpurun_out/prof_t/*/k_results.db gpurun_out/prof_t/k_results.db 2>/dev/null | head -1) "python bench.py --train" > gpurun_out/r03n_train_kernel_stats.txt
rm -rf gpurun_out/prof_t
head -32 gpurun_out/r03n_train_kernel_stats.txt | cut -c1-150
