cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python bench.py --train --steps 30 --warmup 5 2>gpurun_out/r02m_train.err > gpurun_out/r02m_train.json; tail -2 gpurun_out/r02m_train.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r02m_train.json')); print('train', d['value'], d['ms_per_step'], d['host_ms_per_step'])
for k,v in d['kernels'].items(): print(k, round(v['ms_per_step'],3), v['launches_per_step'])
PY
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_train -o t -- python $GRAFT_REPO_ROOT/bench.py --train --steps 10 --warmup 3 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python tools/prof_summary.py $(ls gpurun_out/prof_train/*/t_results.db gpurun_out/prof_train/t_results.db 2>/dev/null | head -1) "python bench.py --train --steps 10 --warmup 3" > gpurun_out/r02m_train_kernel_stats.txt
head -40 gpurun_out/r02m_train_kernel_stats.txt | cut -c1-170
rm -rf gpurun_out/prof_train
