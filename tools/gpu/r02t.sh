cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
# functional run of the multi-rank bench path on one GPU: two ranks, both on cuda:0, gloo for the barrier / max-time
# reduction (RCCL refuses two ranks on one device); the data path has no collective
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 WORLD_SIZE=2 HPL_DIST_BACKEND=gloo LOCAL_RANK=0
RANK=1 python bench.py --gpus 2 --steps 40 --warmup 3 --no-cpu-baseline > gpurun_out/r02t_rank1.out 2> gpurun_out/r02t_rank1.err &
RANK=0 python bench.py --gpus 2 --steps 40 --warmup 3 --no-cpu-baseline > gpurun_out/r02t_rank0.json 2> gpurun_out/r02t_rank0.err
wait
tail -2 gpurun_out/r02t_rank0.err gpurun_out/r02t_rank1.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r02t_rank0.json')); print('2 ranks on one GPU:', round(d['value'],1), d['n_gpus'], d['ms_per_step'], d['host_ms_per_step'], d['pipelined_output_check'])
PY
RANK=1 python bench.py --gpus 2 --train --steps 5 --warmup 1 > gpurun_out/r02t_train1.out 2> gpurun_out/r02t_train1.err &
RANK=0 python bench.py --gpus 2 --train --steps 5 --warmup 1 > gpurun_out/r02t_train0.json 2> gpurun_out/r02t_train0.err
wait
tail -2 gpurun_out/r02t_train0.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r02t_train0.json')); print('train, 2 ranks on one GPU (gloo all-reduce):', round(d['value'],2), d['ms_per_step'])
PY
