cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
for NB in 4 3 4 3; do HPL_SPLIT3_NB=$NB python bench.py --no-cpu-baseline --no-train-probe > gpurun_out/r03q_nb$NB.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/r03q_nb$NB.json')); print('NB $NB', round(d['value'],1), round(d['ms_per_step'],3), d['single_pair_latency_ms']['forward_ms'], d['roofline']['avg_launch_us'], d['roofline']['in_loop']['avg_launch_us'])"; done
