cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for D in frustum surface; do
python bench.py --steps 20000 --no-cpu-baseline --data $D > gpurun_out/r05y_soak_$D.json 2>/dev/null
python -c "
import json
d=json.load(open('gpurun_out/r05y_soak_$D.json')); print('$D 20000 steps:', round(d['value'],1), 'pairs/s', d['pipelined_output_check'], d['device_memory_mb'], d['roofline'].get('frac'))"
done
python bench.py --steps 200 --no-cpu-baseline | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('200 steps:', round(d['value'],1), d['device_memory_mb'])"
