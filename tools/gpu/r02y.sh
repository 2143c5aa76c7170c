cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do for v in 0 1; do
HPL_WG3=$v python bench.py --steps 200 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('rep$rep wg3=$v', round(d['value'],1), 'us', round(r.get('avg_launch_us'),1), 'frac', round(r.get('frac'),3), 'inloop', round((r.get('in_loop') or {}).get('avg_launch_us',0),1))"
done; done
for v in 0 1; do HPL_WG3=$v python bench.py --train --steps 30 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('train wg3=$v', round(d['ms_per_step'],2))"; done
