cd $GRAFT_REPO_ROOT
for i in 1 2 3; do for g in 2 3; do HPL_TAP_GROUPS=$g python bench.py --steps 200 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('G=$g', round(d['value'],1), 'us', round(r.get('avg_launch_us'),1), r.get('launches_per_step'), round(r['frac'],3))"; done; done
for s in 1 2; do python bench.py --steps 200 --no-cpu-baseline --streams $s 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('streams $s', round(d['value'],1))"; done
