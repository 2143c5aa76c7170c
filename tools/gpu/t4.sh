cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_multirank.py -m gpu -x -q 2>&1 | tail -5
python bench.py --steps 50 --no-cpu-baseline --data surface 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print(round(d['value'],1), {k: r.get(k) for k in ('frac','executed_source','launches_per_step','whole_step')})"
