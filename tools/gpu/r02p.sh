cd $GRAFT_REPO_ROOT
for args in "" "--no-lattice" "--lattice-depth 3" "--streams 2" "--streams 4" "--pool 8"; do
python bench.py --steps 200 --no-cpu-baseline $args 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('bench [$args]', round(d['value'],1), d['host_ms_per_step'], 'us', r.get('avg_launch_us'), 'in_loop', (r.get('in_loop') or {}).get('avg_launch_us'))"
done
