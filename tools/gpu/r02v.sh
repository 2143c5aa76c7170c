cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
for args in "" "--no-lattice-thread" "--python-lattice" "--python-forward" "--python-forward --python-lattice"; do
python bench.py --no-cpu-baseline --steps 20 --warmup 5 $args 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('rep$rep [$args]', round(d['value'],1), {k:round(v,2) for k,v in d['host_ms_per_step'].items()})"
done
done
