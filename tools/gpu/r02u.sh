cd $GRAFT_REPO_ROOT
for v in "HPL_PRIO=lattice" "HPL_PRIO=none" "HPL_PRIO=forward"; do
env $v python bench.py --steps 200 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('bench [$v]', round(d['value'],1), d['host_ms_per_step'])"
done
for args in "--steps 20 --warmup 3" "--steps 20 --warmup 3 --lattice-depth 3" "--steps 20 --warmup 3 --python-forward --python-lattice"; do
python bench.py --no-cpu-baseline $args 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('bench [$args]', round(d['value'],1), d['host_ms_per_step'])"
done
