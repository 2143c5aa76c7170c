cd $GRAFT_REPO_ROOT
for args in "--lattice-streams 2 --lattice-depth 3" "--data surface --streams 4" "--data surface --streams 5 --lattice-streams 2 --lattice-depth 3" "--arch HPLFlowNetShallow --points 4096 --lattice-streams 2 --lattice-depth 3" "--arch HPLFlowNetShallow --points 4096 --lattice-streams 2 --lattice-depth 4 --streams 4" "--arch HPLFlowNetShallow --points 4096 --lattice-streams 3 --lattice-depth 4 --streams 4"; do
python bench.py --steps 300 --no-cpu-baseline $args 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('bench [$args]', round(d['value'],1), d['host_ms_per_step'])"
done
