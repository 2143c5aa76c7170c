cd $GRAFT_REPO_ROOT
python -c "import torch; print(torch.cuda.Stream.priority_range())"
run() { python bench.py --no-cpu-baseline --steps 200 "$@" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1))"; }
for i in 1 2; do
echo "default: $(run) | fwd prio 0,-1,-2: $(HPL_FWD_PRIO=0,-1,-2 run) | lattice normal (HPL_PRIO=none): $(HPL_PRIO=none run) | fwd all high: $(HPL_PRIO=forward run) | 1,0,-1 lat none: $(HPL_PRIO=none HPL_FWD_PRIO=1,0,-1 run)"
done
