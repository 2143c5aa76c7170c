# Soak run (on the GPU box from the repo root): the pipelined loop over 5 000 steps on both cloud kinds -- rate, the pipelined output against
# the single-stream forward, device memory (flat: no lattice may be kept alive), range-guard trips.  -> gpurun_out/${R}_soak.txt
cd $GRAFT_REPO_ROOT
python -c "from hplflownet_amd import build; build.build()" || exit 1      # (a library older than its sources is rebuilt here, not measured)
R=${1:-r06}
STEPS=${2:-5000}
mkdir -p gpurun_out
{
for D in frustum surface; do
python bench.py --steps $STEPS --no-cpu-baseline --no-train-probe --data $D --detail gpurun_out/soak_$D.json > /dev/null 2>&1
python -c "
import json
d=json.load(open('gpurun_out/soak_$D.json')); print('$D $STEPS steps:', round(d['value'],1), 'pairs/s', d['pipelined_output_check'], d['device_memory_mb'], 'guard trips', d['config'].get('exact_fallback_launches'))"
done
} > gpurun_out/${R}_soak.txt 2>&1
cat gpurun_out/${R}_soak.txt
