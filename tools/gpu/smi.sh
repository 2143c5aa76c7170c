cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
which rocm-smi amd-smi 2>&1 | head -2
rocm-smi --showclocks --showtemp --showpower --json 2>/dev/null | head -c 1500; echo
python bench.py --steps 12000 --no-cpu-baseline > gpurun_out/soak2.json 2>/dev/null &
BP=$!
sleep 25
for i in 1 2 3 4 5 6; do
rocm-smi --showclocks --showtemp --showpower --json 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin); c=d.get('card0',{})
keep={k:v for k,v in c.items() if any(x in k.lower() for x in ('sclk','mclk','fclk','power','temperature (sensor junction)','temperature (sensor memory)','hbm'))}
print(keep)"
sleep 8
done
wait $BP
python -c "
import json
d=json.load(open('gpurun_out/soak2.json')); print(round(d['value'],1), d['roofline']['avg_launch_us'], d['roofline'].get('shader_clock_ghz'))"
