# The driver's command five times on one box (boxes of the pool and runs on one box differ by a few per cent): every run's line is kept,
# the MEDIAN run by `value` becomes the round's r06_bench_driver_cmd*.json
cd $GRAFT_REPO_ROOT
python -c "from hplflownet_amd import build; build.build()" || exit 1      # (a library older than its sources is rebuilt here, not measured)
O=gpurun_out; mkdir -p $O
R=${1:-r06}
for i in 1 2 3 4 5; do
  python bench.py --gpus 1 --steps 20 --warmup 5 --detail $O/${R}_drv${i}_detail.json > $O/${R}_drv${i}.json 2> /dev/null
done
python - <<PY
import json, shutil
runs = []
for i in range(1, 6):
    d = json.load(open('$O/${R}_drv%d_detail.json' % i))
    runs.append((d['value'], i, d))
out = ['# tools/gpu/driver_cmd_runs.sh: python bench.py --gpus 1 --steps 20 --warmup 5, five runs back to back on one box',
       '# run  value (20 steps)  steady (>= 1 s)  forward-only  dominant launch us  roofline.frac  train ms  board W']
for v, i, d in runs:
    out.append('%d  %.1f  %.1f  %.1f  %.1f  %.3f  %.2f  %s' % (i, v, d['steady']['value'], d['forward_only']['pairs_per_s'], d['roofline']['avg_launch_us'],
               d['roofline']['frac'], (d.get('train') or {}).get('ms_per_step', float('nan')), (d.get('power') or {}).get('package_w')))
runs.sort()
med = runs[2]
out.append('# median run: %d (%.1f pairs/s); min %.1f, max %.1f' % (med[1], med[0], runs[0][0], runs[-1][0]))
open('$O/${R}_driver_cmd_runs.txt', 'w').write('\n'.join(out) + '\n')
print('\n'.join(out))
for suf in ('.json', '_detail.json', '_detail_bf16x3.json'):
    shutil.copy('$O/${R}_drv%d%s' % (med[1], suf), '$O/${R}_bench_driver_cmd%s' % suf)
PY
