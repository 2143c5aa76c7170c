# round 6, closing set (run on the GPU box from the repo root; outputs under gpurun_out/, copied into profiles/ afterwards):
#   1. the PMC passes + the single-forward trace FIRST, their stamped summaries copied into profiles/ on the box, so that every bench
#      line below reads counters taken from this very build (bench.py uses them only when their stamp is this tree's)
#   2. all GPU tests
#   3. the bench lines (default, the driver's command, the default under rocprofv3 --kernel-trace --stats), lattice timeline
#   4. point-count sweep, the training step (bench line, kernel stats, timeline)
cd $GRAFT_REPO_ROOT
python -c "from hplflownet_amd import build; build.build()" || exit 1      # (a library older than its sources is rebuilt here, not measured)
O=gpurun_out; mkdir -p $O
R=r06
export TMPDIR=/tmp
db() { ls $O/$1/*/$2_results.db $O/$1/$2_results.db 2>/dev/null | head -1; }
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $O/pmc_mfma -o m -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-train-probe --detail '' --no-overlap > /dev/null 2>&1
python tools/pmc_mfma.py $(db pmc_mfma m) $O/${R}_mfma_pmc > /dev/null
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch -o f -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-train-probe --detail '' --no-overlap > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write -o w -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-train-probe --detail '' --no-overlap > /dev/null 2>&1
python tools/pmc_traffic.py $(db pmc_fetch f) $(db pmc_write w) $O/${R}_pmc_traffic.json > $O/${R}_pmc_traffic.txt
rocprofv3 --kernel-trace -d $O/ft -o ft -- python tools/chain_run.py frustum 8192 > $O/${R}_chain_run.txt 2>&1
python tools/forward_trace.py $(db ft ft) > $O/${R}_step_timeline.txt
python tools/trace_hbm.py $(db ft ft) $O/${R}_trace_hbm.json > $O/${R}_trace_hbm.txt
cp $O/${R}_mfma_pmc.json profiles/mfma_pmc.json; cp $O/${R}_pmc_traffic.json profiles/pmc_traffic.json; cp $O/${R}_trace_hbm.json profiles/trace_hbm.json
rm -rf $O/pmc_mfma $O/pmc_fetch $O/pmc_write $O/ft
head -6 $O/${R}_mfma_pmc.txt | cut -c1-200; head -3 $O/${R}_pmc_traffic.txt; tail -1 $O/${R}_step_timeline.txt

python -m pytest tests -m gpu -q > $O/${R}_pytest_gpu.txt 2>&1; echo "pytest rc=$?"; grep -n "passed\|failed" $O/${R}_pytest_gpu.txt | tail -2

python bench.py --detail $O/${R}_bench_plain_detail.json > $O/${R}_bench_plain.json 2> $O/${R}_bench_plain.err
python bench.py --gpus 1 --steps 20 --warmup 5 --detail $O/${R}_bench_driver_cmd_detail.json > $O/${R}_bench_driver_cmd.json 2> /dev/null
rocprofv3 --kernel-trace --stats -d $O/prof_k -o k -- python bench.py --detail $O/${R}_bench_detail.json > $O/${R}_bench.json 2> /dev/null
python tools/prof_summary.py $(db prof_k k) "python bench.py" > $O/${R}_kernel_stats.txt; rm -rf $O/prof_k
head -12 $O/${R}_kernel_stats.txt | cut -c1-160
HPL_FUSED_SPLIT=2 rocprofv3 --kernel-trace -d $O/lt -o lt -- python tools/lattice_trace.py run 8192 > $O/${R}_lattice_run.txt 2>&1
python tools/lattice_trace.py show $(db lt lt) > $O/${R}_lattice_show.txt
python tools/lattice_tasks.py $O/${R}_lattice_run.txt $O/${R}_lattice_show.txt > $O/${R}_lattice_tasks.txt 2>&1; rm -rf $O/lt
python - <<PY
import json
for f in ('bench_plain', 'bench_driver_cmd'):
    line=open('$O/${R}_%s.json' % f).read().strip().splitlines()[-1]; print(f, 'line bytes', len(line))
    d=json.load(open('$O/${R}_%s_detail.json' % f)); r=d['roofline']
    print(f, round(d['value'],1), d.get('steady'), d.get('forward_only'), {k:r.get(k) for k in ('frac','achieved','avg_launch_us','executed_fraction','executed_source','whole_step','traffic')}, d.get('train'), d.get('cpu_baseline',{}).get('value'), d.get('epe3d'), d.get('single_pair_latency_ms'))
PY

python tools/point_sweep.py > $O/${R}_point_count_sweep.txt 2>&1; cat $O/${R}_point_count_sweep.txt
python bench.py --train --steps 40 --warmup 5 --no-cpu-baseline --detail $O/${R}_train_bench_detail.json > $O/${R}_train_bench.json 2>/dev/null; tail -c 600 $O/${R}_train_bench.json; echo
rocprofv3 --kernel-trace --stats -d $O/prof_t -o t -- python bench.py --train --steps 60 --no-cpu-baseline --detail '' > /dev/null 2>&1
python tools/prof_summary.py $(db prof_t t) "python bench.py --train --steps 60 --no-cpu-baseline" > $O/${R}_train_kernel_stats.txt; rm -rf $O/prof_t
PROBE_ONLY=native rocprofv3 --kernel-trace -d $O/prof_n -o n -- python tools/train_native_probe.py > $O/${R}_train_probe.txt 2>&1
python tools/train_timeline.py $(db prof_n n) > $O/${R}_train_timeline.txt; rm -rf $O/prof_n
head -1 $O/${R}_train_timeline.txt; tail -2 $O/${R}_train_timeline.txt
