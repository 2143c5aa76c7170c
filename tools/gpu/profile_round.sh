# The round's profile set (run on the GPU box from the repo root): the default bench line, the kernel table of the same command,
# executed MFMA work (PMC), HBM-side traffic (two PMC passes), the launch-by-launch timelines of one lattice build and one
# forward.  Outputs under gpurun_out/; copy the summaries into profiles/ (mfma_pmc.json / pmc_traffic.json are what bench.py reads).
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
R=${1:-r06}
export TMPDIR=/tmp
# (round 6: stdout carries the short contract line, the full record is bench_detail.json beside bench.py)
python bench.py --detail gpurun_out/${R}_bench_plain_detail.json > gpurun_out/${R}_bench_plain.json 2> gpurun_out/${R}_bench_plain.err
python bench.py --gpus 1 --steps 20 --warmup 5 --detail gpurun_out/${R}_bench_driver_cmd_detail.json > gpurun_out/${R}_bench_driver_cmd.json 2> /dev/null
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_k -o k -- python bench.py --detail gpurun_out/${R}_bench_detail.json > gpurun_out/${R}_bench.json 2> /dev/null
python tools/prof_summary.py $(ls gpurun_out/prof_k/*/k_results.db gpurun_out/prof_k/k_results.db 2>/dev/null | head -1) "python bench.py" > gpurun_out/${R}_kernel_stats.txt
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d gpurun_out/pmc_mfma -o m -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-train-probe --detail '' --no-overlap > /dev/null 2>&1
python tools/pmc_mfma.py $(ls gpurun_out/pmc_mfma/*/m_results.db gpurun_out/pmc_mfma/m_results.db 2>/dev/null | head -1) gpurun_out/${R}_mfma_pmc > /dev/null
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/pmc_fetch -o f -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-train-probe --detail '' --no-overlap > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/pmc_write -o w -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-train-probe --detail '' --no-overlap > /dev/null 2>&1
cp profiles/pmc_traffic.json gpurun_out/${R}_pmc_traffic.json 2>/dev/null; python tools/pmc_traffic.py $(ls gpurun_out/pmc_fetch/*/f_results.db gpurun_out/pmc_fetch/f_results.db 2>/dev/null | head -1) $(ls gpurun_out/pmc_write/*/w_results.db gpurun_out/pmc_write/w_results.db 2>/dev/null | head -1) gpurun_out/${R}_pmc_traffic.json > gpurun_out/${R}_pmc_traffic.txt
rocprofv3 --kernel-trace -d gpurun_out/ft -o ft -- python tools/chain_run.py frustum 8192 > gpurun_out/${R}_chain_run.txt 2>&1
DB=$(ls gpurun_out/ft/*/ft_results.db gpurun_out/ft/ft_results.db 2>/dev/null | head -1)
python tools/forward_trace.py $DB > gpurun_out/${R}_step_timeline.txt
python tools/trace_hbm.py $DB gpurun_out/${R}_trace_hbm.json > gpurun_out/${R}_trace_hbm.txt
# one lattice build task by task (HPL_FUSED_SPLIT=2: every task a launch of its own, its name on stderr)
HPL_FUSED_SPLIT=2 rocprofv3 --kernel-trace -d gpurun_out/lt -o lt -- python tools/lattice_trace.py run 8192 > gpurun_out/${R}_lattice_run.txt 2>&1
python tools/lattice_trace.py show $(ls gpurun_out/lt/*/lt_results.db gpurun_out/lt/lt_results.db 2>/dev/null | head -1) > gpurun_out/${R}_lattice_show.txt
python tools/lattice_tasks.py gpurun_out/${R}_lattice_run.txt gpurun_out/${R}_lattice_show.txt > gpurun_out/${R}_lattice_tasks.txt 2>&1
rm -rf gpurun_out/prof_k gpurun_out/pmc_mfma gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/ft gpurun_out/lt
head -12 gpurun_out/${R}_kernel_stats.txt | cut -c1-160; head -6 gpurun_out/${R}_mfma_pmc.txt | cut -c1-200; head -3 gpurun_out/${R}_pmc_traffic.txt
python - <<PY
import json
for f in ('bench_plain', 'bench_driver_cmd'):
    line=open('gpurun_out/${R}_%s.json' % f).read().strip().splitlines()[-1]; print(f, 'line bytes', len(line))
    d=json.load(open('gpurun_out/${R}_%s_detail.json' % f)); r=d['roofline']
    print(f, round(d['value'],1), d.get('steady'), d.get('forward_only'), d['host_ms_per_step'], {k:r.get(k) for k in ('frac','achieved','avg_launch_us','executed_fraction','executed_source','shader_clock_ghz','frac_at_measured_clock','whole_step','traffic','pmc_note','traffic_note')}, d.get('train'), d.get('cpu_baseline',{}).get('value'), d.get('epe3d'))
PY
