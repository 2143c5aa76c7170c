# round 6: persistent guard launches (128 workgroups); guard off / on traces and pipelined rates in one call
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06i; mkdir -p $O
export TMPDIR=/tmp
python -m pytest tests/test_gpu_split3.py tests/test_gpu_kernels.py -x -q > $O/pytest_first.txt 2>&1; echo "first rc=$?"; tail -3 $O/pytest_first.txt
for g in 0 1; do
  HPL_RANGE_GUARD=$g rocprofv3 --kernel-trace -d $O/ft$g -o ft -- python tools/chain_run.py frustum 8192 > $O/chain_run_$g.txt 2>&1
  DB=$(ls $O/ft$g/*/ft_results.db $O/ft$g/ft_results.db 2>/dev/null | head -1)
  python tools/forward_trace.py $DB > $O/step_timeline_guard$g.txt; rm -rf $O/ft$g
  head -1 $O/step_timeline_guard$g.txt; tail -1 $O/step_timeline_guard$g.txt
done
for g in 0 1 0 1; do echo "HPL_RANGE_GUARD=$g"; HPL_RANGE_GUARD=$g python bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-train-probe --detail '' 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), d.get('steady'), d.get('forward_only'), d.get('single_pair_latency_ms'))"; done > $O/guard_ab_bench.txt; cat $O/guard_ab_bench.txt
