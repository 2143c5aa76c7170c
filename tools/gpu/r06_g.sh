# round 6: where the range guard's cost sits -- forward traces with the guard off / on in one call, hpl_amax_rows grid caps
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06g; mkdir -p $O
export TMPDIR=/tmp
for g in 0 1; do
  HPL_RANGE_GUARD=$g rocprofv3 --kernel-trace -d $O/ft$g -o ft -- python tools/chain_run.py frustum 8192 > $O/chain_run_$g.txt 2>&1
  DB=$(ls $O/ft$g/*/ft_results.db $O/ft$g/ft_results.db 2>/dev/null | head -1)
  python tools/forward_trace.py $DB > $O/step_timeline_guard$g.txt; rm -rf $O/ft$g
  head -1 $O/step_timeline_guard$g.txt; tail -1 $O/step_timeline_guard$g.txt
done
for cap in 256 512 1024 2048; do
  HPL_AMAXR_CAP=$cap rocprofv3 --kernel-trace -d $O/fc -o ft -- python tools/chain_run.py frustum 8192 > /dev/null 2>&1
  DB=$(ls $O/fc/*/ft_results.db $O/fc/ft_results.db 2>/dev/null | head -1)
  python tools/forward_trace.py $DB > $O/step_timeline_cap$cap.txt; rm -rf $O/fc
  echo "cap $cap: $(grep k_amax_rows $O/step_timeline_cap$cap.txt | awk '{s+=$3} END {print s}') us in $(grep -c k_amax_rows $O/step_timeline_cap$cap.txt) launches"
done
