# round 6 (experiment): the forward in two priority phases -- launch-bound part on the high-priority stream, from the first wide conv on a
# normal-priority stream (HPL_PHASE_SPLIT=1; HPL_PHASE_BACK=k: cut k ops earlier)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06o; mkdir -p $O
export TMPDIR=/tmp
run() { python bench.py --steps 300 --warmup 10 --no-cpu-baseline --no-train-probe --detail '' 2>$O/err.txt | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), (d.get('steady') or {}).get('value'), (d.get('forward_only') or {}).get('pairs_per_s'), (d.get('single_pair_latency_ms') or {}).get('forward_ms'), d.get('pipelined_output_check'))"; }
for rep in 1 2 3; do
  echo "one stream:                  $(run)"
  echo "two phases:                  $(HPL_PHASE_SPLIT=1 run)"
  echo "two phases, lo = low prio:   $(HPL_PHASE_SPLIT=1 HPL_PHASE_LO_PRIO=1 run)"
done > $O/phase_split_ab.txt; cat $O/phase_split_ab.txt; tail -3 $O/err.txt
