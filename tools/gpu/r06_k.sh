# round 6: which part of the range guard costs the pipelined loop: words only (2), second launches only (3), no epilogue words (4)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06k; mkdir -p $O
export TMPDIR=/tmp
run() { python bench.py --steps 300 --warmup 10 --no-cpu-baseline --no-train-probe --detail '' 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), (d.get('steady') or {}).get('value'), (d.get('forward_only') or {}).get('pairs_per_s'), (d.get('single_pair_latency_ms') or {}).get('forward_ms'))"; }
for rep in 1 2 3; do
  for g in 0 1 2 3 4; do echo "HPL_RANGE_GUARD=$g: $(HPL_RANGE_GUARD=$g run)"; done
done > $O/guard_parts_ab.txt; cat $O/guard_parts_ab.txt
