# round 6: what the guard's (almost always empty) second launches cost the pipelined loop, by their workgroup count; event scope A/B
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06j; mkdir -p $O
export TMPDIR=/tmp
run() { python bench.py --steps 300 --warmup 10 --no-cpu-baseline --no-train-probe --detail '' 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), (d.get('steady') or {}).get('value'), (d.get('forward_only') or {}).get('pairs_per_s'), (d.get('single_pair_latency_ms') or {}).get('forward_ms'))"; }
for rep in 1 2 3; do
  echo "guard off:        $(HPL_RANGE_GUARD=0 run)"
  for w in 8 32 128 1024; do echo "guard on, $w wgs: $(HPL_GUARD_WGS=$w run)"; done
  echo "guard on, 128 wgs, system-scope events: $(HPL_EVENT_SCOPE=system run)"
done > $O/guard_wgs_ab.txt; cat $O/guard_wgs_ab.txt
