# round 6: queue gaps of the pipelined loop (which launches wait for a CU of their own)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06m; mkdir -p $O
export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $O/kt -o k -- python bench.py --steps 400 --warmup 10 --no-cpu-baseline --no-train-probe --detail '' > $O/bench_line.txt 2>/dev/null
DB=$(ls $O/kt/*/k_results.db $O/kt/k_results.db 2>/dev/null | head -1)
python tools/queue_gaps.py $DB 0.10 0.35 > $O/queue_gaps.txt; rm -rf $O/kt
head -40 $O/queue_gaps.txt | cut -c1-130; tail -c 600 $O/bench_line.txt
