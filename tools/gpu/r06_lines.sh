# round 6: the two bench lines again, with the PMC / trace files of the closing set in place (their stamp matches the sources now)
cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
python bench.py --gpus 1 --steps 20 --warmup 5 --detail $O/r06_bench_driver_cmd_detail.json > $O/r06_bench_driver_cmd.json 2> $O/r06_bench_err.txt; echo "rc=$?"
python bench.py --detail $O/r06_bench_plain_detail.json > $O/r06_bench_plain.json 2>/dev/null; echo "rc=$?"
tail -c 3500 $O/r06_bench_driver_cmd.json
