# round 6, first call: gpu tests, the driver's bench command, splat variants, XCD order A/B of the wide launches
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06a; mkdir -p $O
export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest_gpu.txt
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.txt 2> $O/bench_err.txt; echo "bench rc=$?"; tail -c 3300 $O/bench_driver_cmd.txt; cp bench_detail.json $O/bench_driver_cmd_detail.json
python tools/bench_splat_variants.py > $O/splat_variants.txt 2>&1; tail -80 $O/splat_variants.txt
for o in 0 1 0 1; do HPL_XCD_ORDER=$o CASES="bcn1_ g,bcn2_ g,1x1" REPS=10 python tools/bench_split3.py 2>&1 | sed "s/^/xcd_order=$o /" >> $O/xcd_order_ab.txt; done; cat $O/xcd_order_ab.txt
