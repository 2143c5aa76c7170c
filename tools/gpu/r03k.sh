cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_split3.py -q 2>&1 | tail -2
HPL_SPLIT3_NB=3 timeout 600 python -m pytest tests/test_gpu_split3.py -q 2>&1 | tail -1
export ROUNDS=4 CASES="bcn1_ g,bcn2_ g,dense"
for i in 1 2; do
HPL_SPLIT3_NB=3 python tools/bench_split3.py 2>&1 | grep "split3 " | sed 's/^/nb3 /' | cut -c1-170
python tools/bench_split3.py 2>&1 | grep "split3 " | sed 's/^/nb4 /' | cut -c1-170
done
