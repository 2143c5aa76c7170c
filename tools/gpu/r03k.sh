cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
export ROUNDS=4 CASES="bcn1_ g,bcn2_ g"
for i in 1 2; do
HPL_LIB=$PWD/hplflownet_amd/libhplbcl_base.so python tools/bench_split3.py 2>&1 | grep "split3 " | sed 's/^/base /' | cut -c1-170
python tools/bench_split3.py 2>&1 | grep "split3 " | sed 's/^/new  /' | cut -c1-170
done
