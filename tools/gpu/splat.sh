cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "splat" 2>&1 | tail -2
for L in hplflownet_amd/libhplbcl_old.so hplflownet_amd/libhplbcl.so; do
echo "== $L"
HPL_LIB=$GRAFT_REPO_ROOT/$L python tools/bench_splat_slice.py --reps 50 2>&1 | grep -i "splat" | head -12
done
