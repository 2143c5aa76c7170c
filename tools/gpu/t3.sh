cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "split_k_folded" 2>&1 | tail -5
for V in "HPL_SPLIT_FOLD=1" "HPL_SPLIT_FOLD=0" "HPL_SPLIT_FOLD=0 HPL_SPLIT_MID=0" "HPL_SPLIT_FOLD=1 HPL_SPLIT_MID=0"; do
echo "== $V"
env $V python tools/chain_run.py frustum | tail -1
env $V python tools/chain_run.py surface | tail -1
done
