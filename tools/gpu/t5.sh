cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_multirank.py -m gpu -x -q 2>&1 | grep -v "^$" | tail -40
