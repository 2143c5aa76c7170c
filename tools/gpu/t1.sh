cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/tests.log 2>&1; tail -3 gpurun_out/tests.log
rocprofv3 --kernel-trace -d gpurun_out/ct_s -o k -- python tools/chain_run.py surface > /dev/null 2>&1
python tools/chain_trace.py $(ls gpurun_out/ct_s/*/k_results.db gpurun_out/ct_s/k_results.db 2>/dev/null | head -1) gpurun_out/r02_chain_surface_split.txt | head -24
rm -rf gpurun_out/ct_s
