cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
for D in frustum surface; do
rocprofv3 --kernel-trace -d gpurun_out/ct_$D -o k -- python tools/chain_run.py $D > gpurun_out/ct_$D.log 2> gpurun_out/ct_$D.err
tail -1 gpurun_out/ct_$D.log
python tools/chain_trace.py $(ls gpurun_out/ct_$D/*/k_results.db gpurun_out/ct_$D/k_results.db 2>/dev/null | head -1) gpurun_out/r02_chain_$D.txt | head -60
rm -rf gpurun_out/ct_$D
python tools/chain_run.py $D | tail -1
done
