cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python tools/mfma_probe3.py 2>&1 | grep -v amdgpu | tee gpurun_out/r02f_mfma_probe.txt
