cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export HPL_LIB=$PWD/hplflownet_amd/libhplbcl_timing.so
for v in "ABL=0" "ABL=1" "ABL=2"; do echo "=== $v"; env $v python tools/tile_timing.py 2>&1 | grep -v amdgpu | grep -E "blur|dense|wall span|mean residency|cycles per slice|share of"; done > gpurun_out/r02g_ablate.txt
cat gpurun_out/r02g_ablate.txt
