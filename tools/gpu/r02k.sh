cd $GRAFT_REPO_ROOT
export HPL_LIB=$PWD/hplflownet_amd/libhplbcl_timing.so
python tools/pers_timing.py 2>&1 | grep -v amdgpu
