cd $GRAFT_REPO_ROOT
O=gpurun_out/r06q; mkdir -p $O
HPL_LIB=$PWD/hplflownet_amd/libhplbcl_probe.so python tools/tile_phase_probe.py 2>&1 | grep -v amdgpu.ids | tee $O/tile_phase_probe.txt
