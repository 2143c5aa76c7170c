cd $GRAFT_REPO_ROOT
for lib in libhplbcl_timing.so libhplbcl_timing_noprio.so libhplbcl_timing.so libhplbcl_timing_noprio.so; do echo "== $lib"; HPL_LIB=$PWD/hplflownet_amd/$lib python tools/tile_timing.py 2>&1 | grep -E "blur|per workgroup|cycles per slice|share of" | head -8; done
