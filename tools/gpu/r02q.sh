cd $GRAFT_REPO_ROOT
for t in "" 128x128w16; do echo "HPL_TILE=$t"; HPL_TILE=$t python tools/bench_groups.py 2>&1 | grep -E "groups=(1|2) "; done
HPL_TILE=128x128w16 HPL_LIB=$PWD/hplflownet_amd/libhplbcl_timing.so python tools/tile_timing.py 2>&1 | grep -E "blur|wall span|residency|cycles per slice|share of" | head -12
