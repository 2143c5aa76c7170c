cd $GRAFT_REPO_ROOT
for T in "" 64x128 64x128w8 128x128w4; do echo "HPL_TILE=$T"; HPL_TILE=$T python tools/bench_groups.py 2>&1 | grep -i "groups=2\|groups=1\|G=2\|G=1" | head -6; done
