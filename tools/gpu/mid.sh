cd $GRAFT_REPO_ROOT
for T in "" 64x128w8 128x64w8 64x64; do echo "HPL_TILE=$T"; HPL_TILE=$T SHAPES="bcn3_,bcn4_,pair bcn,corr1" REPS=20 python tools/bench_gconv.py 2>&1 | grep -v "^$" | grep "row_perm(mask)\|M=" | cut -c1-110; done
