cd $GRAFT_REPO_ROOT
for T in "" 128x128 128x128w16 128x64w8 64x128w8 64x64; do
HPL_TILE=$T HPL_SPLIT_MID=${MID:-1} python tools/bench_dense.py 2>&1 | tail -1
done
HPL_SPLIT_MID=0 python tools/bench_dense.py 2>&1 | tail -1
