# A/B of the compile-time variants of csrc/gconv3.hip on the dominant launches (one box, interleaved): the one-barrier form
# (-DHPL_PP=0), the single-barrier ping-pong (-DHPL_PP=2), 16 / 32 dummy VALU instructions per half-step (-DHPL_DUMMY_VALU=n).
# Output: gpurun_out/variants_ab.txt
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
cd hplflownet_amd/csrc
build() { /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -ffp-contract=off $2 -c gconv3.hip -o /tmp/g3_$1.o && \
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libhplbcl_$1.so index_ops.o row_order.o splat_slice.o gconv.o /tmp/g3_$1.o wgrad3.o lattice.o executor.o lattice_builder.o; }
build pp0 -DHPL_PP=0 & build pp2 -DHPL_PP=2 & build dv16 -DHPL_DUMMY_VALU=16 & build dv32 -DHPL_DUMMY_VALU=32 & wait
cd ../..
export ROUNDS=3 CASES="bcn1_ g0,bcn2_ g0,dense 25841"
for V in default pp0 pp2 dv16 dv32 default pp0 pp2 dv16 dv32; do
  if [ $V = default ]; then unset HPL_LIB; else export HPL_LIB=$PWD/hplflownet_amd/libhplbcl_$V.so; fi
  python tools/bench_split3.py 2>&1 | grep "split3 " | grep -v dgrad | sed "s/^/$V  /" | cut -c1-170
done | tee gpurun_out/variants_ab.txt
