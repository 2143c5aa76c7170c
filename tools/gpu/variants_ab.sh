# A/B of compile-time variants of csrc/gconv3.hip on the dominant launches (one box, interleaved rounds).
# usage: VARIANTS="pp3:-DHPL_PP=3 pp0:-DHPL_PP=0" bash tools/gpu/variants_ab.sh     -> gpurun_out/variants_ab.txt
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
cd hplflownet_amd/csrc
OBJS="index_ops.o row_order.o splat_slice.o gconv.o wgrad3.o lattice.o lattice_fused.o executor.o lattice_builder.o"
build() { /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -ffp-contract=off $2 -c gconv3.hip -o /tmp/g3_$1.o && \
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libhplbcl_$1.so $OBJS /tmp/g3_$1.o; }
NAMES=""
for V in ${VARIANTS:-pp3:-DHPL_PP=3}; do build ${V%%:*} "${V#*:}" & NAMES="$NAMES ${V%%:*}"; done
wait
cd ../..
export ROUNDS=${ROUNDS:-3} CASES="${CASES:-bcn1_ g0,bcn1_ g1,bcn2_ g0,dense 25841}"
for R in 1 2; do for V in default $NAMES; do
  if [ $V = default ]; then unset HPL_LIB; else export HPL_LIB=$PWD/hplflownet_amd/libhplbcl_$V.so; fi
  python tools/bench_split3.py 2>&1 | grep "split3 " | grep -v dgrad | sed "s/^/$V  /" | cut -c1-170
done; done | tee gpurun_out/variants_ab.txt
if [ -n "$TEST_VARIANT" ]; then HPL_LIB=$PWD/hplflownet_amd/libhplbcl_$TEST_VARIANT.so timeout 600 python -m pytest tests/test_gpu_split3.py -x -q 2>&1 | tail -3; fi
