# diagnostic: cycles per phase of the ping-pong half-step (builds libhplbcl_ph<N>.so on the box with -DHPL_PHASE_PROBE=1 and,
# for N > 0, -DHPL_ABLATE=N: 2 no split + store, 3 no loads and no stores, 7 no gathered loads, 8 no weight loads, ...)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
cd hplflownet_amd/csrc
for N in ${ABLATIONS:-0 2 7 8 3}; do
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -ffp-contract=off -DHPL_PHASE_PROBE=1 -DHPL_ABLATE=$N -c gconv3.hip -o /tmp/g3_ph$N.o &
done
wait
for N in ${ABLATIONS:-0 2 7 8 3}; do
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libhplbcl_ph$N.so index_ops.o row_order.o splat_slice.o gconv.o /tmp/g3_ph$N.o wgrad3.o lattice.o executor.o lattice_builder.o
done
cd ../..
for N in ${ABLATIONS:-0 2 7 8 3}; do echo "ablate $N"; HPL_LIB=$PWD/hplflownet_amd/libhplbcl_ph$N.so python tools/phase_probe.py 2>&1 | grep "wave row" | grep bcn1; done | tee gpurun_out/phase_probe.txt
