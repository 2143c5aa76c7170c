# diagnostic: cycles per phase of the ping-pong half-step of k_gconv3w (builds libhplbcl_ph.so on the box with -DHPL_PHASE_PROBE=1)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
cd hplflownet_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -ffp-contract=off -DHPL_PHASE_PROBE=1 -c gconv3.hip -o /tmp/g3_ph.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libhplbcl_ph.so index_ops.o row_order.o splat_slice.o gconv.o /tmp/g3_ph.o wgrad3.o lattice.o lattice_fused.o executor.o lattice_builder.o
cd ../..
HPL_LIB=$PWD/hplflownet_amd/libhplbcl_ph.so python tools/phase_probe.py 2>&1 | grep "wave row" | tee gpurun_out/phase_probe.txt
