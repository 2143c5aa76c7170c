# diagnostic: per-workgroup residence of the dominant launches (builds libhplbcl_tp.so on the box with -DHPL_PHASE_PROBE=2)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
cd hplflownet_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -ffp-contract=off -DHPL_PHASE_PROBE=2 $HPL_EXTRA_DEFS -c gconv3.hip -o /tmp/g3_tp.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libhplbcl_tp.so index_ops.o row_order.o splat_slice.o gconv.o /tmp/g3_tp.o wgrad3.o lattice.o lattice_fused.o executor.o lattice_builder.o
cd ../..
HPL_LIB=$PWD/hplflownet_amd/libhplbcl_tp.so python tools/tile_probe.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/tile_probe${TP_TAG}.txt
