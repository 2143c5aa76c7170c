# diagnostic: what each part of the split-operand kernel's half-step costs (builds libhplbcl_ablN.so on the box: results wrong, timing only)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
cd hplflownet_amd/csrc
for N in 1 2 3 4 5 6; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -ffp-contract=off -DHPL_ABLATE=$N -c gconv3.hip -o /tmp/g3_$N.o &
done
wait
for N in 1 2 3 4 5 6; do /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libhplbcl_abl$N.so index_ops.o row_order.o splat_slice.o gconv.o /tmp/g3_$N.o wgrad3.o lattice.o executor.o lattice_builder.o; done
cd ../..
export ROUNDS=3 CASES="bcn1_ g0,dense 25841"
python tools/bench_split3.py 2>&1 | grep "split3 " | sed 's/^/full  /' | cut -c1-170
for N in 1 2 3 4 5 6; do HPL_LIB=$PWD/hplflownet_amd/libhplbcl_abl$N.so python tools/bench_split3.py 2>&1 | grep "split3 " | sed "s/^/abl$N  /" | cut -c1-170; done
python tools/bench_split3.py 2>&1 | grep "split3 " | sed 's/^/full  /' | cut -c1-170
