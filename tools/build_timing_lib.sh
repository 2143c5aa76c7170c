#!/bin/bash
# diagnostic build of the library with per-wave cycle stamps in k_gconv (tools/tile_timing.py)
set -e
cd "$(dirname "$0")/../hplflownet_amd/csrc"
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -ffp-contract=off -Wno-unused-function"
/opt/rocm/bin/hipcc $F -DHPL_TIMING $HPL_EXTRA_DEFS -c gconv.hip -o /tmp/gconv_timing.o
for f in index_ops splat_slice lattice executor lattice_builder; do [ -f $f.o ] || /opt/rocm/bin/hipcc $F -c $f.hip -o $f.o; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../${HPL_TIMING_LIB:-libhplbcl_timing.so} /tmp/gconv_timing.o index_ops.o splat_slice.o lattice.o executor.o lattice_builder.o
echo built ../${HPL_TIMING_LIB:-libhplbcl_timing.so}
