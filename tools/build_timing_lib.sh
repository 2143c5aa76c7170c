#!/bin/bash
# diagnostic build of the library with per-wave cycle stamps in the fp32 kernel k_gconv (tools/tile_timing.py); the split-operand
# kernel has its own probes: tools/gpu/phase_probe.sh (cycles per phase), tools/gpu/tile_probe.sh (per-workgroup residence)
set -e
cd "$(dirname "$0")/../hplflownet_amd/csrc"
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -ffp-contract=off -Wno-unused-function"
/opt/rocm/bin/hipcc $F -DHPL_TIMING $HPL_EXTRA_DEFS -c gconv.hip -o /tmp/gconv_timing.o
OBJS="index_ops row_order splat_slice gconv3 wgrad3 lattice lattice_fused executor lattice_builder"
for f in $OBJS; do [ -f $f.o ] || /opt/rocm/bin/hipcc $F -c $f.hip -o $f.o; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../${HPL_TIMING_LIB:-libhplbcl_timing.so} /tmp/gconv_timing.o $(for f in $OBJS; do echo $f.o; done)
echo built ../${HPL_TIMING_LIB:-libhplbcl_timing.so}
