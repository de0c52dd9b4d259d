#!/bin/bash
# Build libneat_hip.so for gfx950 in-tree (travels to the GPU box with the snapshot).
# Two translation units: neat_api.hip (everything but the fused chains) and neat_fused.hip (fused chains; MFMA
# accumulators in the VGPR half, see fused_launch.hpp).
set -e
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -I."
$HIPCC $FLAGS -c neat_api.hip -o neat_api.o "$@" &
pid=$!
$HIPCC $FLAGS -mllvm -amdgpu-mfma-vgpr-form=1 -c neat_fused.hip -o neat_fused.o "$@"
wait $pid
$HIPCC --offload-arch=gfx950 -fPIC -shared neat_api.o neat_fused.o -o libneat_hip.so
rm -f neat_api.o neat_fused.o
echo "built $(pwd)/libneat_hip.so"
