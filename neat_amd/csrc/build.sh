#!/bin/bash
# Build libneat_hip.so for gfx950 in-tree (travels to the GPU box with the snapshot).
set -e
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
$HIPCC --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -I. neat_api.hip -o libneat_hip.so "$@"
echo "built $(pwd)/libneat_hip.so"
