#!/bin/bash
# Probe builds of libneat_hip.so with extra -D flags:  scripts/abl_build.sh NAME -DNEAT_F6_ABLATE=9 ...   -> abl_libs/libneat_NAME.so
# (flags that only touch the fused chains: NAME starting with "f" reuses the cached neat_api.o)
set -e
R=$(cd "$(dirname "$0")/.." && pwd); name=$1; shift
mkdir -p $R/abl_libs; cd $R/neat_amd/csrc
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -I."
api=$R/abl_libs/neat_api_$name.o
if [[ $name == f* ]]; then
  api=$R/abl_libs/neat_api.o
  [ -f $api ] || /opt/rocm/bin/hipcc $F -c neat_api.hip -o $api 2>/dev/null
else
  /opt/rocm/bin/hipcc $F "$@" -c neat_api.hip -o $api 2>/dev/null
fi
/opt/rocm/bin/hipcc $F -mllvm -amdgpu-mfma-vgpr-form=1 "$@" -c neat_fused.hip -o $R/abl_libs/neat_fused_$name.o 2>/dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared $api $R/abl_libs/neat_fused_$name.o -o $R/abl_libs/libneat_$name.so
rm -f $R/abl_libs/neat_fused_$name.o $R/abl_libs/neat_api_$name.o
echo built abl_libs/libneat_$name.so
