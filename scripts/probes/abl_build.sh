#!/bin/bash
# Probe builds of libneat_hip.so with extra -D flags:  scripts/probes/abl_build.sh NAME -DNEAT_F6_ABLATE=9 ...   -> abl_libs/libneat_NAME.so
# The flags go to ONE translation unit: NAME starting with "f": the fused chains' unit (primary build); "x": the f16 TWIN of the fused
# chains' unit (the split-precision chains of kernels_x3.hpp run there); otherwise neat_api.hip (primary).  The other units are
# compiled once without flags and cached in abl_libs/ (delete abl_libs/*.o after editing sources).
set -e
R=$(cd "$(dirname "$0")/../.." && pwd); name=$1; shift
mkdir -p $R/abl_libs; cd $R/neat_amd/csrc
[ -f f16_symbols.h ] || bash build.sh > /dev/null 2>&1
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -I."
C=$R/abl_libs
[ -f $C/neat_api.o ] || /opt/rocm/bin/hipcc $F -c neat_api.hip -o $C/neat_api.o 2>/dev/null &
[ -f $C/neat_api_f16.o ] || /opt/rocm/bin/hipcc $F -DNEAT_HALF=1 -c neat_api.hip -o $C/neat_api_f16.o 2>/dev/null &
[ -f $C/neat_fused.o ] || /opt/rocm/bin/hipcc $F -mllvm -amdgpu-mfma-vgpr-form=1 -fno-slp-vectorize -c neat_fused.hip -o $C/neat_fused.o 2>/dev/null &
[ -f $C/neat_fused_f16.o ] || /opt/rocm/bin/hipcc $F -DNEAT_HALF=1 -mllvm -amdgpu-mfma-vgpr-form=1 -fno-slp-vectorize -c neat_fused.hip -o $C/neat_fused_f16.o 2>/dev/null &
wait
api=$C/neat_api.o; fused=$C/neat_fused.o; fused16=$C/neat_fused_f16.o
if [[ $name == x* ]]; then
  fused16=$C/neat_fused_f16_$name.o
  /opt/rocm/bin/hipcc $F -DNEAT_HALF=1 -mllvm -amdgpu-mfma-vgpr-form=1 -fno-slp-vectorize "$@" -c neat_fused.hip -o $fused16 2>/dev/null
elif [[ $name == f* ]]; then
  fused=$C/neat_fused_$name.o
  /opt/rocm/bin/hipcc $F -mllvm -amdgpu-mfma-vgpr-form=1 -fno-slp-vectorize "$@" -c neat_fused.hip -o $fused 2>/dev/null
else
  api=$C/neat_api_$name.o
  /opt/rocm/bin/hipcc $F "$@" -c neat_api.hip -o $api 2>/dev/null
fi
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared $api $C/neat_api_f16.o $fused $fused16 -o $C/libneat_$name.so
if [[ $name == x* ]]; then rm -f $fused16; elif [[ $name == f* ]]; then rm -f $fused; else rm -f $api; fi
echo built abl_libs/libneat_$name.so
