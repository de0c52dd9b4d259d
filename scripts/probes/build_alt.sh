#!/bin/bash
# alternate library build for A/Bs through NEAT_HIP_LIB (ab_lib.sh / stats_libs.sh):
#   FUSED="-fno-slp-vectorize" API="-DNEAT_X=1" bash scripts/probes/build_alt.sh NAME   ->   abl_libs/libneat_NAME.so
# FUSED / API = extra hipcc flags for neat_fused.hip / neat_api.hip (both twins of each)
set -e
R=$(cd "$(dirname "$0")/../.." && pwd); N=$1; T=/tmp/alt_$N
rm -rf $T; mkdir -p $T/neat_amd $R/abl_libs; cp -r $R/include $T/; cp -r $R/neat_amd/csrc $T/neat_amd/; cd $T/neat_amd/csrc
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -I."
H=/opt/rocm/bin/hipcc
$H $F $API -c neat_api.hip -o a.o & $H $F -DNEAT_HALF=1 $API -c neat_api.hip -o a16.o &
$H $F -mllvm -amdgpu-mfma-vgpr-form=1 -fno-slp-vectorize $FUSED -c neat_fused.hip -o f.o & $H $F -DNEAT_HALF=1 -mllvm -amdgpu-mfma-vgpr-form=1 -fno-slp-vectorize $FUSED -c neat_fused.hip -o f16.o &
wait
$H --offload-arch=gfx950 -fPIC -shared a.o a16.o f.o f16.o -o $R/abl_libs/libneat_$N.so
echo built $R/abl_libs/libneat_$N.so
