#!/bin/bash
# kernel time of the fused primal (save mode and values mode) for every library in abl_libs/:   bash scripts/probes/abl_time.sh [GEN]
R=$PWD; G=${1:-3}; O=$R/gpurun_out/abl; rm -rf $O; mkdir -p $O; cd /tmp; export TMPDIR=/tmp PYTHONPATH=$R
for lib in $R/abl_libs/libneat_*.so; do
  n=$(basename $lib .so)
  for mode in save values; do
    NEAT_LIB=$lib rocprofv3 --kernel-trace --stats --output-format csv -d $O/$n.$mode -- python $R/scripts/probes/probe_fused_pmc.py $G $mode > $O/$n.$mode.log 2>&1
    f=$(find $O/$n.$mode -name "*kernel_stats.csv" | head -1)
    echo "$n $mode: $(grep -i 'fused' $f | awk -F'","' '{printf "%s calls=%s avg_us=%.1f ; ", substr($1,2,60), $2, $4/1000}')"
  done
done
