#!/bin/bash
# per-kernel averages of the C2 train step for every library in abl_libs/:   bash scripts/probes/abl_step.sh PATTERN [PATTERN ...]
R=$PWD; O=$R/gpurun_out/abl_step; rm -rf $O; mkdir -p $O; cd /tmp; export TMPDIR=/tmp PYTHONPATH=$R
for lib in $R/abl_libs/libneat_*.so; do
  n=$(basename $lib .so)
  NEAT_HIP_LIB=$lib rocprofv3 --kernel-trace --stats --output-format csv -d $O/$n -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-prof --no-graph > $O/$n.log 2>&1
  f=$(find $O/$n -name "*kernel_stats.csv" | head -1)
  python - "$f" "$n" "$@" <<'PY'
import csv, sys
f, n, pats = sys.argv[1], sys.argv[2], sys.argv[3:]
for r in csv.DictReader(open(f)):
    if any(p in r["Name"] for p in pats):
        print("%-14s %-62s calls %4s avg %8.1f us  min %8.1f" % (n, r["Name"][:62], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3))
PY
done
