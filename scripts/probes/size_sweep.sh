#!/bin/bash
# per-kernel averages of the train step at several batch sizes (how do the streaming kernels scale with P?):  bash scripts/probes/size_sweep.sh "256 512 1024 2048"
R=$PWD; O=$R/gpurun_out/sweep; rm -rf $O; mkdir -p $O; cd /tmp; export TMPDIR=/tmp PYTHONPATH=$R
for n in ${1:-256 512 1024 2048}; do
  NEAT_BENCH_RAYS=$n rocprofv3 --kernel-trace --stats --output-format csv -d $O/r$n -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-prof --no-graph > $O/r$n.log 2>&1
  f=$(find $O/r$n -name "*kernel_stats.csv" | head -1)
  python - "$f" "$n" <<'PY'
import csv, sys
f, n = sys.argv[1], int(sys.argv[2])
rows = list(csv.DictReader(open(f)))
for pat in ("layer_kernel_ws<4, 16", "layer_kernel_ws<5, 16", "layer_kernel_ws<6, 16", "layer_kernel_ws<2, 16", "layer_kernel_ws<7, 16", "wgrad_kernel_h3", "sdf_fused"):
    for r in rows:
        if pat in r["Name"]:
            print("rays %5d  %-28s calls %4s  avg %8.1f us  max %8.1f us  -> max per 1024 rays %7.1f" % (n, pat, r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MaxNs"]) / 1e3, float(r["MaxNs"]) / 1e3 * 1024 / n))
PY
done
