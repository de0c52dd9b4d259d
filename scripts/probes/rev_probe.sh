#!/bin/bash
# does the adjoint (REV) layer time depend on which fused primal kernel ran before it?   bash scripts/probes/rev_probe.sh
R=$PWD; O=$R/gpurun_out/revp; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp PYTHONPATH=$R; cd $R
for g in 1 3; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/g$g -- python $R/scripts/bench_tune.py 4=$g -- --steps 4 --warmup 2 --no-cpu-baseline --no-prof --no-graph > $O/g$g.log 2>&1
  f=$(find $O/g$g -name "*kernel_stats.csv" | head -1)
  python - "$f" "$g" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if any(k in r["Name"] for k in ("layer_kernel_ws<4, 16", "sdf_fused", "adjoint_seed", "layer_kernel_ws<0, 16, true")):
        print("gen", sys.argv[2], r["Name"][:60], r["Calls"], "avg %.1f max %.1f" % (float(r["AverageNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
done
