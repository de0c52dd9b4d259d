#!/bin/bash
# per-kernel effect of the persistent grid size (tuning key 3) on the streaming layer launches:  bash scripts/probes/grid_probe.sh "256 512 128"
R=$PWD; O=$R/gpurun_out/gridp; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp PYTHONPATH=$R; cd $R
for v in ${1:-256 512}; do
  rocprofv3 --kernel-trace --output-format csv -d $O/v$v -- python $R/scripts/bench_tune.py 3=$v -- --steps 4 --warmup 2 --no-cpu-baseline --no-prof --no-graph > $O/v$v.log 2>&1
  f=$(find $O/v$v -name "*kernel_trace.csv" | head -1)
  python - "$f" "$v" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    for k in ("ws<0, 16, true", "ws<3, 16, true", "ws<2, 16", "ws<7, 16", "ws<5, 16", "ws<6, 16", "ws<0, 16, false"):
        if k in n:
            d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
            if d > 12.0:
                acc[k].append(d)
print("grid = %4s  " % sys.argv[2] + " | ".join("%s %.1f (%d)" % (k, sum(v) / len(v), len(v)) for k, v in sorted(acc.items())))
PY
done
