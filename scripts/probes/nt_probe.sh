#!/bin/bash
# per-kernel effect of the non-temporal fetch bits (tuning key 11) on the full-size streaming layer launches:  bash scripts/probes/nt_probe.sh "15 14 11 10 0"
R=$PWD; O=$R/gpurun_out/ntp; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp PYTHONPATH=$R; cd $R
for v in ${1:-15 14 11 10 0}; do
  rocprofv3 --kernel-trace --output-format csv -d $O/v$v -- python $R/scripts/bench_tune.py 11=$v -- --steps 4 --warmup 2 --no-cpu-baseline --no-prof --no-graph > $O/v$v.log 2>&1
  f=$(find $O/v$v -name "*kernel_trace.csv" | head -1)
  python - "$f" "$v" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    for k in ("ws<4, 16", "ws<2, 16", "ws<5, 16", "ws<6, 16", "ws<7, 16", "ws<0, 16, true"):
        if k in n:
            d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
            if d > 15.0:
                acc[k].append(d)
print("key11 = %2s  " % sys.argv[2] + " | ".join("%s %.1f (%d)" % (k, sum(v) / len(v), len(v)) for k, v in sorted(acc.items())))
PY
done
