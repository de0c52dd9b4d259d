#!/bin/bash
# per-kernel effect of the non-temporal fetch bits (tuning key 11) on the streaming layer kernels:  bash scripts/nt_probe.sh "15 11 3 0"
R=$PWD; O=$R/gpurun_out/ntp; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp PYTHONPATH=$R; cd $R
for v in ${1:-15 11 3 0}; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/v$v -- python $R/scripts/bench_tune.py 11=$v -- --steps 4 --warmup 2 --no-cpu-baseline --no-prof --no-graph > $O/v$v.log 2>&1
  f=$(find $O/v$v -name "*kernel_stats.csv" | head -1)
  python - "$f" "$v" <<'PY'
import csv, sys
out = []
for r in csv.DictReader(open(sys.argv[1])):
    for k in ("ws<4, 16", "ws<2, 16", "ws<2, 20", "ws<5, 16", "ws<6, 16", "ws<7, 16", "wgrad_kernel_h3"):
        if k in r["Name"]:
            out.append("%s max %.1f" % (k, float(r["MaxNs"]) / 1e3))
print("key11 =", sys.argv[2], " | ".join(sorted(out)))
PY
done
