#!/bin/bash
# per-kernel times of the train step WITH the error-bound sampler: bash scripts/probes/sampler_stats.sh PRECISION   (GPU box)
R=$PWD; prec=${1:-fp16x3}
O=$R/gpurun_out/sampler_$prec; rm -rf $O; mkdir -p $O; cd /tmp; export TMPDIR=/tmp PYTHONPATH=$R
cd $R; rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python scripts/bench_workloads.py --precision $prec --only-sampler --steps 10 > $O/run.log 2>&1
f=$(find $O -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print("total kernel ms", tot / 1e6, "(39 steps: 3 x (3 warm-up + 10))")
for r in rows[:40]:
    print(f"{r['Name'][:84]:84s} calls/step={int(r['Calls'])/39:6.1f} avg={float(r['AverageNs'])/1e3:8.1f} us  per step={float(r['TotalDurationNs'])/39e3:8.1f} us")
PY
grep workload $O/run.log | cut -c1-400
