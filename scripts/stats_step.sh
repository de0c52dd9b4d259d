#!/bin/bash
# per-kernel times of the bench step:  bash scripts/stats_step.sh PRECISION [tuning key=value ...]   (on the GPU box)
R=$PWD; prec=${1:-bf16}; shift
O=$R/gpurun_out/stats_$prec; rm -rf $O; mkdir -p $O; cd /tmp; export TMPDIR=/tmp PYTHONPATH=$R
rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python $R/scripts/bench_tune.py "$@" -- --precision $prec --steps 10 --warmup 3 --no-secondary --no-cpu-baseline > $O/run.log 2>&1
f=$(find $O -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print("total kernel ms", tot / 1e6)
for r in rows[:int(28)]:
    print(f"{r['Name'][:80]:80s} calls={r['Calls']:>5s} avg={float(r['AverageNs'])/1e3:8.1f} us  {float(r['Percentage']):5.1f} %")
PY
grep '^{"metric"' $O/run.log | python -c "import sys, json; d = json.loads(sys.stdin.read()); print('ms_per_step', d['ms_per_step'], 'value', d['value'])"
