#!/bin/bash
# kernel sequence of one eager C2 train step (names + durations), from a rocprofv3 kernel trace:  bash scripts/probes/step_sequence.sh
R=$PWD; O=$R/gpurun_out/seq; rm -rf $O; mkdir -p $O; cd /tmp; export TMPDIR=/tmp PYTHONPATH=$R
rocprofv3 --kernel-trace --output-format csv -d $O/t -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-prof --no-graph > $O/run.log 2>&1
python - "$(find $O/t -name '*kernel_trace.csv' | head -1)" <<'PY'
import csv, re, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r['Start_Timestamp']))
adam = [i for i, r in enumerate(rows) if 'adam_flat' in r['Kernel_Name']]
seg = rows[adam[-2] + 1:adam[-1] + 1]
tot = 0.0
for i, r in enumerate(seg):
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    tot += d
    n = re.sub(r'void |at::native::|neat::|\(anonymous namespace\)::', '', r['Kernel_Name'])
    print("%3d %7.1f  %s" % (i, d, n[:110]))
print("kernels", len(seg), "busy us", round(tot, 1))
PY
