#!/bin/bash
# durations of the fp16x3 SDF chains on 133 120 points (get_outputs, no training step): bash scripts/probes/x3_kernel_times.sh   (GPU box)
R=$PWD; O=$R/gpurun_out/x3k; rm -rf $O; mkdir -p $O; cd /tmp; export TMPDIR=/tmp PYTHONPATH=$R
rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python $R/scripts/probes/x3_timing.py 12 > $O/run.log 2>&1
f=$(find $O -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "x3" in r["Name"]:
        print(f"{r['Name'][:62]:62s} calls {r['Calls']:>3s} avg {float(r['AverageNs'])/1e3:7.1f} us  min {float(r['MinNs'])/1e3:7.1f}  max {float(r['MaxNs'])/1e3:7.1f}")
PY
