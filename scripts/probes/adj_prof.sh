#!/bin/bash
# kernel times of get_outputs with the fused / streamed adjoint chain:  bash scripts/probes/adj_prof.sh
R=$PWD; O=$R/gpurun_out/adjp; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp PYTHONPATH=$R; cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d $O/s -- python $R/scripts/probes/adj_ab.py > $O/run.log 2>&1
tail -6 $O/run.log
python - "$(find $O/s -name '*kernel_stats.csv' | head -1)" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if any(k in r["Name"] for k in ("adjoint", "ws<4, 16", "sdf_fused", "ws<0, 16, true", "export", "finalize", "fm_", "oct_to")):
        print("%-70s calls %4s avg %8.1f min %8.1f max %8.1f" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
