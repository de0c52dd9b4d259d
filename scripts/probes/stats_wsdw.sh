# per-kernel averages of the in-kernel-gradient launches over a traced bench run:  bash scripts/probes/stats_wsdw.sh [bench args]
R=$PWD; O=$R/gpurun_out/stw; rm -rf $O; mkdir -p $O; cd /tmp; export TMPDIR=/tmp PYTHONPATH=$R
rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-prof "$@" > $O/run.log 2>&1
f=$(find $O -name "*kernel_stats.csv" | head -1)
grep "wsdw\|dw_gather" $f | cut -d, -f1,2,3,4
python - <<PY
import csv
rows = list(csv.DictReader(open("$f")))
print("wsdw total per step us", sum(float(r["TotalDurationNs"]) for r in rows if "wsdw" in r["Name"]) / 16 / 1e3, "all kernels per step ms", sum(float(r["TotalDurationNs"]) for r in rows) / 16 / 1e6)
PY
grep '"metric"' $O/run.log | python -c "import sys,json; print('ms_per_step', json.loads(sys.stdin.read())['ms_per_step'])"
find $O -name "*kernel_trace*" -delete
