# per-kernel averages (rocprofv3 --stats) of the bench step with a tuning key at two values, alternating:  bash scripts/probes/stats_key.sh KEY [pattern] [precision] [passes] ["v0 v1"]
KEY=$1; PAT=${2:-wgrad|wreduce}; P=${3:-bf16}; N=${4:-2}; VALS=${5:-0 1}
R=$PWD; cd /tmp; export TMPDIR=/tmp PYTHONPATH=$R
for pass in $(seq $N); do for v in $VALS; do
  O=$R/gpurun_out/stk; rm -rf $O; mkdir -p $O
  NEAT_TUNING=$KEY=$v rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python $R/bench.py --precision $P --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-prof > $O/run.log 2>&1
  f=$(find $O -name "*kernel_stats.csv" | head -1)
  echo "== key $KEY = $v"
  python - <<PY
import csv
rows = list(csv.DictReader(open("$f")))
print("   all kernels per step ms %.3f" % (sum(float(r["TotalDurationNs"]) for r in rows) / 16 / 1e6))
for r in rows:
    if any(p in r["Name"] for p in "$PAT".split("|")):
        print("   %-60s %6.1f us x %.2f" % (r["Name"][:60], float(r["AverageNs"]) / 1e3, int(r["Calls"]) / 16))
PY
done; done
