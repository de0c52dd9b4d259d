# per-kernel averages (rocprofv3 --stats) of the bench step for the default build and every abl_libs/libneat_*.so:  bash scripts/probes/stats_libs.sh [pattern] [precision]
PAT=${1:-head_}; P=${2:-bf16}
R=$PWD; cd /tmp; export TMPDIR=/tmp PYTHONPATH=$R
for lib in "" $R/abl_libs/libneat_*.so; do
  O=$R/gpurun_out/stl; rm -rf $O; mkdir -p $O
  NEAT_HIP_LIB=$lib rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python $R/bench.py --precision $P --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-prof > $O/run.log 2>&1
  f=$(find $O -name "*kernel_stats.csv" | head -1)
  echo "== ${lib:-default}"
  python - <<PY
import csv
rows = list(csv.DictReader(open("$f")))
print("   all kernels per step ms %.3f" % (sum(float(r["TotalDurationNs"]) for r in rows) / 16 / 1e6))
for r in rows:
    if any(p in r["Name"] for p in "$PAT".split("|")):
        print("   %-60s %6.1f us x %.2f" % (r["Name"][:60], float(r["AverageNs"]) / 1e3, int(r["Calls"]) / 16))
PY
done
