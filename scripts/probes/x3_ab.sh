# per-launch times of the split-precision forward chains for a list of builds:  bash scripts/probes/x3_ab.sh default xA xB ...
# (names = abl_libs/libneat_NAME.so from scripts/probes/abl_build.sh; "default" = the in-tree library)
R=$PWD; mkdir -p $R/gpurun_out/x3ab
for n in "$@"; do
  O=$R/gpurun_out/x3ab/$n; rm -rf $O; mkdir -p $O
  if [ $n = default ]; then unset NEAT_HIP_LIB; else export NEAT_HIP_LIB=$R/abl_libs/libneat_$n.so; fi
  (cd /tmp && TMPDIR=/tmp PYTHONPATH=$R rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python $R/scripts/probes/x3_time.py fp16x3 $X3_MODE > $O/log.txt 2>&1)
  f=$(find $O -name "*kernel_stats.csv" | head -1)
  echo "== $n: $(tail -1 $O/log.txt)"
  python - $f <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "x3_kernel" in r["Name"]:
        print("   %-62s calls %3s avg %8.1f max %8.1f us" % (r["Name"][15:77], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
  find $O -name "*kernel_trace*" -delete
done
