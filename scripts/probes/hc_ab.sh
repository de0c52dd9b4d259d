# per-launch times of the fused head chains (kernels_heads.hpp; HC_PAT=substring: other kernels) for a list of builds:  bash scripts/probes/hc_ab.sh default fA fB ...
# (names = abl_libs/libneat_NAME.so from scripts/probes/abl_build.sh, "f..." = flags to the primary fused unit; "default" = the in-tree library)
R=$PWD; mkdir -p $R/gpurun_out/hcab
for n in "$@"; do
  O=$R/gpurun_out/hcab/$n; rm -rf $O; mkdir -p $O
  if [ $n = default ]; then unset NEAT_HIP_LIB; else export NEAT_HIP_LIB=$R/abl_libs/libneat_$n.so; fi
  (cd /tmp && TMPDIR=/tmp PYTHONPATH=$R rocprofv3 --kernel-trace --stats --output-format csv -d $O -- timeout 60 python $R/scripts/probes/hc_time.py ${HC_PREC:-bf16} ${HC_S:-128} $HC_TUNE > $O/log.txt 2>&1)
  f=$(find $O -name "*kernel_stats.csv" | head -1)
  echo "== $n: $(tail -1 $O/log.txt)"
  python - $f "${HC_PAT:-head_}" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if sys.argv[2] in r["Name"] and ("chain" in r["Name"] or sys.argv[2] != "head_"):
        print("   %-52s calls %3s avg %8.1f min %8.1f us" % (r["Name"][11:63], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3))
PY
  find $O -name "*kernel_trace*" -delete
done
