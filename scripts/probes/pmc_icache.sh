# instruction-cache counters of the fused primal kernels.  usage: bash scripts/probes/pmc_icache.sh "GENS" "MODES"
GENS=${1:-"1 2 3"}; MODES=${2:-values}
R=$PWD; O=$R/gpurun_out/pmc_ic; rm -rf $O; mkdir -p $O; cd /tmp; export TMPDIR=/tmp PYTHONPATH=$R
for g in $GENS; do for m in $MODES; do
rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_BUSY_CYCLES --output-format csv -d $O/i$g$m -- python $R/scripts/probes/probe_fused_pmc.py $g $m > $O/i$g$m.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_IFETCH_LEVEL SQC_ICACHE_BUSY_CYCLES SQC_TC_INST_REQ SQC_TC_STALL SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES --output-format csv -d $O/j$g$m -- python $R/scripts/probes/probe_fused_pmc.py $g $m > $O/j$g$m.log 2>&1
done; done
python - <<PY
import csv, glob, collections
for d in sorted(glob.glob("$O/*/")):
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(float)); disp = collections.defaultdict(set)
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:60]
            if "fused" not in k: continue
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); disp[k].add(r["Dispatch_Id"])
        for k, v in acc.items():
            print(d.split("/")[-2], k, "launches", len(disp[k]))
            for c, x in sorted(v.items()): print("    %-28s %.4g per launch" % (c, x / len(disp[k])))
PY
find $O -name "*kernel_trace*" -delete; find $O -name "*.csv" -size +2M -delete
