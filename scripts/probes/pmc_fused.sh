# PMC passes over the fused primal kernels.  usage: bash scripts/probes/pmc_fused.sh "GENS" "MODES"   (default "2" "values save")
GENS=${1:-2}; MODES=${2:-values save}
R=$PWD; O=$R/gpurun_out/pmc_fused; rm -rf $O; mkdir -p $O; cd /tmp; export TMPDIR=/tmp PYTHONPATH=$R
for g in $GENS; do for m in $MODES; do
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY --output-format csv -d $O/a$g$m -- python $R/scripts/probes/probe_fused_pmc.py $g $m > $O/a$g$m.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_INSTS_SALU --output-format csv -d $O/b$g$m -- python $R/scripts/probes/probe_fused_pmc.py $g $m > $O/b$g$m.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VMEM SQ_INSTS_SMEM GRBM_GUI_ACTIVE --output-format csv -d $O/c$g$m -- python $R/scripts/probes/probe_fused_pmc.py $g $m > $O/c$g$m.log 2>&1
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
find $O -name "*kernel_trace*" -delete; find $O -name "*.csv" -size +2M -delete; du -sh $O
