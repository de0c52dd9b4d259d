# what sits in the gap of the sampler step's graph?  kernel trace + memory-copy trace, merged by time
R=$PWD; O=$R/gpurun_out/sgap; rm -rf $O; mkdir -p $O; cd /tmp; export TMPDIR=/tmp PYTHONPATH=$R
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O -- python $R/scripts/probes/workload_step.py sampler bf16 6 > $O/run.log 2>&1
python - $O <<'PY'
import csv, glob, sys
O = sys.argv[1]
kt = glob.glob(O + "/**/*kernel_trace.csv", recursive=True)[0]
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:70]) for r in csv.DictReader(open(kt))]
mc = glob.glob(O + "/**/*memory_copy_trace.csv", recursive=True)
if mc:
    for r in csv.DictReader(open(mc[0])):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "MEMCPY " + r.get("Direction", "") + " " + r.get("Bytes", r.get("Size", ""))))
ev.sort()
ends = [i for i, e in enumerate(ev) if "adam_flat" in e[2]]
lo, hi = ends[-3] + 1, ends[-2] + 1
out = open(O + "/merged.txt", "w")
for i in range(lo, hi):
    s, e, n = ev[i]
    out.write(f"{i - lo:4d} {(e - s) / 1e3:8.1f} us gap {(s - ev[i - 1][1]) / 1e3:7.1f}  {n}\n")
PY
find $O -name "*_trace.csv" -delete
