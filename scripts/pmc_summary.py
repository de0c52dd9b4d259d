"""Per-kernel averages of rocprofv3 --pmc counter_collection.csv files:  python scripts/pmc_summary.py DIR [name-substring ...]"""
import collections, csv, glob, sys
root, pats = sys.argv[1], sys.argv[2:]
for f in sorted(glob.glob(root + "/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    disp = collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:70]
        if pats and not any(p in k for p in pats):
            continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        disp[k].add(r["Dispatch_Id"])
    print("==", f.replace(root, ""))
    for k, v in sorted(acc.items()):
        print("  ", k, "launches", len(disp[k]))
        print("      " + "  ".join("%s=%.4g" % (c, x / len(disp[k])) for c, x in sorted(v.items())))
