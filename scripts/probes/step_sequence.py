"""From a rocprofv3 kernel_trace.csv of a bench run: the kernels of the LAST replayed step in launch order (name, duration, gap before)."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
# a step ends with adam_flat_kernel (eval: a forward ends with the last junction_gate / composite kernel)
marker = "adam_flat" if not (len(sys.argv) > 2 and sys.argv[2] == "eval") else "composite_fwd"
ends = [i for i, e in enumerate(ev) if marker in e[2]]
lo, hi = ends[-3] + 1, ends[-2] + 1
tot = 0
for i in range(lo, hi):
    s, e, n = ev[i]
    gap = s - ev[i - 1][1]
    tot += e - s
    print(f"{i - lo:4d} {(e - s) / 1e3:8.1f} us  gap {gap / 1e3:6.1f}  {n[:100]}")
print("launches", hi - lo, "kernel us", tot / 1e3, "span us", (ev[hi - 1][1] - ev[lo - 1][1]) / 1e3)
