"""From a rocprofv3 kernel_trace.csv: busy fraction of the GPU over the steady-state part of a bench run (gaps between kernels)."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
# steady state = the last 40 % of the trace
n = len(ev); part = ev[int(0.6 * n):]
span = part[-1][1] - part[0][0]
busy = 0; cur_s, cur_e = part[0][0], part[0][1]
gaps = []
for s, e, name in part[1:]:
    if s > cur_e:
        busy += cur_e - cur_s; gaps.append((s - cur_e, name)); cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
print(f"kernels {len(part)}, span {span/1e6:.3f} ms, busy {busy/1e6:.3f} ms ({100*busy/span:.1f} %), gaps {len(gaps)}, total gap {sum(g for g,_ in gaps)/1e6:.3f} ms")
gaps.sort(reverse=True)
for g, name in gaps[:12]:
    print(f"  gap {g/1e3:8.1f} us before {name[:80]}")
import collections
c = collections.Counter()
for g, name in gaps: c[min(int(g / 1000), 10)] += 1
print("gap histogram (us bucket -> count):", sorted(c.items()))
