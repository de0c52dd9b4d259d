"""profiles/mfma_util.json: counter-derived MFMA utilisation per kernel and per launch class, from the SQ pass of scripts/pmc_sq.sh
(rocprofv3 --kernel-trace --pmc ... SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES ..., its own run, as MI355X_MICROARCH.md prescribes).

Units (MI355X_MICROARCH.md, "s_memtime tick vs SQ PMC units"): SQ_VALU_MFMA_BUSY_CYCLES counts shader cycles in which a SIMD's matrix
pipe is busy, summed over the chip's 256 CUs x 4 SIMDs (32 per v_mfma_f32_32x32x16_{bf16,f16}); the dense peak of 2.5 PFLOP/s is every
SIMD's pipe busy in every cycle at 2.4 GHz.  So
    mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (launch duration x 2.4e9 Hz x 1024 SIMDs)
with the duration of the SAME dispatch (Start/End timestamps of the counter pass: a launch under counter collection runs a few per
cent slower than in the timed bench, so this is a lower bound on the utilisation of an uninstrumented launch).  It differs from the
algorithmic flop fraction of the bench line by the padded work (217 -> 224 rows, 39 -> 48 inputs, ragged tiles).
`mfma_issue_share` = the same cycles over SQ_BUSY_CYCLES x SIMDs-per-counter-instance when that counter is present (kept raw).

usage: python scripts/make_mfma_util.py <sq counter_collection.csv> [precision]"""
import collections, csv, json, os, sys

CLOCK_HZ, SIMDS = 2.4e9, 256 * 4
CLASSES = {"layer_kernel": ("layer_kernel_ws", "layer_kernel_h", "layer_kernel<"), "wgrad_kernel": ("wgrad_kernel",),
           "sdf_fused_kernel": ("sdf_fused", "sdf_chain_x3"), "sdf_adjoint_kernel": ("sdf_adjoint_w64", "sdf_adjoint_x3"),
           "head_chain_kernel": ("head_chain_kernel", "head_bwd_chain_kernel", "head_chain_x3")}


def main():
    path, prec = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "bf16")
    disp = {}
    for r in csv.DictReader(open(path)):
        d = disp.setdefault(r["Dispatch_Id"], {"name": r["Kernel_Name"], "grid": int(r["Grid_Size"]),
                                                "ns": float(r["End_Timestamp"]) - float(r["Start_Timestamp"])})
        d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    per = collections.defaultdict(list)
    for d in disp.values():
        if "SQ_VALU_MFMA_BUSY_CYCLES" not in d or "neat" not in d["name"]:
            continue
        per[d["name"].split("(")[0].replace("void ", "")].append(d)
    kernels, classes = {}, {}
    for name, ds in sorted(per.items()):
        big = max(x["grid"] for x in ds)
        full = [x for x in ds if x["grid"] == big and x["SQ_VALU_MFMA_BUSY_CYCLES"] >= 0.5 * max(y["SQ_VALU_MFMA_BUSY_CYCLES"] for y in ds)]
        cyc = sum(x["SQ_VALU_MFMA_BUSY_CYCLES"] for x in full) / len(full)
        ns = sum(x["ns"] for x in full) / len(full)
        if cyc <= 0:
            continue
        kernels[name] = {"mfma_busy_cycles_per_launch": cyc, "launch_us_under_counters": ns / 1e3, "launches_sampled": len(full),
                         "mfma_util": cyc / (ns * 1e-9 * CLOCK_HZ * SIMDS)}
        if "SQ_BUSY_CYCLES" in full[0]:
            kernels[name]["sq_busy_cycles_per_launch"] = sum(x["SQ_BUSY_CYCLES"] for x in full) / len(full)
    for cls, pats in CLASSES.items():
        rows = [(k, v) for k, v in kernels.items() if any(p in k for p in pats)]
        if rows:
            cyc = sum(v["mfma_busy_cycles_per_launch"] * v["launches_sampled"] for _, v in rows)
            sec = sum(v["launch_us_under_counters"] * 1e-6 * v["launches_sampled"] for _, v in rows)
            n = sum(v["launches_sampled"] for _, v in rows)
            classes[cls] = {"mfma_util": cyc / (sec * CLOCK_HZ * SIMDS), "mfma_busy_cycles_per_launch": cyc / n,
                            "launch_us_under_counters": 1e6 * sec / n, "launches_sampled": n, "kernels": [k for k, _ in rows]}
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles", "mfma_util.json")
    cur = json.load(open(out)) if os.path.exists(out) else {}
    cur[prec] = {"formula": "SQ_VALU_MFMA_BUSY_CYCLES / (dispatch duration x 2.4e9 Hz x 1024 SIMDs), full-size launches, time-weighted per class",
                 "source": "rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES ... (scripts/pmc_sq.sh pass a)", "classes": classes, "kernels": kernels}
    json.dump(cur, open(out, "w"), indent=1)
    for cls, v in classes.items():
        print(f"{cls:20s} mfma_util {v['mfma_util']:.3f}   ({v['launch_us_under_counters']:.1f} us per launch under counters, {v['launches_sampled']} launches)")
    for k, v in kernels.items():
        print(f"   {k[:70]:70s} {v['mfma_util']:.3f}  {v['launch_us_under_counters']:7.1f} us")


if __name__ == "__main__":
    main()
