"""profiles/traffic.json from two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE, collected separately, each with
--kernel-trace only, as MI355X_MICROARCH.md prescribes).  FETCH_SIZE/WRITE_SIZE are in KiB; FETCH_SIZE is doubled
(gfx950 reports half the bytes of wide coalesced reads).  Per kernel class: mean over the full-size launches.

usage: python scripts/make_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> [precision]"""
import csv, json, os, sys

CLASSES = {"layer_kernel": ("layer_kernel_ws", "layer_kernel_h", "layer_kernel<"), "wgrad_kernel": ("wgrad_kernel", "dw_gather"),
           "sdf_fused_kernel": ("sdf_fused",), "sdf_adjoint_kernel": ("sdf_adjoint_w64",),
           "sdf_chain_x3_kernel": ("sdf_chain_x3",), "sdf_adjoint_x3_kernel": ("sdf_adjoint_x3",),
           "head_chain_kernel": ("head_chain_kernel", "head_bwd_chain_kernel", "head_chain_x3")}


def collect(path, counter):
    out = {}
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        name = r["Kernel_Name"]
        for cls, pats in CLASSES.items():
            if any(p in name for p in pats):
                # full-size launches only: persistent kernels have ~256 workgroups of 512 threads, the old ones >= 100k threads
                big = int(r["Grid_Size"]) >= 100000 or (("_ws" in name or "_h3" in name or "_w64" in name or "_x3" in name or "head_" in name) and int(r["Grid_Size"]) >= 100 * 512)
                if big:
                    out.setdefault(cls, {}).setdefault(name.split("(")[0], []).append(float(r["Counter_Value"]) * 1024.0)
    return out


def main():
    fetch, write = collect(sys.argv[1], "FETCH_SIZE"), collect(sys.argv[2], "WRITE_SIZE")
    prec = sys.argv[3] if len(sys.argv) > 3 else "bf16"
    res = {}
    for cls in CLASSES:
        f = [2.0 * v for vs in fetch.get(cls, {}).values() for v in vs]
        w = [v for vs in write.get(cls, {}).values() for v in vs]
        if not f or not w:
            continue
        fm, wm = sum(f) / len(f), sum(w) / len(w)
        per = {k: {"fetch_bytes_corrected": 2.0 * sum(v) / len(v), "write_bytes": sum(write[cls][k]) / len(write[cls][k]), "launches": len(v)}
               for k, v in fetch[cls].items() if k in write.get(cls, {})}
        res[cls] = {"hbm_bytes_per_launch": fm + wm, "fetch_bytes_corrected": fm, "write_bytes": wm, "launches_sampled": len(f),
                    "note": "full-size launches only; FETCH_SIZE x2 gfx950 correction applied", "per_kernel": per}
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles", "traffic.json")
    cur = json.load(open(path)) if os.path.exists(path) else {}
    cur[prec] = res
    json.dump(cur, open(path, "w"), indent=1)
    print(json.dumps({k: {kk: vv for kk, vv in v.items() if kk != "per_kernel"} for k, v in res.items()}, indent=1))


if __name__ == "__main__":
    main()
