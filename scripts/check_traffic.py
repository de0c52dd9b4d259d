"""Fails (exit 1) when the PMC traffic per launch of a kernel class (profiles/traffic.json, fresh from scripts/make_traffic.py) differs by
more than 10 % from the algorithmic bytes per launch the library accounts for the same class in the bench line (roofline.all_kernels):
a byte cut claimed by construction must show in the counters, and wasted re-reads must not hide.
The layer class (the roofline's dominant kernel) is held to +-10 %, the other classes to +-15 %.

usage: python scripts/check_traffic.py <bench line json> [precision]"""
import json, os, sys

line = json.load(open(sys.argv[1]))
prec = sys.argv[2] if len(sys.argv) > 2 else "bf16"
tj = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles", "traffic.json")))[prec]
bad = 0
for cls, k in line["roofline"]["all_kernels"].items():
    t = tj.get(cls) or tj.get({"sdf_fused_kernel": "sdf_chain_x3_kernel", "sdf_adjoint_kernel": "sdf_adjoint_x3_kernel"}.get(cls, ""), None)
    if not t:
        print(f"{cls}: no PMC entry")
        continue
    alg = k["bytes_per_launch"]
    if cls in ("sdf_fused_kernel", "sdf_adjoint_kernel"):
        # two launches per step, the main pass (133 120 points at C2) and the junction-sized one (1024 points): the bench line averages
        # them, the PMC figure is the full-size launch alone
        alg *= 2.0 * 133120.0 / 134144.0
    ratio = t["hbm_bytes_per_launch"] / alg
    lim = 0.10 if cls == "layer_kernel" else 0.15
    flag = "" if abs(ratio - 1.0) <= lim else "  <-- outside +-%d %%" % round(100 * lim)
    bad += bool(flag)
    print(f"{cls:20s} PMC {t['hbm_bytes_per_launch'] / 1e6:8.1f} MB / launch, algorithmic {alg / 1e6:8.1f} MB: ratio {ratio:.3f}{flag}")
sys.exit(1 if bad else 0)
