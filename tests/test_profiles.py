"""The committed evidence under profiles/ is coherent with itself (no GPU needed): the bench line's roofline figures for the dominant
kernel class agree with the rocprofv3 kernel statistics of the same command, and the PMC traffic per launch agrees with the library's
algorithmic bytes (scripts/check_traffic.py).  Guards against refreshing one file of a round and not the others."""
import csv
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROF = os.path.join(ROOT, "profiles")


def _latest_round():
    rounds = sorted({f[:3] for f in os.listdir(PROF) if f.startswith("r") and f[1:3].isdigit() and f.endswith("_bench_bf16.json")})
    return rounds[-1]


def test_bench_line_carries_the_contract_fields():
    line = json.load(open(os.path.join(PROF, _latest_round() + "_bench_bf16.json")))
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
                "data", "config", "roofline", "cpu_baseline"):
        assert key in line, key
    assert line["config"].get("workload") and "model" not in line["config"]
    r = line["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in r, key
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    c = line["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0 and c["sample"]
    # value = ray-samples of the whole job / time
    cfg = line["config"]
    assert abs(line["value"] - cfg["global_rays"] * cfg["samples_per_ray"] / (line["ms_per_step"] * 1e-3)) <= 1e-6 * line["value"]
    # every BASELINE config is a leg of the same run
    for leg in ("c2_parity_grade", "sampler_step_bf16", "c3_dtu_2048x128_bf16", "c4_rank_shape_512x128_bf16", "c5_hierarchical_64+64_fp16", "eval_chunk_2048_bf16"):
        assert leg in line["secondary"], leg


def test_kernel_stats_agree_with_the_bench_line():
    rnd = _latest_round()
    line = json.load(open(os.path.join(PROF, rnd + "_bench_bf16.json")))
    rows = list(csv.DictReader(open(os.path.join(PROF, rnd + "_bf16_kernel_stats.csv"))))
    r = line["roofline"]
    assert r["kernel"] == "layer_kernel"
    # the class of the library's event brackets = the streaming layer launches (layer_kernel_ws*, layer_kernel_wsdw*)
    sel = [x for x in rows if "layer_kernel_ws" in x["Name"]]
    assert sel
    calls = sum(int(x["Calls"]) for x in sel)
    avg_us = sum(float(x["TotalDurationNs"]) for x in sel) / calls / 1e3
    # HIP events around eager launches vs the tracer's kernel durations: within 10 %
    assert abs(avg_us / r["avg_launch_us"] - 1.0) <= 0.10, (avg_us, r["avg_launch_us"])
    # the traced kernel time per step brackets the step time of the line (the tracer adds ~2-3 us per launch)
    steps = 16                                   # 3 warm-up + capture + 10 timed + 2 eager steps of the traced command
    per_step_ms = sum(float(x["TotalDurationNs"]) for x in rows) / steps / 1e6
    assert 0.9 * line["ms_per_step"] <= per_step_ms <= 1.15 * line["ms_per_step"], (per_step_ms, line["ms_per_step"])
    # achieved = algorithmic flops (or bytes: the roof the class's byte mix puts it under) per launch / average launch duration
    tflops = r["flop_per_launch"] / (r["avg_launch_us"] * 1e-6) / 1e12
    if r["unit"] == "GB/s":                      # round 6 on: `bound` follows flop_per_byte against the ridge, both fractions carried
        assert r["bound"] == "hbm" and r["flop_per_byte"] < r["ridge_flop_per_byte"]
        assert abs(r["achieved"] - r["hbm"]["bytes_per_launch"] / (r["avg_launch_us"] * 1e-6) / 1e9) <= 1e-6 * r["achieved"]
        assert abs(r["frac_mfma"] - tflops / 2500.0) <= 1e-6 and abs(r["frac_hbm"] - r["frac"]) <= 1e-12
        assert 0.0 < r["step_frac_of_mfma_peak"] < r["frac_mfma"] * 1.5
        u = r["mfma_util_from_counters"]
        # counter-derived utilisation of the dominant class = its algorithmic flop fraction up to the padded work and the counter pass's slower launches
        assert u is not None and abs(u["layer_kernel"] / r["frac_mfma"] - 1.0) <= 0.15, (u, r["frac_mfma"])
    else:
        assert abs(r["achieved"] - tflops) <= 1e-6 * r["achieved"]


def test_pmc_traffic_agrees_with_the_algorithmic_bytes():
    rnd = _latest_round()
    out = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "check_traffic.py"), os.path.join(PROF, rnd + "_bench_bf16.json"), "bf16"],
                         capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    line = json.load(open(os.path.join(PROF, rnd + "_bench_bf16.json")))
    tj = json.load(open(os.path.join(PROF, "traffic.json")))["bf16"]
    assert abs(line["roofline"]["traffic"] / tj["layer_kernel"]["hbm_bytes_per_launch"] - 1.0) <= 0.02      # the line quotes THIS traffic.json
