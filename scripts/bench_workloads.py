"""Secondary workloads of SURVEY 8d (not the headline): (1) the full train step WITH the error-bound sampler (conf default
N_samples = 64 -> 98 samples per ray), (2) eval-mode forward in 2048-ray chunks as neat-final-parsing.py drives the model.
Prints one JSON line each.  (3) C3-style: 2048 rays x 128 given samples with the DTU conf switches (dbscan_enabled, 1024 global junctions), eager.
(4) C4's per-rank shape: 512 rays x 128 given samples with the DTU switches (one of eight data-parallel ranks; world size 1 here).
(5) C5-style: 1024 rays, hierarchical 64 coarse + 64 fine depths (HierarchicalSampler), abc model.
Prints one JSON line each.  usage: python scripts/bench_workloads.py [--precision bf16]"""
import argparse, json, sys, time
import torch
sys.path.insert(0, '.')
from neat_amd import synth
from neat_amd.train import Trainer, synthetic_batch

ap = argparse.ArgumentParser()
ap.add_argument("--precision", default="bf16")
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--only-sampler", action="store_true")
args = ap.parse_args()
dev = torch.device("cuda:0")
torch.manual_seed(42)
tr = Trainer(device=dev, state_dict={k: torch.tensor(v) for k, v in synth.synth_state_dict(42, "rough").items()})
tr.model.set_precision(args.precision)
_, inp, gt = synthetic_batch(42, 1024, dev)
def timed_steps(label):
    for _ in range(3):
        tr.step(inp, gt)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out, lo = tr.step(inp, gt)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    S = out["points"].shape[0] // 1024 if out["points"].dim() == 2 else out["points"].shape[1]
    print(json.dumps({"workload": "train step with ErrorBoundSampler, 1024 rays, conf-default sampler", "sampler": label,
                      "rounds": tr.model.ray_sampler.rounds_taken(), "ms_per_step": 1e3 * dt, "samples_per_ray": S,
                      "ray_samples_per_s": 1024 * S / dt, "rays_per_s": 1024 / dt, "precision": args.precision}), flush=True)


timed_steps("host decides (one sync per round), eager")
tr.model.ray_sampler.sync_free = True
timed_steps("device decides (no sync), eager")
graphed = tr.capture(inp, gt)
timed_steps("device decides (no sync), HIP graph" if graphed else f"capture failed: {tr.capture_error!r}")
tr._graphs.clear()
tr.model.ray_sampler.sync_free = False
if args.only_sampler:
    sys.exit(0)
tr.model.eval()
_, inp2, _ = synthetic_batch(43, 2048, dev)
with torch.no_grad():
    for _ in range(3):
        o = tr.model(inp2)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        o = tr.model(inp2)
    torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / args.steps
print(json.dumps({"workload": "eval forward (sampler + render + junction block), one 2048-ray chunk", "ms_per_chunk": 1e3 * dt,
                  "rays_per_s": 2048 / dt, "precision": args.precision}))

# ---- C3-style: DTU switches (dbscan_enabled -> device DBSCAN, use_median off, 1024 junction latents), 2048 x 128 given samples
import copy
from neat_amd import networks
conf = copy.deepcopy(synth.ABC_NEAT_A_MODEL_CONF)
conf.update(dbscan_enabled=True, use_median=False)
conf["global_junctions"] = dict(conf["global_junctions"], num_junctions=1024)
sd = synth.synth_state_dict(42, "rough", num_junctions=1024)
for dbscan in (True, False):
    conf["dbscan_enabled"] = dbscan
    tr3 = Trainer(model_conf=conf, device=dev, state_dict={k: torch.tensor(v) for k, v in sd.items()})
    tr3.model.set_precision(args.precision)
    _, inp3, gt3 = synthetic_batch(42, 2048, dev)
    tr3.model.z_vals_override = torch.tensor(synth.synth_z_vals(42, 2048, 128)).to(dev)
    for _ in range(3):
        tr3.step(inp3, gt3)
    graphed = tr3.capture(inp3, gt3)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        tr3.step(inp3, gt3)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    print(json.dumps({"workload": f"C3-style train step: 2048 rays x 128 given samples, 1024 junction latents, dbscan_enabled={dbscan}",
                      "launch": "hip graph" if graphed else "eager", "ms_per_step": 1e3 * dt, "ray_samples_per_s": 2048 * 128 / dt,
                      "rays_per_s": 2048 / dt, "precision": args.precision}))

# ---- C4 per-rank shape and C5-style hierarchical step
conf["dbscan_enabled"] = True
tr4 = Trainer(model_conf=conf, device=dev, state_dict={k: torch.tensor(v) for k, v in sd.items()})
tr4.model.set_precision(args.precision)
_, inp4, gt4 = synthetic_batch(42, 512, dev)
tr4.model.z_vals_override = torch.tensor(synth.synth_z_vals(42, 512, 128)).to(dev)
for _ in range(3):
    tr4.step(inp4, gt4)
graphed = tr4.capture(inp4, gt4)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(args.steps):
    tr4.step(inp4, gt4)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / args.steps
print(json.dumps({"workload": "C4 per-rank shape: 512 rays x 128 given samples, DTU switches (1024 junction latents, device DBSCAN), world size 1",
                  "launch": "hip graph" if graphed else "eager", "ms_per_step": 1e3 * dt, "ray_samples_per_s": 512 * 128 / dt,
                  "rays_per_s": 512 / dt, "precision": args.precision}), flush=True)

conf5 = copy.deepcopy(synth.ABC_NEAT_A_MODEL_CONF)
conf5.update(hip_sampler="hierarchical", hip_sampler_coarse=64, hip_sampler_fine=64)
tr5 = Trainer(model_conf=conf5, device=dev, state_dict={k: torch.tensor(v) for k, v in synth.synth_state_dict(42, "rough").items()})
tr5.model.set_precision(args.precision)
_, inp5, gt5 = synthetic_batch(42, 1024, dev)
for _ in range(3):
    tr5.step(inp5, gt5)
graphed5 = tr5.capture(inp5, gt5)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(args.steps):
    out5, _ = tr5.step(inp5, gt5)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / args.steps
S5 = out5["points"].shape[0] // 1024 if out5["points"].dim() == 2 else out5["points"].shape[1]
print(json.dumps({"workload": "C5-style train step: 1024 rays, hierarchical 64 coarse + 64 fine depths (HierarchicalSampler), abc model",
                  "launch": "hip graph" if graphed5 else "eager", "ms_per_step": 1e3 * dt, "samples_per_ray": S5,
                  "ray_samples_per_s": 1024 * S5 / dt, "rays_per_s": 1024 / dt, "precision": args.precision}), flush=True)
