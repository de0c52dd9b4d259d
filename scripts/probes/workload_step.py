"""One secondary workload alone, for kernel traces:  python scripts/probes/workload_step.py {c2|c3|c4|c5|eval|sampler} [precision] [steps]
(train steps replay a HIP graph; `eval` = the no-grad forward of one 2048-ray chunk, sampler included)."""
import copy, json, sys, time
import torch
sys.path.insert(0, '.')
from neat_amd import synth
from neat_amd.train import Trainer, synthetic_batch
what = sys.argv[1]
prec = sys.argv[2] if len(sys.argv) > 2 else "bf16"
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
dev = torch.device("cuda:0")
conf = copy.deepcopy(synth.ABC_NEAT_A_MODEL_CONF)
nj = 64
if what in ("c3", "c4"):
    conf.update(dbscan_enabled=True, use_median=False)
    conf["global_junctions"] = dict(conf["global_junctions"], num_junctions=1024)
    nj = 1024
if what == "c5":
    conf.update(hip_sampler="hierarchical", hip_sampler_coarse=64, hip_sampler_fine=64)
R = {"c2": 1024, "c3": 2048, "c4": 512, "c5": 1024, "eval": 2048, "sampler": 1024}[what]
tr = Trainer(model_conf=conf, device=dev, state_dict={k: torch.tensor(v) for k, v in synth.synth_state_dict(42, "rough", num_junctions=nj).items()})
tr.model.set_precision(prec)
_, inp, gt = synthetic_batch(42 if what != "eval" else 43, R, dev)
if what in ("c2", "c3", "c4"):
    tr.model.z_vals_override = torch.tensor(synth.synth_z_vals(42, R, 128)).to(dev)
if what == "sampler":
    tr.model.ray_sampler.sync_free = True
if what == "eval":
    tr.model.eval()
    with torch.no_grad():
        for _ in range(3):
            tr.model(inp)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            tr.model(inp)
        torch.cuda.synchronize()
    print(json.dumps({"workload": what, "precision": prec, "ms": 1e3 * (time.perf_counter() - t0) / steps}))
    sys.exit(0)
for _ in range(3):
    tr.step(inp, gt)
graphed = tr.capture(inp, gt)
for _ in range(5):
    tr.step(inp, gt)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    tr.step(inp, gt)
torch.cuda.synchronize()
print(json.dumps({"workload": what, "precision": prec, "ms": 1e3 * (time.perf_counter() - t0) / steps, "graph": bool(graphed)}))
