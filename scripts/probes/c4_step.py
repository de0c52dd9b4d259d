"""C4's per-rank step alone (512 rays x 128 samples, DTU switches), graph replay: for kernel traces.  usage: python scripts/probes/c4_step.py [precision] [steps]"""
import copy, json, sys, time
import torch
sys.path.insert(0, '.')
from neat_amd import synth
from neat_amd.train import Trainer, synthetic_batch
prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dev = torch.device("cuda:0")
conf = copy.deepcopy(synth.ABC_NEAT_A_MODEL_CONF)
conf.update(dbscan_enabled=True, use_median=False)
conf["global_junctions"] = dict(conf["global_junctions"], num_junctions=1024)
sd = synth.synth_state_dict(42, "rough", num_junctions=1024)
tr = Trainer(model_conf=conf, device=dev, state_dict={k: torch.tensor(v) for k, v in sd.items()})
tr.model.set_precision(prec)
_, inp, gt = synthetic_batch(42, 512, dev)
tr.model.z_vals_override = torch.tensor(synth.synth_z_vals(42, 512, 128)).to(dev)
for _ in range(3):
    tr.step(inp, gt)
graphed = tr.capture(inp, gt)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    tr.step(inp, gt)
torch.cuda.synchronize()
print(json.dumps({"c4_ms_per_step": 1e3 * (time.perf_counter() - t0) / steps, "graph": bool(graphed)}))
