import sys, copy, torch
sys.path.insert(0, '.')
from neat_amd import synth, ops
from neat_amd.train import Trainer, synthetic_batch
dev = torch.device("cuda:0")
conf = copy.deepcopy(synth.ABC_NEAT_A_MODEL_CONF)
conf.update(dbscan_enabled=True, use_median=False)
conf["global_junctions"] = dict(conf["global_junctions"], num_junctions=1024)
sd = synth.synth_state_dict(42, "rough", num_junctions=1024)
tr3 = Trainer(model_conf=conf, device=dev, state_dict={k: torch.tensor(v) for k, v in sd.items()})
tr3.model.set_precision("bf16")
_, inp3, gt3 = synthetic_batch(42, 2048, dev)
tr3.model.z_vals_override = torch.tensor(synth.synth_z_vals(42, 2048, 128)).to(dev)
for i in range(4):
    out, lo = tr3.step(inp3, gt3)
torch.cuda.synchronize()
print("done", float(lo["loss"].detach()))
