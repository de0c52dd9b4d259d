"""bench.py's secondary legs, twice, next to a live headline trainer (as in bench.py): run-to-run spread of the replayed steps."""
import sys, torch
sys.path.insert(0, '.')
import bench
from neat_amd import synth
from neat_amd.train import Trainer, synthetic_batch
dev = torch.device('cuda:0')
sd = {k: torch.tensor(v) for k, v in synth.synth_state_dict(42, "rough").items()}
tr = Trainer(device=dev, state_dict=sd)
_, inp, gt = synthetic_batch(42, 1024, dev)
tr.model.z_vals_override = torch.tensor(synth.synth_z_vals(42, 1024, 128)).to(dev)
tr.model.set_precision("bf16")
for _ in range(3):
    tr.step(inp, gt)
tr.capture(inp, gt)
for _ in range(20):
    tr.step(inp, gt)
torch.cuda.synchronize()
for rep in range(2):
    legs = bench.secondary_legs(dev, sd, "bf16", 20, lambda m: None)
    print(rep, {k: round(v["ms_per_step"], 3) for k, v in legs.items()}, flush=True)
