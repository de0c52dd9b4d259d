import sys, time, torch
from neat_amd import synth
from neat_amd.train import Trainer, synthetic_batch
dev = torch.device("cuda:0")
tr = Trainer(device=dev, state_dict={k: torch.tensor(v) for k, v in synth.synth_state_dict(42, "rough").items()})
tr.model.set_precision("bf16")
tr.model.eval()
_, inp2, _ = synthetic_batch(43, 2048, dev)
with torch.no_grad():
    for _ in range(6):
        o = tr.model(inp2)
torch.cuda.synchronize()
print("rounds", tr.model.ray_sampler.last_rounds)
