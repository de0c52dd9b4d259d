import sys, torch
sys.path.insert(0, '.')
from neat_amd import synth
from neat_amd.train import Trainer, synthetic_batch
dev = torch.device("cuda:0")
torch.manual_seed(42)
tr = Trainer(device=dev, state_dict={k: torch.tensor(v) for k, v in synth.synth_state_dict(42, "rough").items()})
tr.model.set_precision(sys.argv[1] if len(sys.argv) > 1 else "bf16")
_, inp, gt = synthetic_batch(42, int(sys.argv[2]) if len(sys.argv) > 2 else 1024, dev)
n = int(sys.argv[3]) if len(sys.argv) > 3 else 3
sync = len(sys.argv) > 4
for i in range(n):
    out, lo = tr.step(inp, gt)
    if sync:
        torch.cuda.synchronize()
    print("step", i, "rounds", tr.model.ray_sampler.last_rounds, flush=True)
torch.cuda.synchronize()
print("done", float(lo["loss"].detach()))
