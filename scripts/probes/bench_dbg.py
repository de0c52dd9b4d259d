import sys, time, torch
sys.path.insert(0, '.')
from neat_amd import synth
from neat_amd.train import Trainer, synthetic_batch
dev = torch.device("cuda:0")
torch.manual_seed(42)
tr = Trainer(device=dev, state_dict={k: torch.tensor(v) for k, v in synth.synth_state_dict(42, "rough").items()})
_, inp, gt = synthetic_batch(42, 1024, dev)
tr.model.z_vals_override = torch.tensor(synth.synth_z_vals(42, 1024, 128)).to(dev)
tr.model.set_precision("bf16")
for _ in range(3):
    tr.step(inp, gt); torch.cuda.synchronize()
print("capture", tr.capture(inp, gt))
torch.cuda.synchronize()
for rep in range(4):
    ts = []
    for _ in range(20):
        t0 = time.perf_counter()
        tr.step(inp, gt)
        ts.append(1e3 * (time.perf_counter() - t0))
    torch.cuda.synchronize()
    print("host ms per step:", " ".join("%.2f" % t for t in ts), "| replays", tr.replays, "eager", tr.eager_steps)
t0 = time.perf_counter()
for _ in range(20):
    _, losses = tr.step(inp, gt)
torch.cuda.synchronize()
print("20 steps:", 1e3 * (time.perf_counter() - t0) / 20)
