"""Host time to enqueue graph-replayed train steps vs GPU time to run them (is the step host- or GPU-bound?)."""
import sys, time, torch
sys.path.insert(0, '.')
from neat_amd import synth
from neat_amd.train import Trainer, synthetic_batch
dev = torch.device('cuda:0')
tr = Trainer(device=dev, state_dict={k: torch.tensor(v) for k, v in synth.synth_state_dict(42, 'rough').items()})
_, inp, gt = synthetic_batch(42, 1024, dev)
tr.model.z_vals_override = torch.tensor(synth.synth_z_vals(42, 1024, 128)).to(dev)
tr.model.set_precision('bf16')
for _ in range(3): tr.step(inp, gt)
print("captured:", tr.capture(inp, gt))
for _ in range(5): tr.step(inp, gt)
torch.cuda.synchronize()
for n in (20, 50):
    t0 = time.perf_counter()
    for _ in range(n): tr.step(inp, gt)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{n} steps: host enqueue {1e3 * (t1 - t0) / n:.3f} ms/step, total {1e3 * (t2 - t0) / n:.3f} ms/step")
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(20): tr.step(inp, gt)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats('cumulative').print_stats(18)
