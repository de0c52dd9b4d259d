import sys, time, torch
sys.path.insert(0, '.')
from neat_amd import synth
from neat_amd.train import Trainer, synthetic_batch
dev = torch.device('cuda:0')
sd = {k: torch.tensor(v) for k, v in synth.synth_state_dict(42, "rough").items()}
_, inp, gt = synthetic_batch(42, 1024, dev)
for trial in range(2):
    torch.manual_seed(42)
    tr = Trainer(device=dev, state_dict=sd)
    tr.model.set_precision("bf16")
    tr.model.ray_sampler.sync_free = True
    for _ in range(3):
        tr.step(inp, gt)
    tr.capture(inp, gt)
    torch.cuda.synchronize()
    ts = []
    for i in range(24):
        t0 = time.perf_counter()
        tr.step(inp, gt)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        ts.append((1e3 * (t1 - t0), 1e3 * (time.perf_counter() - t0)))
    print("trial", trial, "host ms / total ms per step:", " ".join("%.2f/%.2f" % t for t in ts), flush=True)
    # without per-step sync
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(20): tr.step(inp, gt)
    torch.cuda.synchronize(); print("   20 steps unsynced: %.3f ms/step" % ((time.perf_counter() - t0) / 20 * 1e3), flush=True)
    del tr
