"""Where does the host spend the eager train step with the sampler on?  Per-step wall times with and without the cyclic GC, then a
cProfile of 10 steps.  usage: python scripts/probes/host_stall.py"""
import cProfile, gc, pstats, sys, time
import torch
sys.path.insert(0, '.')
from neat_amd import synth
from neat_amd.train import Trainer, synthetic_batch

dev = torch.device("cuda:0")
torch.manual_seed(42)
tr = Trainer(device=dev, state_dict={k: torch.tensor(v) for k, v in synth.synth_state_dict(42, "rough").items()})
tr.model.set_precision("bf16")
_, inp, gt = synthetic_batch(42, 1024, dev)


def steps(n):
    out = []
    for _ in range(n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        tr.step(inp, gt)
        torch.cuda.synchronize()
        out.append(1e3 * (time.perf_counter() - t0))
    return out


steps(3)
print("gc on :", " ".join(f"{t:.1f}" for t in steps(12)), "gc counts", gc.get_count(), "objects", len(gc.get_objects()))
gc.disable()
print("gc off:", " ".join(f"{t:.1f}" for t in steps(12)))
gc.enable()
gc.freeze()
print("gc on, startup heap frozen:", " ".join(f"{t:.1f}" for t in steps(12)))
tr.model.ray_sampler.sync_free = True
steps(3)
print("sync-free, frozen:", " ".join(f"{t:.1f}" for t in steps(12)))
gc.disable()
print("sync-free, gc off:", " ".join(f"{t:.1f}" for t in steps(12)))
pr = cProfile.Profile()
pr.enable()
steps(10)
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
