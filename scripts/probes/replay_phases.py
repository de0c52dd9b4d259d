"""Where a replayed step spends its time: the HIP graph alone, graph + Adam, the whole Trainer.step.
    python scripts/probes/replay_phases.py [sampler]      (sampler: depths from the ErrorBoundSampler instead of given depths)"""
import sys, time, torch
sys.path.insert(0, '.')
from neat_amd import synth
from neat_amd.train import Trainer, synthetic_batch
dev = torch.device("cuda:0")
torch.manual_seed(42)
tr = Trainer(device=dev, state_dict={k: torch.tensor(v) for k, v in synth.synth_state_dict(42, "rough").items()})
tr.model.set_precision("fp16" if "fp16" in sys.argv[1:] else "bf16")
_, inp, gt = synthetic_batch(42, 1024, dev)
if "sampler" not in sys.argv[1:]:
    tr.model.z_vals_override = torch.tensor(synth.synth_z_vals(42, 1024, 128)).to(dev)
for _ in range(3):
    tr.step(inp, gt)
assert tr.capture(inp, gt), tr.capture_error
entry = next(iter(tr._graphs.values()))


def timed(fn, n=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n


print("graph.replay() only      : %.3f ms" % timed(entry.graph.replay))
print("replay + Adam + scheduler: %.3f ms" % timed(lambda: tr._finish_step(entry)))
print("refill + replay + Adam   : %.3f ms" % timed(lambda: (tr._refill_randoms(entry), tr._finish_step(entry))))
print("Trainer.step             : %.3f ms" % timed(lambda: tr.step(inp, gt)))
lr = tr.optimizer.param_groups[0]["lr"]
tr.optimizer.param_groups[0]["lr"] = 0.0
tr.scheduler.base_lrs = [0.0]
print("replay + Adam (lr = 0: the weights stay put, same GPU work every step): %.3f ms" % timed(lambda: (entry.graph.replay(), tr.optimizer.step())))
tr.optimizer.param_groups[0]["lr"] = lr

# host cost of one launch (the call returns when the graph is enqueued)
torch.cuda.synchronize()
ts = []
for _ in range(10):
    t0 = time.perf_counter()
    entry.graph.replay()
    ts.append(1e3 * (time.perf_counter() - t0))
    torch.cuda.synchronize()
print("host time of graph.replay() with an idle device: " + " ".join("%.2f" % t for t in ts) + " ms")
