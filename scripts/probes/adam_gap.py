"""GPU-side duration of graph replay and of the Adam launch that follows it (HIP events), given depths vs sampler.  python scripts/probes/adam_gap.py [sampler]"""
import sys, torch
sys.path.insert(0, '.')
from neat_amd import synth
from neat_amd.train import Trainer, synthetic_batch
dev = torch.device("cuda:0")
torch.manual_seed(42)
tr = Trainer(device=dev, state_dict={k: torch.tensor(v) for k, v in synth.synth_state_dict(42, "rough").items()})
tr.model.set_precision("bf16")
_, inp, gt = synthetic_batch(42, 1024, dev)
if "sampler" not in sys.argv[1:]:
    tr.model.z_vals_override = torch.tensor(synth.synth_z_vals(42, 1024, 128)).to(dev)
for _ in range(3):
    tr.step(inp, gt)
assert tr.capture(inp, gt), tr.capture_error
entry = next(iter(tr._graphs.values()))
for _ in range(3):
    tr._finish_step(entry)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(4 * 10)]
torch.cuda.synchronize()
for i in range(10):
    ev[4 * i].record(); entry.graph.replay(); ev[4 * i + 1].record()
    ev[4 * i + 2].record(); tr.optimizer.step(); ev[4 * i + 3].record()
torch.cuda.synchronize()
g = [ev[4 * i].elapsed_time(ev[4 * i + 1]) for i in range(10)]
gap = [ev[4 * i + 1].elapsed_time(ev[4 * i + 2]) for i in range(10)]
a = [ev[4 * i + 2].elapsed_time(ev[4 * i + 3]) for i in range(10)]
nxt = [ev[4 * i + 3].elapsed_time(ev[4 * i + 4]) for i in range(9)]
print("graph ms      :", " ".join("%.3f" % x for x in g))
print("graph->adam ms:", " ".join("%.3f" % x for x in gap))
print("adam ms       :", " ".join("%.3f" % x for x in a))
print("adam->graph ms:", " ".join("%.3f" % x for x in nxt))
