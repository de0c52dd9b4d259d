"""Which torch (ATen) GPU kernels does one C2 train step still launch, and from which Python line?"""
import sys, torch
sys.path.insert(0, '.')
from torch.profiler import profile, ProfilerActivity
from neat_amd import synth
from neat_amd.train import Trainer, synthetic_batch
dev = torch.device('cuda:0')
tr = Trainer(device=dev, state_dict={k: torch.tensor(v) for k, v in synth.synth_state_dict(42, "rough").items()})
_, inp, gt = synthetic_batch(42, 1024, dev)
tr.model.z_vals_override = torch.tensor(synth.synth_z_vals(42, 1024, 128)).to(dev)
tr.model.set_precision("bf16")
for _ in range(3):
    tr.step_eager(inp, gt)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    tr.step_eager(inp, gt)
    torch.cuda.synchronize()
rows = []
for ev in prof.events():
    if ev.device_type == torch.autograd.DeviceType.CPU and ev.name.startswith("aten::") and ev.device_time_total > 0:
        kids = [c for c in ev.cpu_children if c.name.startswith("aten::") and c.device_time_total > 0]
        if kids:
            continue
        st = [s_ for s_ in (ev.stack or []) if "neat_amd" in s_ or "bench" in s_]
        rows.append((ev.name, ev.device_time_total, st[0] if st else (ev.stack[-1] if ev.stack else "?")))
for name, t, where in rows:
    print(f"{name:32s} {t:7.1f} us   {where}")
print("total", sum(r[1] for r in rows), "us in", len(rows), "ops")
