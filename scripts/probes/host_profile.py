import sys, time, torch
sys.path.insert(0, '.')
from neat_amd import synth
from neat_amd.train import Trainer, synthetic_batch
from torch.profiler import profile, ProfilerActivity
dev = torch.device('cuda:0')
tr = Trainer(device=dev, state_dict={k: torch.tensor(v) for k, v in synth.synth_state_dict(42, 'rough').items()})
_, inp, gt = synthetic_batch(42, 1024, dev)
tr.model.z_vals_override = torch.tensor(synth.synth_z_vals(42, 1024, 128)).to(dev)
tr.model.set_precision('bf16')
for _ in range(3): tr.step(inp, gt)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU], record_shapes=False) as prof:
    for _ in range(5): tr.step(inp, gt)
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=40, max_name_column_width=60))
