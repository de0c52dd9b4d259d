"""Forward pass of the C2 workload (1024 rays x 128 samples + 2048 eikonal points, train mode: everything saved) under precision
fp16x3, a few times; run it under rocprofv3 --kernel-trace --stats to get the per-launch times of the split-precision chains
(scripts/probes/x3_ab.sh does, for a list of probe builds)."""
import sys
import torch
sys.path.insert(0, '.')
from neat_amd import synth
from neat_amd.train import Trainer, synthetic_batch

dev = torch.device('cuda:0')
prec = sys.argv[1] if len(sys.argv) > 1 else "fp16x3"
tr = Trainer(device=dev, state_dict={k: torch.tensor(v) for k, v in synth.synth_state_dict(42, "rough").items()})
_, inp, gt = synthetic_batch(42, 1024, dev)
tr.model.z_vals_override = torch.tensor(synth.synth_z_vals(42, 1024, 128)).to(dev)
tr.model.set_precision(prec)
if len(sys.argv) > 2 and sys.argv[2] == "eval":      # forward-only variants (nothing saved)
    tr.model.eval()
    with torch.no_grad():
        for _ in range(6):
            out = tr.model(inp)
            sv = tr.model.implicit_network.get_sdf_vals(torch.rand(133120, 3, device=dev) * 2 - 1)
else:
    for _ in range(6):
        out = tr.model(inp)
torch.cuda.synchronize()
print("ok", float(out["rgb_values"].sum()))
