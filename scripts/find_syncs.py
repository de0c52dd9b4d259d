"""List host<->device synchronisation points in one train step (torch sync debug mode)."""
import sys, warnings, traceback, torch
sys.path.insert(0, '.')
from neat_amd import synth
from neat_amd.train import Trainer, synthetic_batch
dev = torch.device('cuda:0')
tr = Trainer(device=dev, state_dict={k: torch.tensor(v) for k, v in synth.synth_state_dict(42, 'rough').items()})
_, inp, gt = synthetic_batch(42, 1024, dev)
tr.model.z_vals_override = torch.tensor(synth.synth_z_vals(42, 1024, 128)).to(dev)
tr.model.set_precision('bf16')
for _ in range(3):
    tr.step(inp, gt)
torch.cuda.synchronize()
def showwarning(message, category, filename, lineno, file=None, line=None):
    st = [f for f in traceback.extract_stack() if 'neat_amd' in f.filename or 'scripts' in f.filename]
    print("SYNC:", str(message)[:80], "|", " <- ".join(f"{f.filename.split('/')[-1]}:{f.lineno}" for f in reversed(st[-3:])))
warnings.showwarning = showwarning
warnings.simplefilter("always")
torch.cuda.set_sync_debug_mode("warn")
tr.step(inp, gt)
torch.cuda.set_sync_debug_mode("default")
torch.cuda.synchronize()
