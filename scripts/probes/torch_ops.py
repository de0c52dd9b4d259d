"""Which torch ops (not neat_hip launches) does one eager training step issue, and from where?
python scripts/probes/torch_ops.py   (on the GPU box)"""
import sys, collections, traceback, torch
sys.path.insert(0, '.')
from torch.utils._python_dispatch import TorchDispatchMode
from neat_amd import synth
from neat_amd.train import Trainer, synthetic_batch
dev = torch.device('cuda:0')
tr = Trainer(device=dev, state_dict={k: torch.tensor(v) for k, v in synth.synth_state_dict(42, 'rough').items()})
_, inp, gt = synthetic_batch(42, 1024, dev)
tr.model.z_vals_override = torch.tensor(synth.synth_z_vals(42, 1024, 128)).to(dev)
tr.model.set_precision('bf16')
for _ in range(3): tr.step_eager(inp, gt)
torch.cuda.synchronize()
SKIP = ('view', 'reshape', 'expand', 'select', 'slice', 'unsqueeze', 'squeeze', 'detach', 't.default', 'transpose', 'alias', 'split', 'as_strided',
        'empty', 'unbind', 'permute', '_unsafe_view', 'is_pinned', 'lift_fresh', 'sym_', '_local_scalar')
agg = collections.Counter()
class Log(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if not any(s in name for s in SKIP):
            fr = [f for f in traceback.extract_stack() if '/neat_amd/' in f.filename]
            where = ' < '.join(f"{f.filename.split('/neat_amd/')[-1]}:{f.lineno}" for f in fr[-2:][::-1]) if fr else '(autograd engine)'
            shp = [tuple(a.shape) for a in args if isinstance(a, torch.Tensor)][:2]
            agg[(where, name, str(shp))] += 1
        return func(*args, **(kwargs or {}))
with Log():
    tr.step_eager(inp, gt)
torch.cuda.synchronize()
for (where, name, shp), c in sorted(agg.items()):
    print(f"{c:3d} {name:34s} {shp:40s} {where}")
print(sum(agg.values()), "ops")
