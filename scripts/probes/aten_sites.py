"""Which Python lines of neat_amd issue ATen operators on CUDA tensors during one C2 train step (forward + loss; the backward pass
runs their derivatives)?  TorchDispatchMode sees every operator call with its Python stack.   python scripts/probes/aten_sites.py"""
import sys, traceback, collections
import torch
sys.path.insert(0, '.')
from torch.utils._python_dispatch import TorchDispatchMode
from neat_amd import synth
from neat_amd.train import Trainer, synthetic_batch

dev = torch.device('cuda:0')
tr = Trainer(device=dev, state_dict={k: torch.tensor(v) for k, v in synth.synth_state_dict(42, "rough").items()})
_, inp, gt = synthetic_batch(42, 1024, dev)
tr.model.z_vals_override = torch.tensor(synth.synth_z_vals(42, 1024, 128)).to(dev)
tr.model.set_precision("bf16")
for _ in range(2):
    tr.step_eager(inp, gt)
sites = collections.Counter()


class Log(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        flat = [a for a in list(args) + list((kwargs or {}).values()) if isinstance(a, torch.Tensor)]
        outs = out if isinstance(out, (tuple, list)) else [out]
        cuda = any(t.is_cuda for t in flat) or any(isinstance(o, torch.Tensor) and o.is_cuda for o in outs)
        name = str(func)
        if cuda and not any(s in name for s in ("view", "reshape", "detach", "alias", "expand", "select", "slice", "squeeze", "unsqueeze", "t.default", "transpose", "permute", "as_strided", "_unsafe_view", "empty", "split", "unbind")):
            fr = [f for f in traceback.extract_stack() if "neat_amd" in f.filename and "scripts" not in f.filename]
            where = f"{fr[-1].filename.split('/')[-1]}:{fr[-1].lineno} {fr[-1].line}" if fr else "?"
            sites[(name, where)] += 1
        return out


with Log():
    out = tr.model(inp)
    losses = tr.loss(out, gt)
    tr.optimizer.zero_grad(set_to_none=True)
    losses["loss"].backward()
for (name, where), n in sorted(sites.items(), key=lambda kv: kv[0][1]):
    print(f"{n:3d} x {name:40s} {where}")
