import sys, torch
sys.path.insert(0,'.')
from neat_amd import synth
from neat_amd.train import Trainer, synthetic_batch
T=torch.tensor
dev=torch.device('cuda:0')
torch.manual_seed(5)
tr = Trainer(device=dev, state_dict={k: T(v) for k, v in synth.synth_state_dict(5, "rough").items()})
_, inp, gt = synthetic_batch(31, 64, dev)
tr.model.z_vals_override = T(synth.synth_z_vals(31, 64, 32)).to(dev)
bad = dict(gt); bad["lines2d"] = gt["lines2d"].clone(); bad["lines2d"][..., 4] = float("nan")
out = tr.model(inp); lo = tr.loss(out, bad)
print({k: float(v) for k,v in lo.items() if torch.is_tensor(v) and v.numel()==1})
print('flag', tr.loss.nan_flag, getattr(tr.loss,'_pending',None))
