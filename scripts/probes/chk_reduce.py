"""Gradients of one full eager C2 train step with the two-stage vs the one-launch partial reduction (tuning key 6)."""
import sys, torch
sys.path.insert(0, '.')
from neat_amd import _lib, synth
from neat_amd.train import Trainer, synthetic_batch
dev = torch.device('cuda:0')
def run(v):
    _lib.check(_lib.lib().neat_set_tuning(6, v), "t")
    torch.manual_seed(42)
    tr = Trainer(device=dev, state_dict={k: torch.tensor(t) for k, t in synth.synth_state_dict(42, 'rough').items()})
    _, inp, gt = synthetic_batch(42, 1024, dev)
    tr.model.z_vals_override = torch.tensor(synth.synth_z_vals(42, 1024, 128)).to(dev)
    tr.model.set_precision('bf16')
    out = tr.model(inp); losses = tr.loss(out, gt)
    tr.optimizer.zero_grad(set_to_none=True)
    losses["loss"].backward()
    return {k: p.grad.detach().clone() for k, p in tr.model.named_parameters() if p.grad is not None}, float(losses["loss"])
(g0, l0), (g2, l2) = run(0), run(2)
print("loss", l0, l2)
for k in g0:
    err = float((g0[k] - g2[k]).abs().max()); ref = float(g0[k].abs().max())
    if err > 2e-5 * ref + 1e-12: print(k, tuple(g0[k].shape), err, ref)
print("done")
