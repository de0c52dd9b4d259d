"""Fused head chains (tuning key 14: 2 = forward + backward, 1 = forward, 0 = per-layer launches) against the per-layer path on the same
build: one C2-shaped train step (1024 rays x S samples + 2048 eikonal points), outputs and all gradients; then the step times.
  python scripts/probes/head_chain_check.py [precision] [samples per ray]"""
import sys, time
import torch
sys.path.insert(0, '.')
from neat_amd import synth, _lib
from neat_amd.train import Trainer, synthetic_batch

dev = torch.device('cuda:0')
prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
S = int(sys.argv[2]) if len(sys.argv) > 2 else 128
R = 1024


def run(key):
    _lib.check(_lib.lib().neat_set_tuning(14, key), "tuning")
    torch.manual_seed(0)
    import numpy as np, random
    np.random.seed(0); random.seed(0)
    tr = Trainer(device=dev, state_dict={k: torch.tensor(v) for k, v in synth.synth_state_dict(42, "rough").items()})
    _, inp, gt = synthetic_batch(42, R, dev)
    tr.model.z_vals_override = torch.tensor(synth.synth_z_vals(42, R, S)).to(dev)
    tr.model.set_precision(prec)
    out = tr.model(inp)
    loss = tr.loss(out, gt)["loss"]
    tr.optimizer.zero_grad(set_to_none=True)
    loss.backward()
    grads = {n: p.grad.detach().clone() for n, p in tr.model.named_parameters() if p.grad is not None}
    outs = {k: v.detach().clone() for k, v in out.items() if torch.is_tensor(v) and v.dtype.is_floating_point}
    # timing: eager steps
    for _ in range(3):
        tr.step_eager(inp, gt)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        tr.step_eager(inp, gt)
    torch.cuda.synchronize()
    return outs, grads, float(loss), (time.perf_counter() - t0) / 10 * 1e3


ref = run(0)
for key in (1, 2):
    got = run(key)
    print(f"key 14 = {key}: loss {got[2]:.7f} (per-layer {ref[2]:.7f}); eager step {got[3]:.3f} ms (per-layer {ref[3]:.3f})")
    worst = ("", 0.0)
    for k, v in got[0].items():
        if k in ref[0] and v.shape == ref[0][k].shape:
            d = float((v - ref[0][k]).abs().max())
            if d > worst[1]:
                worst = (k, d)
    print("   outputs: worst max |d|", worst)
    wg = ("", 0.0)
    for k, g in got[1].items():
        r = ref[1][k]
        rel = float((g - r).norm() / (r.norm() + 1e-30))
        if rel > wg[1]:
            wg = (k, rel)
        if not torch.isfinite(g).all():
            print("   NON-FINITE gradient", k)
    print("   gradients: worst relative L2", wg)
