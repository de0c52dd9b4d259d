"""Fused adjoint chain (tuning key 13 = 1) against the seed + eight streaming launches (key 13 = 0): normals and the full train-step
gradients must agree bit for bit (same bf16 products, same k order, same epilogue arithmetic); then the times.
    python scripts/probes/adj_ab.py [P]"""
import sys
import torch
sys.path.insert(0, '.')
from neat_amd import _lib, networks, synth
from neat_amd.train import Trainer, synthetic_batch

dev = torch.device('cuda:0')
P = int(sys.argv[1]) if len(sys.argv) > 1 else 133120
lib = _lib.lib()
m = networks.VolSDFNetwork(synth.ABC_NEAT_A_MODEL_CONF)
m.load_state_dict({k: torch.tensor(v) for k, v in synth.synth_state_dict(42, "rough").items()})
m.to(dev).eval().set_precision("bf16")
x = (torch.rand(P, 3, generator=torch.Generator().manual_seed(0)) * 4 - 2).to(dev)


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


res, tm = {}, {}
for k in (0, 1):
    _lib.check(lib.neat_set_tuning(13, k), "tuning")
    with torch.no_grad():
        s, f, g = m.implicit_network.get_outputs(x)
        res[k] = (s.clone(), f.clone(), g.clone())
        tm[k] = timeit(lambda: m.implicit_network.get_outputs(x))
for name, a, b in zip(("sdf", "feat", "grad"), res[0], res[1]):
    d = float((a - b).abs().max())
    print(f"{name:5s} max |fused - streamed| = {d:.3e} (scale {float(a.abs().max()):.3e})  finite {bool(torch.isfinite(b).all())}")
print(f"get_outputs: streamed {tm[0]:.1f} us, fused {tm[1]:.1f} us")

grads = {}
for k in (0, 1):
    _lib.check(lib.neat_set_tuning(13, k), "tuning")
    torch.manual_seed(1)
    tr = Trainer(device=dev, state_dict={kk: torch.tensor(v) for kk, v in synth.synth_state_dict(42, "rough").items()})
    tr.model.set_precision("bf16")
    _, inp, gt = synthetic_batch(42, 256, dev)
    tr.model.z_vals_override = torch.tensor(synth.synth_z_vals(42, 256, 64)).to(dev)
    out = tr.model(inp)
    lo = tr.loss(out, gt)
    lo["loss"].backward()
    grads[k] = {n: p.grad.clone() for n, p in tr.model.named_parameters()}
worst = max(float((grads[0][n] - grads[1][n]).abs().max()) for n in grads[0])
print("train-step gradients: max |fused - streamed| over all tensors =", worst)
_lib.check(lib.neat_set_tuning(13, 1), "tuning")
