"""Fused SDF primal chain: the kernel generations against each other at the C2 size (P = 133 120 points).
Agreement (same bf16 products, fp32 accumulation in the same k order) and device time per launch (HIP events).
    python scripts/probes/time_fused.py [P]
"""
import sys
import torch
sys.path.insert(0, '.')
from neat_amd import _lib, networks, synth

dev = torch.device('cuda:0')
P = int(sys.argv[1]) if len(sys.argv) > 1 else 133120
lib = _lib.lib()
m = networks.VolSDFNetwork(synth.ABC_NEAT_A_MODEL_CONF)
m.load_state_dict({k: torch.tensor(v) for k, v in synth.synth_state_dict(42, "rough").items()})
m.to(dev).eval().set_precision("bf16")
g = torch.Generator().manual_seed(0)
x = (torch.rand(P, 3, generator=g) * 4 - 2).to(dev)


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


res = {}
for gen in (1, 2, 3, 4):
    _lib.check(lib.neat_set_tuning(4, gen), "neat_set_tuning")
    with torch.no_grad():
        sdf = m.implicit_network.get_sdf_vals(x).clone()
        s2, feat, grad = m.implicit_network.get_outputs(x)
        res[gen] = (sdf, s2.clone(), feat.clone(), grad.clone())
        t_val = timeit(lambda: m.implicit_network.get_sdf_vals(x))
        t_out = timeit(lambda: m.implicit_network.get_outputs(x))
    print(f"generation {gen}: get_sdf_vals {t_val:8.1f} us   get_outputs (fused primal + adjoint chain + exports) {t_out:8.1f} us", flush=True)
for gen in (2, 3, 4):
  for name, a, b in zip(("sdf(values)", "sdf", "feat", "grad"), res[1], res[gen]):
    d = float((a - b).abs().max())
    print(f"  {name:12s} max |gen{gen} - gen1| = {d:.3e}  (scale {float(a.abs().max()):.3e})")
    assert torch.isfinite(b).all()
    assert d <= 2e-5 * max(1.0, float(a.abs().max())), name
_lib.check(lib.neat_set_tuning(4, 3), "neat_set_tuning")
print("OK")
