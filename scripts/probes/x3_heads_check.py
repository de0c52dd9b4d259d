"""fp16x3 heads vs fp32 heads (stand-alone module API) at a multi-batch size: which points differ?"""
import sys, torch
sys.path.insert(0, '.')
from neat_amd import networks, synth, ops
dev = torch.device('cuda:0')
P = int(sys.argv[1]) if len(sys.argv) > 1 else 64 * 256 * 3 + 64
m = networks.VolSDFNetwork(synth.ABC_NEAT_A_MODEL_CONF)
m.load_state_dict({k: torch.tensor(v) for k, v in synth.synth_state_dict(42, "rough").items()})
m.to(dev).eval()
g = torch.Generator().manual_seed(0)
x = (torch.rand(P, 3, generator=g) * 2 - 1).to(dev)
n = torch.randn(P, 3, generator=g).to(dev)
v = torch.nn.functional.normalize(torch.randn(P, 3, generator=g), dim=-1).to(dev)
f = torch.randn(P, 256, generator=g).to(dev)
res = {}
for prec in ("fp32", "fp16x3"):
    m.set_precision(prec)
    with torch.no_grad():
        res[prec] = (m.rendering_network(x, n, v, f), m.attraction_network(x, n, v, f))
for i, name in enumerate(("rgb", "lines")):
    d = (res["fp16x3"][i] - res["fp32"][i]).abs().reshape(P, -1).max(1).values
    bad = (d > 1e-4).nonzero().flatten()
    print(name, "max err", float(d.max()), "bad points", bad.numel(), "batches of bad points:", sorted(set((bad // 64).tolist()))[:20])
