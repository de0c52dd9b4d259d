"""Main pass only (neat_render_forward + neat_render_backward on 1024 rays x S samples + 2048 eikonal points, random cotangents), a few
times: the fused head chains without the junction block / loss around them (probe builds with wrong results cannot hang a matching).
Run under rocprofv3 --kernel-trace --stats (scripts/probes/hc_ab.sh).   python scripts/probes/hc_time.py [precision] [samples per ray] [tuning key=value ...]"""
import sys
import torch
sys.path.insert(0, '.')
from neat_amd import networks, synth, ops, _lib
for kv in sys.argv[3:]:               # tuning keys: key=value
    k, v = kv.split('=')
    _lib.check(_lib.lib().neat_set_tuning(int(k), int(v)), f"tuning {kv}")

dev = torch.device('cuda:0')
prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
S = int(sys.argv[2]) if len(sys.argv) > 2 else 128
R, E = 1024, 2048
m = networks.VolSDFNetwork(synth.ABC_NEAT_A_MODEL_CONF)
m.load_state_dict({k: torch.tensor(v) for k, v in synth.synth_state_dict(42, "rough").items()})
m.to(dev).train()
m.set_precision(prec)
g = torch.Generator().manual_seed(0)
o = (torch.rand(R, 3, generator=g) * 0.2 - 0.1 + torch.tensor([0.0, 0.0, -2.5])).to(dev)
d = torch.nn.functional.normalize(torch.randn(R, 3, generator=g) * 0.2 + torch.tensor([0.0, 0.0, 1.0]), dim=-1).to(dev)
z = (torch.sort(torch.rand(R, S, generator=g), dim=-1).values * 3.0 + 1.0).to(dev)
eik = (torch.rand(E, 3, generator=g) * 2 - 1).to(dev)
beta = torch.tensor([0.1], device=dev, requires_grad=True)
h = m.handle()
for _ in range(6):
    out = ops.render_rays(h, o, d, z, beta, 3.0, 1.0, False, eik)
    loss = sum(t.sum() for t in out[:5])
    loss.backward()
torch.cuda.synchronize()
print("ok", float(loss))
