import sys, numpy as np, torch
sys.path.insert(0, '.')
from neat_amd import synth, networks
from oracle import neat_oracle as O
T = torch.tensor
dev = torch.device('cuda:0')
sd = synth.synth_state_dict(1, "rough")
m = networks.VolSDFNetwork(synth.ABC_NEAT_A_MODEL_CONF)
m.load_state_dict({k: T(v) for k, v in sd.items()}); m.to(dev).train(); m.set_precision('bf16')
p = O.params_from_numpy(sd)
g = torch.Generator().manual_seed(0)
x = (torch.rand(200, 3, generator=g) * 2 - 1) * 1.5
print('step 1: values', flush=True)
with torch.no_grad():
    s = m.implicit_network.get_sdf_vals(x.to(dev)); torch.cuda.synchronize()
ref = O.sdf_values(p, x)
print('  max err', float((s.cpu() - ref).abs().max()), flush=True)
print('step 2: outputs fwd', flush=True)
sdf, feat, grad = m.implicit_network.get_outputs(x.to(dev)); torch.cuda.synchronize()
rs, rf, rg = O.sdf_outputs(p, x)
print('  sdf', float((sdf.cpu() - rs).abs().max()), 'feat', float((feat.detach().cpu() - rf).abs().max()), 'grad', float((grad.detach().cpu() - rg).abs().max()), flush=True)
print('step 3: backward', flush=True)
(sdf.sum() + (grad ** 2).sum() + feat.mean()).backward(); torch.cuda.synchronize()
print('  ok, grad norm lin4.weight_v', float(m.implicit_network.lin4.weight_v.grad.norm()), flush=True)
print('step 4: render fwd', flush=True)
from neat_amd import rend_util
R, S = 96, 128
sc = synth.synth_scene(seed=1, n_rays=R, view=1)
d, c = rend_util.get_camera_params(T(sc["uv"]).to(dev), T(sc["pose"]).to(dev), T(sc["intrinsics"]).to(dev))
d = d.reshape(-1, 3); c = c.expand(R, 3).contiguous()
z = T(synth.synth_z_vals(1, R, S)).to(dev)
m.zero_grad()
out = m._render(c, d, z, False); torch.cuda.synchronize()
print('  fwd ok', float(out[0].sum()), flush=True)
print('step 5: render bwd', flush=True)
(out[0].sum() + out[1].sum()).backward(); torch.cuda.synchronize()
print('  bwd ok', flush=True)
print('step 6: render with eik points', flush=True)
eik = (torch.rand(2 * R, 3) * 2 - 1).to(dev)
out = m._render(c, d, z, False, eik, with_eik=True); torch.cuda.synchronize()
print('  fwd ok', flush=True)
(out[0].sum() + out[-1].pow(2).sum()).backward(); torch.cuda.synchronize()
print('  bwd ok', flush=True)
