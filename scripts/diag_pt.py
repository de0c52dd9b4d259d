import sys, torch
sys.path.insert(0, '.')
from neat_amd import synth, networks, rend_util, _lib
T = torch.tensor
dev = torch.device('cuda:0')
pt = int(sys.argv[1]); R = int(sys.argv[2]); S = int(sys.argv[3])
_lib.lib().neat_set_tuning(0, pt)
sd = synth.synth_state_dict(1, "rough")
m = networks.VolSDFNetwork(synth.ABC_NEAT_A_MODEL_CONF)
m.load_state_dict({k: T(v) for k, v in sd.items()}); m.to(dev).train(); m.set_precision('bf16')
sc = synth.synth_scene(seed=1, n_rays=R, view=1)
d, c = rend_util.get_camera_params(T(sc["uv"]).to(dev), T(sc["pose"]).to(dev), T(sc["intrinsics"]).to(dev))
d = d.reshape(-1, 3); c = c.expand(R, 3).contiguous()
z = T(synth.synth_z_vals(1, R, S)).to(dev)
out = m._render(c, d, z, False); torch.cuda.synchronize()
print('fwd ok', flush=True)
(out[0].sum() + out[1].sum()).backward(); torch.cuda.synchronize()
print('bwd ok pt', pt, R, S, flush=True)
