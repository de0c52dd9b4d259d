import sys, numpy as np, torch
sys.path.insert(0, '.')
from neat_amd import synth, networks, rend_util
from tests.util_replay import RngReplay
T = torch.tensor
dev = torch.device('cuda:0')
for variant in ('init', 'rough'):
    g = dict(np.load(f'tests/golden/g6_sampler_eval_{variant}.npz'))
    m = networks.VolSDFNetwork(synth.ABC_NEAT_A_MODEL_CONF)
    m.load_state_dict({k: T(v) for k, v in synth.synth_state_dict(42, variant).items()})
    m.to(dev).eval()
    d, c = rend_util.get_camera_params(T(g['uv']).to(dev), T(g['pose']).to(dev), T(g['intrinsics']).to(dev))
    d = d.reshape(-1, 3); c = c.expand(d.shape[0], 3).contiguous()
    with RngReplay([('randint', None), ('randint', T(g['eik_idx']))]):
        z, ze = m.ray_sampler.get_z_vals(d, c, m)
    err = np.abs(z.cpu().numpy() - g['z_vals'])
    print(variant, 'rounds', m.ray_sampler.last_rounds, 'max', err.max(), 'n>1e-4', (err > 1e-4).sum(), 'n>1e-3', (err > 1e-3).sum(), 'of', err.size,
          'rays affected', (err.max(1) > 1e-4).sum())
    bad = np.argwhere(err > 1e-3)
    print(bad[:10])
