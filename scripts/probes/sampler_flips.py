import sys, numpy as np, torch
sys.path.insert(0,'.')
from tests.test_gpu_parity import *
import tests.test_gpu_parity as tg
from tests.util_replay import RngReplay
from neat_amd import rend_util
dev=torch.device('cuda:0')
import os
GOLD=lambda n: dict(np.load(os.path.join('tests/golden', n+'.npz')))
for variant in ("init","rough"):
    m = build_model(dev, variant)
    g = GOLD(f"g6_sampler_eval_{variant}")
    d, c = rend_util.get_camera_params(T(g["uv"]).to(dev), T(g["pose"]).to(dev), T(g["intrinsics"]).to(dev))
    d = d.reshape(-1, 3); c = c.expand(d.shape[0], 3).contiguous()
    with RngReplay([("randint", None), ("randint", T(g["eik_idx"]))]):
        z, ze = m.ray_sampler.get_z_vals(d, c, m)
    err=np.abs(z.cpu().numpy()-g["z_vals"]); print(variant,"eval: flips(>2e-4):",int((err>2e-4).sum()),"of",err.size,"max",err.max())
    g = GOLD(f"g6_sampler_train_{variant}")
    m.train()
    with RngReplay([("rand", T(g["t_rand"])), ("randint", None), ("rand", T(g["u_final"])), ("randperm", T(g["perm"])), ("randint", T(g["eik_idx"]))]):
        z, ze = m.ray_sampler.get_z_vals(d, c, m)
    err=np.abs(z.cpu().numpy()-g["z_vals"]); print(variant,"train: flips(>2e-4):",int((err>2e-4).sum()),"of",err.size,"max",err.max())
