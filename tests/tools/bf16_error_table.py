import sys, numpy as np, torch
sys.path.insert(0,'.')
from tests.test_gpu_parity import *
from tests.util_replay import RngReplay
from neat_amd.loss import VolSDFLoss
from neat_amd import networks
dev=torch.device('cuda:0')
for (R,S,seed) in [(96,128,1),(33,50,2),(256,128,3)]:
    sd = synth.synth_state_dict(seed, "rough")
    sc = synth.synth_scene(seed=seed, n_rays=R, view=seed)
    z = T(synth.synth_z_vals(seed, R, S))
    gen = torch.Generator().manual_seed(seed)
    eik_idx = torch.randint(S, (R,), generator=gen)
    eik_uniform = torch.empty(R, 3).uniform_(-3, 3, generator=gen)
    p, ref, ref_lo = oracle_train_step(sd, sc, z, eik_idx, eik_uniform)
    for prec in (sys.argv[1:] or ["bf16","fp16","fp32"]):
        m = networks.VolSDFNetwork(synth.ABC_NEAT_A_MODEL_CONF)
        m.load_state_dict({k: T(v) for k, v in sd.items()})
        m.to(dev).train().set_precision(prec)
        m.z_vals_override = z.to(dev)
        with RngReplay([("randint", eik_idx), ("uniform_", eik_uniform)]):
            out = m(scene_inputs(sc, dev))
        lo = VolSDFLoss(**synth.ABC_NEAT_A_LOSS_CONF)(out, {"rgb": T(sc["gt_rgb"]).to(dev), "lines2d": T(sc["gt_lines2d"]).to(dev)})
        lo["loss"].backward()
        worst=(0,''); worstmax=(0,'')
        for k, prm in m.named_parameters():
            r = p[k].grad
            if r is None or r.numel() < 2: continue
            gq = prm.grad.detach().cpu()
            rel = float((gq-r).norm()/(r.norm()+1e-30)); mx=float((gq-r).abs().max()/(r.abs().max()+1e-30))
            if rel>worst[0]: worst=(rel,k)
            if mx>worstmax[0]: worstmax=(mx,k)
        outs={k: float((out[k].detach().cpu()-ref[k].detach()).abs().max()/max(1.0,float(ref[k].abs().max()))) for k in ("rgb_values","lines3d","depth","xyz","sdf","grad_theta")}
        print(R,S,prec,"worst relL2 %.3e (%s) worst max-rel %.3e (%s) loss err %.2e"%(worst[0],worst[1],worstmax[0],worstmax[1],abs(float(lo["loss"])-float(ref_lo["loss"]))), {k:"%.1e"%v for k,v in outs.items()})
