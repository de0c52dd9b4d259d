"""bf16 build vs CPU oracle and vs the fp32 build: error report (run on the GPU box)."""
import sys, numpy as np, torch
sys.path.insert(0, '.')
from neat_amd import synth, networks
from neat_amd.loss import VolSDFLoss
from tests.test_gpu_parity import oracle_train_step, scene_inputs
from tests.util_replay import RngReplay
T = torch.tensor
dev = torch.device('cuda:0')
R, S, seed = 96, 128, 1
sd = synth.synth_state_dict(seed, "rough"); sc = synth.synth_scene(seed=seed, n_rays=R, view=seed)
z = T(synth.synth_z_vals(seed, R, S))
gen = torch.Generator().manual_seed(seed)
eik_idx = torch.randint(S, (R,), generator=gen); eik_uniform = torch.empty(R, 3).uniform_(-3, 3, generator=gen)
p, ref, ref_lo = oracle_train_step(sd, sc, z, eik_idx, eik_uniform)
res = {}
for prec in ("fp32", "bf16"):
    m = networks.VolSDFNetwork(synth.ABC_NEAT_A_MODEL_CONF)
    m.load_state_dict({k: T(v) for k, v in sd.items()}); m.to(dev).train(); m.set_precision(prec)
    m.z_vals_override = z.to(dev)
    with RngReplay([("randint", eik_idx), ("uniform_", eik_uniform)]):
        out = m(scene_inputs(sc, dev))
    lo = VolSDFLoss(**synth.ABC_NEAT_A_LOSS_CONF)(out, {"rgb": T(sc["gt_rgb"]).to(dev), "lines2d": T(sc["gt_lines2d"]).to(dev)})
    lo["loss"].backward()
    torch.cuda.synchronize()
    print(f"== {prec}: loss {lo['loss'].item():.6f} (oracle {ref_lo['loss'].item():.6f})")
    for k in ("rgb_values", "lines3d", "depth", "xyz", "grad_theta", "sdf"):
        a, b = out[k].detach().cpu(), ref[k].detach()
        print(f"   {k:12s} max abs err {float((a-b).abs().max()):.3e}  (scale {float(b.abs().max()):.3e}) finite={bool(torch.isfinite(a).all())}")
    worst = []
    for k, prm in m.named_parameters():
        r = p[k].grad
        if r is None or prm.grad is None: continue
        g = prm.grad.cpu()
        rel = float((g - r).abs().max() / max(float(r.abs().max()), 1e-9))
        cos = float((g.flatten() @ r.flatten()) / (g.norm() * r.norm() + 1e-30))
        worst.append((rel, cos, k))
    worst.sort(reverse=True)
    print("   worst grads (max-rel-err, cosine, name):")
    for w in worst[:6]: print("     %.3e  %.6f  %s" % w)
    print("   median rel err %.3e ; min cosine %.6f" % (np.median([w[0] for w in worst]), min(w[1] for w in worst)))
