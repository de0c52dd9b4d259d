"""One training step of the hot path as the reference trainer runs it (code/training/volsdf_train.py:361-374,408):
forward -> loss -> zero_grad/backward -> [gradient all-reduce] -> Adam step -> per-iteration ExponentialLR."""
import torch

from . import networks, synth
from .dp import FlatGradBucket
from .loss import VolSDFLoss
from .wireframe import WireframeGraph


class Trainer:
    def __init__(self, model_conf=None, loss_conf=None, lr=5.0e-4, decay_steps=200000, device="cuda:0", state_dict=None):
        self.device = torch.device(device)
        self.model = networks.VolSDFNetwork(model_conf or synth.ABC_NEAT_A_MODEL_CONF)
        if state_dict is not None:
            self.model.load_state_dict(state_dict)
        self.model.to(self.device).train()
        self.loss = VolSDFLoss(**(loss_conf or synth.ABC_NEAT_A_LOSS_CONF))
        if self.device.type == "cuda":
            from .optim import FlatAdam
            self.optimizer = FlatAdam(self.model.parameters(), lr=lr)      # torch.optim.Adam semantics, one HIP launch
        else:
            self.optimizer = torch.optim.Adam(self.model.parameters(), lr=lr)
        self.scheduler = torch.optim.lr_scheduler.ExponentialLR(self.optimizer, 0.1 ** (1.0 / decay_steps))
        self.bucket = FlatGradBucket(self.model.parameters())

    def step(self, model_input, ground_truth):
        out = self.model(model_input)
        losses = self.loss(out, ground_truth)
        self.optimizer.zero_grad(set_to_none=True)
        losses["loss"].backward()
        self.bucket.all_reduce_mean()
        self.optimizer.step()
        self.scheduler.step()
        return out, losses


def synthetic_batch(seed, n_rays, device, view=0):
    """Model input + ground truth in the trainer's layout from neat_amd.synth (no dataset, no network access)."""
    sc = synth.synth_scene(seed=seed, n_rays=n_rays, view=view)
    wf = WireframeGraph(torch.tensor(sc["wf_vertices"]), torch.tensor(sc["wf_vconf"]), torch.tensor(sc["wf_edges"]),
                        torch.tensor(sc["wf_weights"]), sc["res"], sc["res"])
    inp = {k: torch.tensor(sc[k]).to(device) for k in ("intrinsics", "pose", "uv", "uv_proj")}
    inp["wireframe"] = [wf]
    gt = {"rgb": torch.tensor(sc["gt_rgb"]).to(device), "lines2d": torch.tensor(sc["gt_lines2d"]).to(device)}
    return sc, inp, gt
