"""Density models with the reference's names (code/model/density.py).  Only LaplaceDensity is used by the confs."""
import torch
from torch import nn


class Density(nn.Module):
    def __init__(self, params_init=None):
        super().__init__()
        for name, value in (params_init or {}).items():
            setattr(self, name, nn.Parameter(torch.tensor(float(value))))

    def forward(self, sdf, beta=None):
        return self.density_func(sdf, beta=beta)


class LaplaceDensity(Density):
    """sigma(s) = 1/beta * Laplace(0, beta).cdf(-s) ; beta = |beta_param| + beta_min  (density.py:16-30).
    Inside the fused main pass the density lives in the compositing kernels; this torch form serves the
    sampler's per-ray beta and external callers."""

    def __init__(self, params_init=None, beta_min=0.0001):
        super().__init__(params_init=params_init)
        self.beta_min = float(beta_min)

    def density_func(self, sdf, beta=None):
        if beta is None:
            beta = self.get_beta()
        return (0.5 + 0.5 * sdf.sign() * torch.expm1(-sdf.abs() / beta)) / beta

    def get_beta(self):
        # (no per-step cache of this two-launch expression: a cached tensor keeps its autograd graph -- and with it the parameter's
        # AccumulateGrad node, bound to the stream of the step that built it -- alive into the next step; a HIP-graph capture that then
        # reuses the node records a wait on that other stream and hipStreamEndCapture dies on the unjoined fork.  Tried in round 5.)
        return self.beta.abs() + self.beta_min
