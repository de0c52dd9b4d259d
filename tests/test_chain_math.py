"""The per-layer algebra implemented by the HIP kernels (tests/chain_emulation.py) equals autograd's
double backward through the oracle SDF network.  CPU only, float64 for a tight check."""
import torch

from neat_amd import synth
from oracle import neat_oracle as O
from tests import chain_emulation as C


def test_double_backward_algebra():
    torch.manual_seed(0)
    p = O.params_from_numpy(synth.synth_state_dict(3, "rough"))
    p = {k: v.double() for k, v in p.items()}
    W = [O.wn_weight(p, f"implicit_network.lin{l}").clone().requires_grad_(True) for l in range(9)]
    b = [p[f"implicit_network.lin{l}.bias"].clone().requires_grad_(True) for l in range(9)]
    x = (torch.rand(37, 3, dtype=torch.float64) * 2 - 1) * 1.2
    A, B = torch.randn(37, 257, dtype=torch.float64), torch.randn(37, 3, dtype=torch.float64)
    # autograd reference on effective weights
    xr = x.clone().requires_grad_(True)
    _, out, _ = C.forward(W, b, xr)
    (g,) = torch.autograd.grad(out[:, 0].sum(), xr, create_graph=True)
    ((out * A).sum() + (g * B).sum()).backward()
    with torch.no_grad():
        Wd, bd = [w.detach() for w in W], [v.detach() for v in b]
        h, out2, E = C.forward(Wd, bd, x)
        u, g2 = C.adjoint(Wd, h, x)
        assert (out2 - out).abs().max() < 1e-12 and (g2 - g).abs().max() < 1e-7   # softplus threshold: 1-exp(-20) vs 1
        dW, db = C.backward(Wd, h, u, x, E, A, B)
    for l in range(9):
        assert (dW[l] - W[l].grad).abs().max() <= 1e-7 * (1 + W[l].grad.abs().max()), l
        assert (db[l] - b[l].grad).abs().max() <= 1e-7 * (1 + b[l].grad.abs().max()), l
