"""CPU experiment for VERDICT r5 #1c: can the fp16x3 build's SAMPLER queries run with fewer than three f16 products per MAC and still
reproduce the reference's depths G6 at the fp32-grade bar (tests/test_gpu_parity.py::close_sampler: <= 0.3 % of the samples off by more
than 2e-4)?  The oracle's Algorithm 1 is run with an SDF network whose matmuls emulate the split-f16 schemes (operands rounded to f16
planes, fp32 accumulation, activations kept as hi + lo between layers like kernels_x3.hpp):
    x3 : hi*hi + lo*hi + hi*lo (what runs today)      a2 : (hi + lo) activations x hi weights      w2 : hi activations x (hi + lo) weights
    x1 : hi*hi (= hip_sampler_fast_values)
    python tests/tools/sampler_products_emul.py
"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, ".")
from neat_amd import synth                      # noqa: E402
from oracle import neat_oracle as O             # noqa: E402

T = torch.tensor
GOLDEN = os.path.join("tests", "golden")
h16 = lambda x: x.to(torch.float16).float()


def planes(x):
    hi = h16(x)
    return hi, h16(x - hi)


def mm(a, W, scheme):
    ah, al = planes(a)
    wh, wl = planes(W)
    out = ah @ wh.t()
    if scheme in ("x3", "a2"):
        out = out + al @ wh.t()
    if scheme in ("x3", "w2"):
        out = out + ah @ wl.t()
    return out


def sdf_values(p, x, scheme):
    e = O.posenc(x, 6)
    h = e
    for l in range(9):
        if l == 4:
            h = torch.cat([h, e], 1) / O.SQRT2
        W = O.wn_weight(p, f"implicit_network.lin{l}")
        h = mm(h, W, scheme) + p[f"implicit_network.lin{l}.bias"]
        if l < 8:
            h = O._softplus100(h)
    return O.sphere_clamp(h[:, :1], x, 3.0, 20.0)


def main():
    torch.manual_seed(0)
    for variant in ("init", "rough"):
        p = O.params_from_numpy(synth.synth_state_dict(42, variant))
        for mode in ("eval", "train"):
            g = dict(np.load(os.path.join(GOLDEN, f"g6_sampler_{mode}_{variant}.npz")))
            d, o = O.camera_rays(T(g["uv"]), T(g["pose"]), T(g["intrinsics"]))
            d = d.reshape(-1, 3)
            o = o.expand(d.shape[0], 3)
            rand = {k: T(g[k]) for k in (("t_rand", "u_final", "perm", "eik_idx") if mode == "train" else ("eik_idx",))}
            row = []
            for scheme in ("f32", "x3", "a2", "w2", "x1"):
                fn = (lambda x: O.sdf_values(p, x)) if scheme == "f32" else (lambda x, s=scheme: sdf_values(p, x, s))
                with torch.no_grad():
                    z, _ = O.error_bound_sampler(fn, O.beta_of(p), d, o, training=mode == "train", rand=rand)
                err = (z - T(g["z_vals"])).abs().numpy()
                row.append(f"{scheme} {100 * float((err > 2e-4).mean()):6.3f} % (max {err.max():.3f})")
            print(f"{variant:5s} {mode:5s} | " + " | ".join(row), flush=True)
    print("bar: <= 0.300 % of the samples off by more than 2e-4 (close_sampler)")


if __name__ == "__main__":
    main()
