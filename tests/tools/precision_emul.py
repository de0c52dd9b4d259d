"""CPU emulation of mixed-precision schemes for the train step (design tool, not product, not a test).

The oracle's F.linear calls are replaced by matmuls whose OPERANDS are rounded per role -- primal / adjoint (the
create_graph pass for the normals) / tangent / reverse / the two weight-gradient pairs of the SDF network, forward / backward /
weight gradient of the heads -- and whose activations are rounded at storage.  Everything else stays fp32, as in the kernels.
Prints, per scheme, the output errors (relative to each tensor's scale; bar 1e-4) and the worst gradient error (max-abs over
max; bar 2e-3) against the plain fp32 oracle on the same inputs.

    python tests/tools/precision_emul.py [R S]
"""
import math
import sys

import torch

sys.path.insert(0, ".")
from neat_amd import synth                      # noqa: E402
from neat_amd.wireframe import WireframeGraph   # noqa: E402
from oracle import neat_oracle as O             # noqa: E402

T = torch.tensor
PHASE = {"v": 1}          # 1: forward (+ the create_graph pass), 2: loss.backward()
CFG = {}


def q1(x, m):
    if m == "f32":
        return x
    if m == "bf16":
        return x.to(torch.bfloat16).float()
    if m == "f16":
        return x.to(torch.float16).float()
    if m == "f16s":                    # fp16 after a per-tensor power-of-two scale (largest magnitude -> [1, 2))
        mx = float(x.abs().max())
        if mx == 0.0:
            return x
        s = 2.0 ** (-math.floor(math.log2(mx)))
        return (x * s).to(torch.float16).float() / s
    raise ValueError(m)


def split(x, m):
    """x -> list of planes whose sum represents x in mode m"""
    if m in ("f32", "bf16", "f16", "f16s"):
        return [q1(x, m)]
    base = m[:-2]                      # 'bf16x2' / 'f16x2'
    hi = q1(x, base)
    return [hi, q1(x - hi, base)]


def qmm(a, b, ma, mb):
    """a @ b with operands in modes ma / mb; two-plane operands drop the lo*lo product like the kernels"""
    pa, pb = split(a, ma), split(b, mb)
    out = pa[0] @ pb[0]
    if len(pa) > 1:
        out = out + pa[1] @ pb[0]
    if len(pb) > 1:
        out = out + pa[0] @ pb[1]
    return out


def role_of(net, kind):
    """kind: 'y' (x W^T), 'dx' (g W), 'dw' (g^T x); depth bookkeeping through PHASE and the nesting flag"""
    return CFG.get((net, kind, PHASE["v"]), CFG.get((net, kind), ("f32", "f32")))


class Y(torch.autograd.Function):            # y = x @ W^T
    @staticmethod
    def forward(ctx, x, W, net, kind):
        ctx.save_for_backward(x, W)
        ctx.net, ctx.kind = net, kind
        ma, mb = role_of(net, kind)
        return qmm(x, W.t(), ma, mb)

    @staticmethod
    def backward(ctx, g):
        x, W = ctx.saved_tensors
        # derivative roles: a 'y' of the forward pass gives 'dx' / 'dw'; the 'y' that appears as derivative of an adjoint dx
        # (tangent) gives reverse-of-tangent terms that the step never needs beyond first order -> f32
        kx = {"primal": "adjoint" if PHASE["v"] == 1 else "reverse", "tangent": "f32", "head": "head_dx"}[ctx.kind]
        kw = {"primal": "wgrad1", "tangent": "f32", "head": "head_dw"}[ctx.kind]
        gx = DX.apply(g, W, ctx.net, kx) if ctx.needs_input_grad[0] else None
        gW = DW.apply(g, x, ctx.net, kw) if ctx.needs_input_grad[1] else None
        return gx, gW, None, None


class DX(torch.autograd.Function):           # gx = g @ W
    @staticmethod
    def forward(ctx, g, W, net, kind):
        ctx.save_for_backward(g, W)
        ctx.net, ctx.kind = net, kind
        ma, mb = role_of(net, kind)
        return qmm(g, W, ma, mb)

    @staticmethod
    def backward(ctx, y):
        g, W = ctx.saved_tensors
        if ctx.kind == "adjoint":
            gg = Y.apply(y, W, ctx.net, "tangent") if ctx.needs_input_grad[0] else None
            gW = DW.apply(g, y, ctx.net, "wgrad2") if ctx.needs_input_grad[1] else None
        else:
            gg = y @ W.t() if ctx.needs_input_grad[0] else None
            gW = g.t() @ y if ctx.needs_input_grad[1] else None
        return gg, gW, None, None


class DW(torch.autograd.Function):           # gW = g^T @ x
    @staticmethod
    def forward(ctx, g, x, net, kind):
        ctx.save_for_backward(g, x)
        ma, mb = role_of(net, kind)
        return qmm(g.t(), x, ma, mb)

    @staticmethod
    def backward(ctx, y):
        g, x = ctx.saved_tensors
        return (x @ y.t() if ctx.needs_input_grad[0] else None), (g @ y if ctx.needs_input_grad[1] else None), None, None


class FShim:
    def __init__(self, real):
        self._real = real

    def __getattr__(self, k):
        return getattr(self._real, k)

    def linear(self, x, W, b=None):
        net = getattr(W, "_net", None)
        if net is None:
            return self._real.linear(x, W, b)
        y = Y.apply(x, W, net, "primal" if net == "sdf" else "head")
        return y if b is None else y + b


def install():
    real_wn = O.wn_weight

    def wn(p, prefix):
        W = real_wn(p, prefix)
        W._net = "sdf" if prefix.startswith("implicit") else "head"
        return W
    O.wn_weight = wn
    O.F = FShim(O.F)
    real_sp = O._softplus100

    def sp(x):
        h = real_sp(x)
        m = CFG.get("store", "f32")
        if m == "f32":
            return h
        hq = sum(split(h, m))
        return h + (hq - h).detach()
    O._softplus100 = sp


def run(sd, sc, z, eik_idx, eik_uniform):
    PHASE["v"] = 1
    p = O.params_from_numpy(sd, requires_grad=True)
    wf = WireframeGraph(T(sc["wf_vertices"]), T(sc["wf_vconf"]), T(sc["wf_edges"]), T(sc["wf_weights"]), 512, 512)
    ref = O.full_forward(p, {k: T(sc[k]) for k in ("intrinsics", "pose", "uv", "uv_proj")}, wf.line_segments(), wf.vertices,
                         training=True, rand={"eik_idx": eik_idx, "eik_uniform": eik_uniform}, z_vals=z)
    lo = O.neat_loss(ref, T(sc["gt_rgb"]), T(sc["gt_lines2d"]))
    PHASE["v"] = 2
    lo["loss"].backward()
    return p, ref, lo


def scheme(fwd, bwd, wg, store=None, head_fwd=None, head_bwd=None, head_wg=None):
    """fwd / bwd / wg = (activation mode, weight mode) of the forward chains (primal, adjoint), the backward chains (tangent,
    reverse) and the weight gradients (both operands are activations)"""
    head_fwd, head_bwd, head_wg = head_fwd or fwd, head_bwd or bwd, head_wg or wg
    c = {("sdf", "primal"): fwd, ("sdf", "adjoint"): fwd, ("sdf", "tangent"): bwd, ("sdf", "reverse"): bwd,
         ("sdf", "wgrad1"): wg, ("sdf", "wgrad2"): wg,
         ("head", "head"): head_fwd, ("head", "head_dx"): head_bwd, ("head", "head_dw"): head_wg}
    c["store"] = store or fwd[0]
    return c


SCHEMES = {
    "fp32": scheme(("f32", "f32"), ("f32", "f32"), ("f32", "f32")),
    "bf16x3 all": scheme(("bf16x2", "bf16x2"), ("bf16x2", "bf16x2"), ("bf16x2", "bf16x2")),
    "x3 chains, bf16 wgrad": scheme(("bf16x2", "bf16x2"), ("bf16x2", "bf16x2"), ("bf16", "bf16")),
    "x3 fwd, bf16 bwd+wgrad": scheme(("bf16x2", "bf16x2"), ("bf16", "bf16"), ("bf16", "bf16")),
    "x3 chains, f16s wgrad": scheme(("bf16x2", "bf16x2"), ("bf16x2", "bf16x2"), ("f16s", "f16s")),
    "x3 fwd, f16s bwd+wgrad": scheme(("bf16x2", "bf16x2"), ("f16s", "f16"), ("f16s", "f16s")),
    "x3 fwd, f16s bwd, x3 wgrad": scheme(("bf16x2", "bf16x2"), ("f16s", "f16"), ("bf16x2", "bf16x2")),
    "x3 fwd, f16s bwd w/ f16x2 W, f16s wgrad": scheme(("bf16x2", "bf16x2"), ("f16s", "f16x2"), ("f16s", "f16s")),
}


def main():
    R, S = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (96, 128)
    seed = 1
    sd = synth.synth_state_dict(seed, "rough")
    sc = synth.synth_scene(seed=seed, n_rays=R, view=seed)
    z = T(synth.synth_z_vals(seed, R, S))
    gen = torch.Generator().manual_seed(seed)
    eik_idx = torch.randint(S, (R,), generator=gen)
    eik_uniform = torch.empty(R, 3).uniform_(-3, 3, generator=gen)
    p0, ref0, lo0 = run(sd, sc, z, eik_idx, eik_uniform)          # plain oracle (before install)
    install()
    keys = ("rgb_values", "lines3d", "depth", "xyz", "sdf", "grad_theta", "l3d", "lines2d_calib")
    for name, cfg in SCHEMES.items():
        CFG.clear()
        CFG.update(cfg)
        p, ref, lo = run(sd, sc, z, eik_idx, eik_uniform)
        outs = {k: float((ref[k].detach() - ref0[k].detach()).abs().max() / max(1.0, float(ref0[k].abs().max()))) for k in keys}
        worst, wname, worst_l2, l2name = 0.0, "", 0.0, ""
        for k in p:
            r = p0[k].grad
            if r is None or p[k].grad is None:
                continue
            e = float((p[k].grad - r).abs().max() / max(float(r.abs().max()), 1e-6))
            l2 = float((p[k].grad - r).norm() / (r.norm() + 1e-30))
            if e > worst:
                worst, wname = e, k
            if r.numel() > 1 and l2 > worst_l2:
                worst_l2, l2name = l2, k
        print(f"{name:28s} out max {max(outs.values()):.1e} ({max(outs, key=outs.get)}) | loss {abs(float(lo['loss']) - float(lo0['loss'])):.1e} | "
              f"grad max/max {worst:.1e} ({wname}) relL2 {worst_l2:.1e} ({l2name})", flush=True)
        print("     ", {k: f"{v:.1e}" for k, v in outs.items()}, flush=True)


if __name__ == "__main__":
    main()
