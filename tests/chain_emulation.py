"""Host-side (torch, CPU) emulation of the exact per-layer algebra the HIP kernels implement for the
SDF MLP: primal chain, adjoint chain (normals), tangent chain + primal reverse (double backward).
Used by tests to validate the derivation in DESIGN.md against autograd; not part of the product."""
import math

import torch

SQ = 1.0 / math.sqrt(2.0)


def pe_rows(x, L=6):
    parts = [x]
    for k in range(L):
        f = 2.0 ** k
        parts += [torch.sin(f * x), torch.cos(f * x)]
    return torch.cat(parts, -1)            # [P, 3+6L]


def pe_jac_apply(x, v, L=6):
    """J v : tangent of PE along v [P,3] -> [P,39]."""
    parts = [v]
    for k in range(L):
        f = 2.0 ** k
        parts += [f * torch.cos(f * x) * v, -f * torch.sin(f * x) * v]
    return torch.cat(parts, -1)


def pe_jac_t_apply(x, e, L=6):
    """J^T e : [P,39] -> [P,3]."""
    g = e[:, :3].clone()
    for k in range(L):
        f = 2.0 ** k
        g = g + f * torch.cos(f * x) * e[:, 3 + 6 * k:6 + 6 * k] - f * torch.sin(f * x) * e[:, 6 + 6 * k:9 + 6 * k]
    return g


def dphi(h):      # softplus'(a) from h = softplus_100(a):  sigmoid(100 a) = 1 - exp(-100 h)
    return -torch.expm1(-100.0 * h)


def forward(W, b, x):
    """W[l]: effective weights [out,in] (layer 4 takes cat(h4, E)/sqrt2).  returns h[1..8], out[P,257], E."""
    E = pe_rows(x)
    h = [E]
    for l in range(9):
        inp = torch.cat([h[l], E], 1) * SQ if l == 4 else h[l]
        a = inp @ W[l].t() + b[l]
        if l < 8:
            h.append(torch.nn.functional.softplus(a, beta=100.0))
        else:
            out = a
    return h, out, E


def adjoint(W, h, x):
    """u[l] = d sdf_raw / d a_l  for l=0..7  ; g = d sdf_raw/dx."""
    P = x.shape[0]
    u = [None] * 8
    vh = W[8][0:1, :].expand(P, 256)                 # adjoint of h_8
    e_skip = None
    for l in range(7, -1, -1):
        u[l] = vh * dphi(h[l + 1])
        v = u[l] @ W[l]                              # adjoint of in_l
        if l == 4:
            vh, e_skip = v[:, :217] * SQ, v[:, 217:] * SQ
        elif l == 0:
            e0 = v
        else:
            vh = v
    return u, pe_jac_t_apply(x, e0 + e_skip)


def backward(W, h, u, x, E, out_bar, g_bar):
    """Given cotangents of out[P,257] and of g[P,3] -> dW[l], db[l] (effective weights)."""
    P = x.shape[0]
    Eh = pe_jac_apply(x, g_bar)                       # tangent seed
    vhat = [None] * 9                                 # tangent of in_l (pre 1/sqrt2 concat handled inline)
    m = [None] * 8
    th = Eh                                           # tangent of h_l
    for l in range(8):
        vin = torch.cat([th, Eh], 1) * SQ if l == 4 else th
        vhat[l] = vin
        t = vin @ W[l].t()                            # tangent of a_l
        s = dphi(h[l + 1])
        th = t * s                                    # tangent of h_{l+1}
        m[l] = t * u[l] * 100.0 * (1.0 - s)           # extra cotangent of a_l (phi''/phi' = 100 (1-s))
    vhat[8] = th
    dW, db = [None] * 9, [None] * 9
    a_bar = out_bar
    for l in range(8, -1, -1):
        inp = torch.cat([h[l], E], 1) * SQ if l == 4 else h[l]
        dW[l] = a_bar.t() @ inp
        db[l] = a_bar.sum(0)
        if l == 8:
            dW[l][0] += vhat[8].sum(0)                # u_8 = e_0
        else:
            dW[l] += u[l].t() @ vhat[l]
        if l > 0:
            hb = a_bar @ W[l]
            if l == 4:
                hb = hb[:, :217] * SQ
            a_bar = hb * dphi(h[l]) + m[l - 1]
    return dW, db
