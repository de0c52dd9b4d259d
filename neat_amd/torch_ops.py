"""torch.ops.neat_hip.* -- the hot path's C-ABI entry points registered with the PyTorch dispatcher.

north_star asks for the inner loop "exposed as torch ops" and SURVEY §8(b) lists a `TORCH_LIBRARY(neat_hip, m)` registration as the
native boundary beneath the module API.  The boundary proper stays the plain C ABI (include/neat_hip.h: nothing torch-typed crosses
it); this module is the dispatcher-side binding of that ABI for callers who want `torch.ops.neat_hip.render_rays(...)` instead of
`neat_amd.ops.render_rays(handle, ...)`: schemas of tensors and scalars only (the 57 parameters travel as a `Tensor[]`), autograd
registered per op, and the CUDA dispatch key ONLY -- a CPU tensor reaches no kernel and the dispatcher raises NotImplementedError
(there is no CPU path; the oracle is test infrastructure).

The ops are thin on purpose: argument checks, workspace sizing and the launches are those of `neat_amd.ops`, so both bindings run
the same kernels and give the same bits (tests/test_gpu_parity.py::test_torch_ops_*).  `neat_amd.networks` keeps calling
`neat_amd.ops` directly: one Python frame less per call, and its per-model `NetHandle` owns the packed-weight cache.

    import neat_amd.torch_ops                       # registers the library
    params = neat_amd.torch_ops.net_params(model)   # [weight_v, weight_g, bias] x 19 layers, the C ABI's order
    rgb, lines3d, depth, xyz, eik_grad, weights, sdf, points, nmap = torch.ops.neat_hip.render_rays(
        origins, dirs, z, beta, params, eik_points, radius, 20.0, precision, False)
"""
from typing import List, Optional, Tuple

import torch
from torch import Tensor
from torch.library import custom_op, register_autograd

from . import _lib, ops

_HANDLES = {}


def _handle(params: List[Tensor], precision: int) -> ops.NetHandle:
    """NetHandle (packed-weight cache, re-packed when a parameter's version changes) for this list of parameter tensors."""
    n = len(params) // 3
    if len(params) != 3 * n or n not in (ops.N_SDF, _lib.NUM_LAYERS):
        raise RuntimeError("neat_hip: params must hold (weight_v, weight_g, bias) of the 9 SDF layers or of all 19 layers")
    key = (int(precision),) + tuple(t.data_ptr() for t in params)
    h = _HANDLES.get(key)
    if h is None:
        if len(_HANDLES) >= 8:
            _HANDLES.pop(next(iter(_HANDLES)))
        h = _HANDLES[key] = ops.NetHandle()
        h.precision = int(precision)
    h.set_layers(0, [tuple(params[3 * i:3 * i + 3]) for i in range(n)])
    return h


def net_params(model) -> List[Tensor]:
    """The 57 parameters of a VolSDFNetwork-shaped module in the C ABI's layer order."""
    out = []
    for name, count in ops.NET_LAYOUT:
        net = getattr(model, name)
        for l in range(count):
            lin = getattr(net, f"lin{l}")
            out += [lin.weight_v, lin.weight_g, lin.bias]
    return out


# ---- rays (rend_util.get_camera_params, reference utils/rend_util.py:55-81,95-108) ---------------------------------------------------
@custom_op("neat_hip::camera_rays", mutates_args=(), device_types="cuda")
def camera_rays(uv: Tensor, pose: Tensor, intrinsics: Tensor) -> Tuple[Tensor, Tensor]:
    dirs, _, origins = ops.camera_rays(uv, pose, intrinsics, with_origins=True)
    return dirs, origins


# ---- SDF network (neat_wfr_rend_a.py:78-137) --------------------------------------------------------------------------------------------
@custom_op("neat_hip::sdf_values", mutates_args=(), device_types="cuda")
def sdf_values(x: Tensor, params: List[Tensor], radius: float, scale: float, precision: int) -> Tensor:
    return ops.sdf_values(_handle(params, precision), x, radius, scale)


@custom_op("neat_hip::sdf_outputs", mutates_args=(), device_types="cuda")
def sdf_outputs(x: Tensor, params: List[Tensor], radius: float, scale: float,
                precision: int) -> Tuple[Tensor, Tensor, Tensor, Tensor, Tensor]:
    """-> forward()[P,257], clamped sdf [P,1], feature [P,256], d sdf/dx [P,3], and the workspace its backward reads."""
    lib = _lib.lib()
    h = _handle(params, precision)
    x = ops._f32c(x)
    P = x.shape[0]
    packed, netp = h.packed()
    dev = x.device
    ws = torch.empty(lib.neat_sdf_ws_floats(P, 1, h.precision), device=dev, dtype=torch.float32)
    out, sdf = torch.empty(P, 257, device=dev), torch.empty(P, 1, device=dev)
    feat, grad = torch.empty(P, 256, device=dev), torch.empty(P, 3, device=dev)
    if P:
        _lib.check(lib.neat_sdf_forward(ops._p(packed), ops.ctypes.byref(netp), ops._p(x), P, 1, h.precision, float(radius), float(scale),
                                        ops._p(ws), ops._p(out), ops._p(sdf), ops._p(feat), ops._p(grad), ops._stream()), "neat_sdf_forward")
    return out, sdf, feat, grad, ws


@custom_op("neat_hip::sdf_outputs_backward", mutates_args=(), device_types="cuda")
def sdf_outputs_backward(ws: Tensor, params: List[Tensor], P: int, precision: int, d_out: Optional[Tensor], d_sdf: Optional[Tensor],
                         d_feat: Optional[Tensor], d_grad: Optional[Tensor]) -> List[Tensor]:
    lib = _lib.lib()
    h = _handle(params, precision)
    packed, netp = h.packed()
    gr, views, _ = ops._grad_buffers(h, 0, ops.N_SDF, ws.device)
    d_out, d_sdf, d_feat, d_grad = (ops._f32c(t) for t in (d_out, d_sdf, d_feat, d_grad))
    _lib.check(lib.neat_sdf_backward(ops._p(packed), ops.ctypes.byref(netp), ops._p(ws), P, h.precision, ops._p(d_out), ops._p(d_sdf),
                                     ops._p(d_feat), ops._p(d_grad), ops.ctypes.byref(gr), ops._stream()), "neat_sdf_backward")
    return [v.clone() for v in views]       # custom-op outputs may not be views of one another: 27 small copies of one flat buffer


def _sdf_setup(ctx, inputs, output):
    x, params, radius, scale, precision = inputs
    ctx.set_materialize_grads(False)
    ctx.ws, ctx.params, ctx.P, ctx.precision = output[4], params, x.shape[0], precision


def _sdf_backward(ctx, d_out, d_sdf, d_feat, d_grad, _d_ws):
    grads = torch.ops.neat_hip.sdf_outputs_backward(ctx.ws, ctx.params, ctx.P, ctx.precision, d_out, d_sdf, d_feat, d_grad)
    ctx.ws = None
    return None, grads, None, None, None


register_autograd("neat_hip::sdf_outputs", _sdf_backward, setup_context=_sdf_setup)


# ---- main pass (neat_wfr_rend_a.py:392-422,515-536) ----------------------------------------------------------------------------------
@custom_op("neat_hip::render_rays", mutates_args=(), device_types="cuda")
def render_rays(origins: Tensor, dirs: Tensor, z: Tensor, beta: Tensor, params: List[Tensor], eik_points: Optional[Tensor], radius: float,
                scale: float, precision: int,
                want_normal_map: bool) -> Tuple[Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor]:
    """-> rgb [R,3], lines3d [R,2,3], depth [R], xyz [R,3], eik_grad [E,3], weights, sdf, points, normal_map, workspace."""
    lib = _lib.lib()
    h = _handle(params, precision)
    origins, dirs, z = (ops._f32c(t) for t in (origins, dirs, z))
    beta_d = ops._f32c(beta.reshape(1))
    R, S = z.shape
    dev = z.device
    eik = ops._f32c(eik_points) if eik_points is not None and eik_points.shape[0] else None
    E = 0 if eik is None else eik.shape[0]
    packed, netp = h.packed()
    prec = h.precision
    ws = torch.empty(lib.neat_render_ws_floats(R, S, E, prec), device=dev, dtype=torch.float32)
    points, weights, sdf = torch.empty(R, S, 3, device=dev), torch.empty(R, S, device=dev), torch.empty(R, S, device=dev)
    rgb, lines3d = torch.empty(R, 3, device=dev), torch.empty(R, 2, 3, device=dev)
    depth, xyz = torch.empty(R, device=dev), torch.empty(R, 3, device=dev)
    nmap = torch.empty(R, 3, device=dev) if want_normal_map else None
    eik_grad = torch.empty(E, 3, device=dev)
    _lib.check(lib.neat_render_forward(ops._p(packed), ops.ctypes.byref(netp), ops._p(origins), ops._p(dirs), ops._p(z), R, S, prec,
                                       ops._p(beta_d), 0.0, float(radius), float(scale), ops._p(ws), ops._p(points), ops._p(weights), ops._p(sdf),
                                       ops._p(rgb), ops._p(lines3d), ops._p(depth), ops._p(xyz), ops._p(nmap), ops._p(eik), E,
                                       ops._p(eik_grad) if E else None, ops._stream()), "neat_render_forward")
    if nmap is None:
        nmap = torch.empty(0, device=dev)
    return rgb, lines3d, depth, xyz, eik_grad, weights, sdf, points, nmap, ws


@custom_op("neat_hip::render_rays_backward", mutates_args=(), device_types="cuda")
def render_rays_backward(ws: Tensor, dirs: Tensor, z: Tensor, beta: Tensor, params: List[Tensor], E: int, precision: int,
                         d_rgb: Optional[Tensor], d_lines3d: Optional[Tensor], d_depth: Optional[Tensor], d_xyz: Optional[Tensor],
                         d_eik: Optional[Tensor]) -> List[Tensor]:
    """-> [d beta] + the 57 parameter gradients."""
    lib = _lib.lib()
    h = _handle(params, precision)
    packed, netp = h.packed()
    R, S = z.shape
    dev = ws.device
    gr, views, _ = ops._grad_buffers(h, 0, _lib.NUM_LAYERS, dev)
    dirs, z = ops._f32c(dirs), ops._f32c(z)
    beta_d = ops._f32c(beta.reshape(1))
    d_rgb, d_lines3d, d_depth, d_xyz, d_eik = (ops._f32c(t) for t in (d_rgb, d_lines3d, d_depth, d_xyz, d_eik))
    dbeta_ray, dbeta = torch.empty(R, device=dev), torch.empty(1, device=dev)
    _lib.check(lib.neat_render_backward(ops._p(packed), ops.ctypes.byref(netp), ops._p(ws), ops._p(dirs), ops._p(z), R, S, E, h.precision,
                                        ops._p(beta_d), 0.0, ops._p(d_rgb), ops._p(d_lines3d), ops._p(d_depth), ops._p(d_xyz),
                                        ops._p(d_eik) if E else None, None, ops.ctypes.byref(gr), ops._p(dbeta_ray), ops._p(dbeta), ops._stream()),
               "neat_render_backward")
    return [dbeta.reshape(beta.shape)] + [v.clone() for v in views]


def _render_setup(ctx, inputs, output):
    origins, dirs, z, beta, params, eik_points, radius, scale, precision, want_normal_map = inputs
    ctx.set_materialize_grads(False)
    ctx.ws, ctx.dirs, ctx.z, ctx.beta, ctx.params, ctx.precision = output[9], dirs, z, beta.detach(), params, precision
    ctx.E = output[4].shape[0]


def _render_backward(ctx, d_rgb, d_lines3d, d_depth, d_xyz, d_eik, *_unused):
    g = torch.ops.neat_hip.render_rays_backward(ctx.ws, ctx.dirs, ctx.z, ctx.beta, ctx.params, ctx.E, ctx.precision,
                                                d_rgb, d_lines3d, d_depth, d_xyz, d_eik if ctx.E else None)
    ctx.ws = None
    return None, None, None, g[0], list(g[1:]), None, None, None, None, None


register_autograd("neat_hip::render_rays", _render_backward, setup_context=_render_setup)


@custom_op("neat_hip::render_rays_eval", mutates_args=(), device_types="cuda")
def render_rays_eval(origins: Tensor, dirs: Tensor, z: Tensor, beta: Tensor, params: List[Tensor], radius: float, scale: float,
                     precision: int) -> Tuple[Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor]:
    """Forward-only main pass (nothing saved): rgb, lines3d, depth, xyz, weights, sdf, points."""
    rgb, lines3d, depth, xyz, _, weights, sdf, points, _ = ops.render_rays_eval(_handle(params, precision), origins, dirs, z, beta,
                                                                                radius, scale)
    return rgb, lines3d, depth, xyz, weights, sdf, points


# ---- compositing weights, samplers, matching (rend_a :540-554; ray_sampler.py:16-59; rend_a :473) -------------------------------------
@custom_op("neat_hip::volume_weights", mutates_args=(), device_types="cuda")
def volume_weights(z: Tensor, sdf: Tensor, beta: Tensor) -> Tensor:
    return ops.volume_weights(z, sdf, beta)


@custom_op("neat_hip::sample_pdf", mutates_args=(), device_types="cuda")
def sample_pdf(bins: Tensor, weights: Tensor, u: Tensor) -> Tensor:
    return ops.sample_pdf(bins, weights, u)[0]


@custom_op("neat_hip::linear_sum_assignment", mutates_args=(), device_types="cuda")
def linear_sum_assignment(cost: Tensor) -> Tuple[Tensor, Tensor, Tensor]:
    """-> row_ind, col_ind (int64, padded with -1), number of matches (int32 [1]); nothing leaves the device."""
    return ops.linear_sum_assignment(cost)


OPS = ("camera_rays", "sdf_values", "sdf_outputs", "sdf_outputs_backward", "render_rays", "render_rays_backward", "render_rays_eval",
       "volume_weights", "sample_pdf", "linear_sum_assignment")
