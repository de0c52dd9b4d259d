"""torch ops over libneat_hip.so: autograd Functions whose forward/backward are the HIP kernels.

PyTorch is plumbing here (device memory, current stream, autograd graph); all P-sized arithmetic
happens in neat_amd/csrc.  Every op requires CUDA float32 tensors and raises if the library is absent.
"""
import ctypes
import warnings

import torch

from . import _lib

# layer order of the C ABI (include/neat_hip.h)
NET_LAYOUT = (("implicit_network", 9), ("rendering_network", 5), ("attraction_network", 5))
LAYER_OUT = [256, 256, 256, 217, 256, 256, 256, 256, 257, 256, 256, 256, 256, 3, 256, 256, 256, 256, 6]
LAYER_IN = [39, 256, 256, 256, 256, 256, 256, 256, 256, 289, 256, 256, 256, 256, 265, 256, 256, 256, 256]
N_SDF = 9


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def copy_batch(pairs):
    """[(dst, src), ...] -> one launch per 16 copies (neat_copy_batch): dst device tensors, src device or PINNED host tensors of the same
    byte size, both contiguous.  The step prefix of a replayed graph: the fresh batch and the CPU-drawn randoms go into the captured
    tensors without a multi-tensor ATen launch and without copy-engine transfers (each of which idles the GPU for ~6 us before the next kernel)."""
    lib = _lib.lib()
    for i0 in range(0, len(pairs), 16):
        chunk = pairs[i0:i0 + 16]
        n = len(chunk)
        src = (ctypes.c_void_p * n)(*[s.data_ptr() for _, s in chunk])
        dst = (ctypes.c_void_p * n)(*[d.data_ptr() for d, _ in chunk])
        nb = (ctypes.c_longlong * n)(*[d.numel() * d.element_size() for d, _ in chunk])
        _lib.check(lib.neat_copy_batch(src, dst, nb, n, _stream()), "neat_copy_batch")


def copy_batch_ok(dst, src):
    """Can (dst, src) ride in copy_batch?  Same dtype and element count, contiguous, 4-byte granular; src on dst's device or pinned."""
    return (dst.dtype == src.dtype and dst.numel() == src.numel() and dst.is_contiguous() and src.is_contiguous()
            and (dst.numel() * dst.element_size()) % 4 == 0 and dst.data_ptr() % 4 == 0 and src.data_ptr() % 4 == 0
            and dst.numel() > 0 and (src.device == dst.device or (src.device.type == "cpu" and src.is_pinned())))


def _f32c(t):
    if t is None:
        return None
    if t.dtype != torch.float32 or not t.is_cuda:
        raise RuntimeError("neat_amd ops need CUDA float32 tensors (no CPU path): got %s on %s" % (t.dtype, t.device))
    return t.contiguous()


def _as_u8(mask):
    """bool / uint8 mask -> contiguous uint8 (a bool tensor is reinterpreted, not copied)."""
    if mask is None:
        return None
    mask = mask.contiguous()
    return mask.view(torch.uint8) if mask.dtype == torch.bool else mask.to(torch.uint8)


class NetHandle:
    """The 19 weight-normed layers of one model (heads may be absent) + a cache of the packed weights."""

    def __init__(self):
        self.layers = [None] * _lib.NUM_LAYERS      # (weight_v, weight_g, bias) Parameters
        self.precision = 0                          # _lib.PRECISIONS: 0 = fp32 (parity build), 1 = bf16 MFMA (throughput build), 2 = bf16x3, 3 = f16 MFMA, 4 = f16x3 forward + f16 backward
        self._key = None
        self._packed = None
        self._netp = None

    def set_layers(self, first, triples):
        for i, t in enumerate(triples):
            self.layers[first + i] = t

    def tensors(self, first=0, count=_lib.NUM_LAYERS):
        out = []
        for l in range(first, first + count):
            if self.layers[l] is None:
                raise RuntimeError(f"layer {l} is not attached to this NetHandle")
            out.extend(self.layers[l])
        return out

    def has_heads(self):
        return all(l is not None for l in self.layers)

    def _fill_netp(self):
        netp = _lib.NetParams()
        dev = None
        for l, tr in enumerate(self.layers):
            if tr is None:
                continue
            v, g, b = (x.detach() for x in tr)
            if tuple(v.shape) != (LAYER_OUT[l], LAYER_IN[l]) or g.numel() != LAYER_OUT[l] or b.numel() != LAYER_OUT[l]:
                raise RuntimeError(f"layer {l}: unsupported shape {tuple(v.shape)} (the HIP path implements the "
                                   "architecture of the shipped confs: 8x256 SDF MLP, PE-6/PE-4, 4x256 heads)")
            for t in (v, g, b):
                if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
                    raise RuntimeError("parameters must be contiguous CUDA float32 (call model.cuda())")
            netp.v[l], netp.g[l], netp.b[l] = v.data_ptr(), g.data_ptr(), b.data_ptr()
            dev = v.device
        return netp, dev

    def packed(self):
        """Returns (packed weights tensor, NetParams).  Re-packs (2 kernel launches) whenever a parameter changed."""
        present = [t for l in self.layers if l is not None for t in l]
        # a tensor DERIVED from a parameter per forward (heads with mode = 'nerf', inside_out: neat_amd.networks) is keyed by its source
        # parameter: a fresh tensor may land on the address a previous one had, with version 0 again
        ptrs = tuple(t.data_ptr() for t in present)
        key = (self.precision,) + tuple(getattr(t, "_neat_key", None) or (p_, t._version) for t, p_ in zip(present, ptrs))
        if key != self._key:
            lib = _lib.lib()
            netp, dev = self._fill_netp()
            packed = torch.empty(lib.neat_packed_floats(self.precision), device=dev, dtype=torch.float32)
            _lib.check(lib.neat_pack_weights(ctypes.byref(netp), _p(packed), self.precision, _stream()), "neat_pack_weights")
            self._key, self._packed, self._netp, self._ptrs = key, packed, netp, ptrs
        elif ptrs != getattr(self, "_ptrs", None):
            # same values (same source parameters), other tensors: the kernels also read biases, gains and weight_v through NetParams'
            # raw pointers -- they must follow the derived tensors that are alive NOW, not the ones the pack was made from
            self._netp, _ = self._fill_netp()
            self._ptrs = ptrs
        return self._packed, self._netp


def _grad_buffers(handle, first, count, device):
    """One flat buffer with (dv, dg, db) views per layer -> (NetGrads struct, [views in v,g,b order], flat)."""
    sizes, shapes = [], []
    for l in range(first, first + count):
        v, g, b = handle.layers[l]
        sizes += [LAYER_OUT[l] * LAYER_IN[l], LAYER_OUT[l], LAYER_OUT[l]]
        shapes += [v.shape, g.shape, b.shape]
    flat = torch.empty(sum(sizes), device=device, dtype=torch.float32)
    views = [t.view(sh) for t, sh in zip(flat.split(sizes), shapes)]
    gr = _lib.NetGrads()
    base, off = flat.data_ptr(), 0
    for i, l in enumerate(range(first, first + count)):
        gr.dv[l] = base + 4 * off; off += sizes[3 * i]
        gr.dg[l] = base + 4 * off; off += sizes[3 * i + 1]
        gr.db[l] = base + 4 * off; off += sizes[3 * i + 2]
    return gr, views, flat


class SdfOutputsFn(torch.autograd.Function):
    """ImplicitNetwork.forward / get_outputs / gradient in one op.
    returns (forward()[P,257], clamped sdf [P,1], feature [P,256], d sdf/dx [P,3]); differentiable wrt the
    27 SDF parameters (including the double backward through d sdf/dx); x is treated as a constant."""

    @staticmethod
    def forward(ctx, handle, x, radius, scale, *params):
        lib = _lib.lib()
        ctx.set_materialize_grads(False)
        if x.requires_grad and torch.is_grad_enabled():
            # the reference's get_outputs is differentiable in x (rend_a :429: points3d = sum(w p) carries the weights' graph);
            # loss_wfr never uses that path, and this op does not implement d/dx -- say so instead of dropping it silently
            warnings.warn("neat_amd: sdf_outputs treats its input points as constants; gradients do not flow through x "
                          "(INTEGRATION.md, 'Restrictions')", stacklevel=3)
        x = _f32c(x.detach())
        P = x.shape[0]
        packed, netp = handle.packed()
        prec = handle.precision
        ws = torch.empty(lib.neat_sdf_ws_floats(P, 1, prec), device=x.device, dtype=torch.float32)
        out = torch.empty(P, 257, device=x.device)
        sdf = torch.empty(P, 1, device=x.device)
        feat = torch.empty(P, 256, device=x.device)
        grad = torch.empty(P, 3, device=x.device)
        _lib.check(lib.neat_sdf_forward(_p(packed), ctypes.byref(netp), _p(x), P, 1, prec, float(radius), float(scale), _p(ws),
                                        _p(out), _p(sdf), _p(feat), _p(grad), _stream()), "neat_sdf_forward")
        ctx.handle, ctx.P, ctx.ws, ctx.packed, ctx.netp, ctx.prec = handle, P, ws, packed, netp, prec
        ctx.keep = params          # netp holds raw pointers: tensors derived per forward (networks: nerf heads, inside_out) must outlive the backward
        return out, sdf, feat, grad

    @staticmethod
    def backward(ctx, d_out, d_sdf, d_feat, d_grad):
        lib = _lib.lib()
        h = ctx.handle
        gr, views, _ = _grad_buffers(h, 0, N_SDF, ctx.ws.device)
        d_out, d_sdf, d_feat, d_grad = (_f32c(t) for t in (d_out, d_sdf, d_feat, d_grad))
        _lib.check(lib.neat_sdf_backward(_p(ctx.packed), ctypes.byref(ctx.netp), _p(ctx.ws), ctx.P, ctx.prec, _p(d_out), _p(d_sdf),
                                         _p(d_feat), _p(d_grad), ctypes.byref(gr), _stream()), "neat_sdf_backward")
        ctx.ws = None
        return (None, None, None, None, *views)


def sdf_outputs(handle, x, radius, scale):
    if x.shape[0] == 0:
        z = x.new_zeros
        return z(0, 257), z(0, 1), z(0, 256), z(0, 3)
    return SdfOutputsFn.apply(handle, x, radius, scale, *handle.tensors(0, N_SDF))


def sdf_values(handle, x, radius, scale, gate=None, fast=False):
    """get_sdf_vals without autograd (sampler path): primal chain only, nothing saved.
    gate = (int32 tensor, index, value): the launch does nothing unless tensor[index] == value on the device (sync-free sampler).
    fast (fp16x3 only): the one-product f16 chain on the fp16x3 pack (NEAT_F16X3_FASTVALUES)."""
    lib = _lib.lib()
    x = _f32c(x.detach())
    P = x.shape[0]
    sdf = torch.empty(P, 1, device=x.device)
    if P == 0:
        return sdf
    packed, netp = handle.packed()
    prec = 5 if (fast and handle.precision == 4) else handle.precision
    ws = torch.empty(lib.neat_sdf_ws_floats(P, 0, prec), device=x.device, dtype=torch.float32)
    if gate is not None:
        ctl, idx, val = gate
        _lib.check(lib.neat_sdf_values_gated(_p(packed), ctypes.byref(netp), _p(x), P, prec, float(radius), float(scale), _p(ws),
                                             _p(sdf), ctypes.c_void_p(ctl.data_ptr() + 4 * idx), int(val), _stream()),
                   "neat_sdf_values_gated")
        return sdf
    _lib.check(lib.neat_sdf_forward(_p(packed), ctypes.byref(netp), _p(x), P, 0, prec, float(radius), float(scale), _p(ws),
                                    None, _p(sdf), None, None, _stream()), "neat_sdf_forward(values)")
    return sdf


def sdf_values_rays(handle, origins, dirs, z, radius, scale, gate=None, fast=False):
    """sdf_values at the points origins + z dirs of R rays x S depths -> [R S, 1]; the points are formed by the launch that lays them
    out for the SDF kernels (no addcmul, no layout change: two launches less per sampler round)."""
    lib = _lib.lib()
    origins, dirs, z = (_f32c(t.detach()) for t in (origins, dirs, z))
    R, S = z.shape
    sdf = torch.empty(R * S, 1, device=z.device)
    if R * S == 0:
        return sdf
    packed, netp = handle.packed()
    prec = 5 if (fast and handle.precision == 4) else handle.precision
    ws = torch.empty(lib.neat_sdf_ws_floats(R * S, 0, prec), device=z.device, dtype=torch.float32)
    gptr, gval = (None, 0) if gate is None else (ctypes.c_void_p(gate[0].data_ptr() + 4 * gate[1]), int(gate[2]))
    _lib.check(lib.neat_sdf_values_rays(_p(packed), ctypes.byref(netp), _p(origins), _p(dirs), _p(z), R, S, prec, float(radius), float(scale),
                                        _p(ws), _p(sdf), gptr, gval, _stream()), "neat_sdf_values_rays")
    return sdf


def sampler_init(z, beta_param, beta_min, beta_c, nctl):
    """What Algorithm 1 computes before its first round (ray_sampler.py:131-143) in one launch -> (beta0 [1] = |beta_param| + beta_min,
    per-ray beta [R] = sqrt(beta_c sum gap^2), ctl int32 [nctl] zeroed)."""
    z = _f32c(z.detach())
    R, n = z.shape
    dev = z.device
    beta0, beta, ctl = torch.empty(1, device=dev), torch.empty(R, device=dev), torch.empty(nctl, dtype=torch.int32, device=dev)
    _lib.check(_lib.lib().neat_sampler_init(_p(z), R, n, _p(_f32c(beta_param.detach().reshape(1))), float(beta_min), float(beta_c), _p(beta0),
                                            _p(beta), _p(ctl), nctl, _stream()), "neat_sampler_init")
    return beta0, beta, ctl


def heads_forward(handle, points, normals, view_dirs, feats):
    """rgb [P,3] and line endpoints [P,2,3] of the two heads on given inputs (forward only)."""
    lib = _lib.lib()
    points, normals, view_dirs, feats = (_f32c(t.detach()) for t in (points, normals, view_dirs, feats))
    P = points.shape[0]
    packed, netp = handle.packed()
    ws = torch.empty(lib.neat_heads_ws_floats(P, handle.precision), device=points.device, dtype=torch.float32)
    rgb = torch.empty(P, 3, device=points.device)
    lines = torch.empty(P, 2, 3, device=points.device)
    _lib.check(lib.neat_heads_forward(_p(packed), ctypes.byref(netp), _p(points), _p(normals), _p(view_dirs), _p(feats), P,
                                      handle.precision, _p(ws), _p(rgb), _p(lines), _stream()), "neat_heads_forward")
    return rgb, lines


class RenderRaysFn(torch.autograd.Function):
    """Main pass of VolSDFNetwork.forward (rend_a :392-422): points -> SDF MLP (+normals) -> both heads ->
    Laplace density -> alpha compositing; optionally E extra points ride along through the SDF network only and
    return its raw gradient (the eikonal term, :515-527).  Differentiable wrt all 57 network parameters and beta."""

    @staticmethod
    def forward(ctx, handle, origins, dirs, z, beta, beta_min, radius, scale, want_normal_map, eik_points, bg_color, *params):
        lib = _lib.lib()
        ctx.set_materialize_grads(False)
        origins, dirs, z = (_f32c(t.detach()) for t in (origins, dirs, z))
        beta_d = _f32c(beta.detach().reshape(1))
        R, S = z.shape
        dev = z.device
        eik = _f32c(eik_points.detach()) if eik_points is not None else None
        E = 0 if eik is None else eik.shape[0]
        packed, netp = handle.packed()
        prec = handle.precision
        ws = torch.empty(lib.neat_render_ws_floats(R, S, E, prec), device=dev, dtype=torch.float32)
        points = torch.empty(R, S, 3, device=dev)
        weights = torch.empty(R, S, device=dev)
        sdf = torch.empty(R, S, device=dev)
        rgb = torch.empty(R, 3, device=dev)
        lines3d = torch.empty(R, 2, 3, device=dev)
        depth = torch.empty(R, device=dev)
        xyz = torch.empty(R, 3, device=dev)
        nmap = torch.empty(R, 3, device=dev) if want_normal_map else None
        eik_grad = torch.empty(E, 3, device=dev)
        _lib.check(lib.neat_render_forward(_p(packed), ctypes.byref(netp), _p(origins), _p(dirs), _p(z), R, S, prec, _p(beta_d), float(beta_min),
                                           float(radius), float(scale), _p(ws), _p(points), _p(weights), _p(sdf), _p(rgb),
                                           _p(lines3d), _p(depth), _p(xyz), _p(nmap), _p(eik) if E else None, E,
                                           _p(eik_grad) if E else None, _stream()), "neat_render_forward")
        ctx.handle, ctx.shape, ctx.ws, ctx.packed, ctx.netp, ctx.prec = handle, (R, S, E), ws, packed, netp, prec
        ctx.keep = params          # (see SdfOutputsFn)
        ctx.dirs, ctx.z, ctx.beta_d, ctx.beta_shape, ctx.beta_min = dirs, z, beta_d, beta.shape, float(beta_min)
        # white_bkgd (rend_a :411-413): what the weights leave of a ray is filled with the background colour; the backward pass sends
        # the cotangent of the opacity, -(d_rgb . bg), through the compositing kernel (d_acc)
        ctx.bg = None if bg_color is None else _f32c(bg_color.detach().reshape(3))
        if ctx.bg is not None:
            rgb = rgb + (1.0 - weights.sum(-1, keepdim=True)) * ctx.bg.unsqueeze(0)
        if nmap is None:
            nmap = torch.empty(0, device=dev)
        ctx.mark_non_differentiable(weights, sdf, points, nmap)
        return rgb, lines3d, depth, xyz, eik_grad, weights, sdf, points, nmap

    @staticmethod
    def backward(ctx, d_rgb, d_lines3d, d_depth, d_xyz, d_eik, *_unused):
        lib = _lib.lib()
        R, S, E = ctx.shape
        h = ctx.handle
        dev = ctx.ws.device
        gr, views, _ = _grad_buffers(h, 0, _lib.NUM_LAYERS, dev)
        d_rgb, d_lines3d, d_depth, d_xyz, d_eik = (_f32c(t) for t in (d_rgb, d_lines3d, d_depth, d_xyz, d_eik))
        dbeta_ray = torch.empty(R, device=dev)
        dbeta = torch.empty(1, device=dev)          # sgn(beta) * sum of the per-ray partials: one tiny launch inside neat_render_backward
        d_acc = None if ctx.bg is None or d_rgb is None else _f32c(-(d_rgb @ ctx.bg))
        _lib.check(lib.neat_render_backward(_p(ctx.packed), ctypes.byref(ctx.netp), _p(ctx.ws), _p(ctx.dirs), _p(ctx.z), R, S, E,
                                            ctx.prec, _p(ctx.beta_d), ctx.beta_min, _p(d_rgb), _p(d_lines3d), _p(d_depth), _p(d_xyz),
                                            _p(d_eik) if E else None, _p(d_acc), ctypes.byref(gr), _p(dbeta_ray), _p(dbeta),
                                            _stream()),
                   "neat_render_backward")
        ctx.ws = None
        return (None, None, None, None, dbeta.reshape(ctx.beta_shape), None, None, None, None, None, None, *views)


def render_rays_eval(handle, origins, dirs, z, beta, radius, scale, want_normal_map=False, beta_min=0.0):
    """Forward-only main pass (no autograd graph, no backward workspace): what eval / inference callers run
    (neat-final-parsing.py:203-218, 2048-ray chunks).  Same results as render_rays."""
    lib = _lib.lib()
    origins, dirs, z = (_f32c(t.detach()) for t in (origins, dirs, z))
    beta_d = _f32c(beta.detach().reshape(1))
    R, S = z.shape
    dev = z.device
    packed, netp = handle.packed()
    prec = handle.precision
    ws = torch.empty(lib.neat_render_eval_ws_floats(R, S, prec), device=dev, dtype=torch.float32)
    points, weights, sdf = torch.empty(R, S, 3, device=dev), torch.empty(R, S, device=dev), torch.empty(R, S, device=dev)
    rgb, lines3d = torch.empty(R, 3, device=dev), torch.empty(R, 2, 3, device=dev)
    depth, xyz = torch.empty(R, device=dev), torch.empty(R, 3, device=dev)
    nmap = torch.empty(R, 3, device=dev) if want_normal_map else None
    _lib.check(lib.neat_render_forward_eval(_p(packed), ctypes.byref(netp), _p(origins), _p(dirs), _p(z), R, S, prec, _p(beta_d), float(beta_min),
                                            float(radius), float(scale), _p(ws), _p(points), _p(weights), _p(sdf), _p(rgb),
                                            _p(lines3d), _p(depth), _p(xyz), _p(nmap), _stream()), "neat_render_forward_eval")
    if nmap is None:
        nmap = torch.empty(0, device=dev)
    return rgb, lines3d, depth, xyz, torch.empty(0, 3, device=dev), weights, sdf, points, nmap


def render_rays(handle, origins, dirs, z, beta, radius, scale, want_normal_map=False, eik_points=None, bg_color=None, beta_min=0.0):
    """-> rgb [R,3], lines3d [R,2,3], depth [R], xyz [R,3], eik_grad [E,3], weights, sdf, points, normal_map"""
    if not handle.has_heads():
        raise RuntimeError("render_rays needs the SDF network and both heads attached to the NetHandle")
    if not torch.is_grad_enabled() and eik_points is None:
        out = render_rays_eval(handle, origins, dirs, z, beta, radius, scale, want_normal_map, beta_min)
        if bg_color is not None:           # white_bkgd (rend_a :411-413)
            out = (out[0] + (1.0 - out[5].sum(-1, keepdim=True)) * bg_color.reshape(1, 3),) + tuple(out[1:])
        return out
    return RenderRaysFn.apply(handle, origins, dirs, z, beta, beta_min, radius, scale, want_normal_map, eik_points, bg_color, *handle.tensors())


def camera_rays(uv, pose, intrinsics, with_origins=False):
    """rend_util.get_camera_params for pose matrices: uv [1,R,2], pose [1,4,4], K [1,3|4,3|4] -> dirs [1,R,3], cam [1,3]
    (with_origins: also the camera centre repeated per ray, [R,3], written by the same launch)."""
    lib = _lib.lib()
    if pose.shape[0] != 1 or pose.shape[1:] != (4, 4):
        raise NotImplementedError("camera_rays: one 4x4 pose per call (the reference forward is single-view; "
                                  "quaternion poses are not used by the shipped datasets)")
    uv_c, pose_c, K_c = _f32c(uv.detach()), _f32c(pose.detach()), _f32c(intrinsics.detach())
    R = uv_c.shape[1]
    dirs = torch.empty(1, R, 3, device=uv_c.device)
    origins = torch.empty(R, 3, device=uv_c.device) if with_origins else None
    _lib.check(lib.neat_camera_rays(_p(uv_c), _p(pose_c), _p(K_c), int(K_c.shape[-1]), R, _p(dirs), _p(origins), _stream()),
               "neat_camera_rays")
    if with_origins:
        return dirs, pose_c[:, :3, 3], origins
    return dirs, pose_c[:, :3, 3]


def eik_points(uniform, origins, dirs, z_eik, extra=None, z=None, idx=None):
    """[uniform | origins + z_eik dirs | extra] -> [2R + J, 3] in one launch (no gradient: the inputs are draws and detached depths).
    z_eik None: the depth of ray r is z[r, idx[r]] (the draw's index into the ray's depths; the gather happens inside the launch)."""
    uniform, origins, dirs = (_f32c(t.detach()) for t in (uniform, origins, dirs))
    R = uniform.shape[0]
    if z_eik is not None:
        z_eik, zc, ic, S = _f32c(z_eik.detach().reshape(-1)), None, None, 0
    else:
        zc, ic = _f32c(z.detach()), idx.detach().reshape(-1).to(torch.int64).contiguous()
        S = zc.shape[1]
        if zc.shape[0] != R or ic.shape[0] != R:
            raise RuntimeError("eik_points: z [R,S] and idx [R]")
    ex = _f32c(extra.detach()) if extra is not None and extra.shape[0] > 0 else None
    J = ex.shape[0] if ex is not None else 0
    out = torch.empty(2 * R + J, 3, device=uniform.device)
    _lib.check(_lib.lib().neat_eik_points(_p(uniform), _p(origins), _p(dirs), _p(z_eik), _p(ex), R, J, _p(out), _p(zc), S, _p(ic), _stream()),
               "neat_eik_points")
    return out


def volume_weights(z, sdf, beta):
    lib = _lib.lib()
    z, sdf = _f32c(z.detach()), _f32c(sdf.detach().reshape(z.shape))
    beta_d = _f32c(beta.detach().reshape(1))
    w = torch.empty_like(z)
    _lib.check(lib.neat_volume_weights(_p(z), _p(sdf), z.shape[0], z.shape[1], _p(beta_d), _p(w), _stream()), "neat_volume_weights")
    return w


# ---------------------------------------------------------------------------------------------------------------
# a3: per-ray sampler kernels (ErrorBoundSampler bookkeeping)
# ---------------------------------------------------------------------------------------------------------------
def sampler_bound(z, sdf_old, sdf_new, order, beta_in, beta0, eps, iters, flag):
    """-> (merged sdf [R,n], new per-ray beta [R]); flag (int32[1]) |= any(beta > beta0)."""
    lib = _lib.lib()
    z = _f32c(z)
    R, n = z.shape
    sdf_new = _f32c(sdf_new)
    n_old = 0
    if order is not None:
        sdf_old = _f32c(sdf_old)
        n_old = sdf_old.shape[1]
        order = order.contiguous()
    sdf_out = torch.empty(R, n, device=z.device)
    beta_out = torch.empty(R, device=z.device)
    _lib.check(lib.neat_sampler_bound(_p(z), n, R, _p(sdf_old) if order is not None else None, _p(sdf_new),
                                      _p(order), n_old, _p(_f32c(beta_in)), _p(_f32c(beta0.reshape(1))), float(eps), int(iters),
                                      _p(sdf_out), _p(beta_out), _p(flag), _stream()), "neat_sampler_bound")
    return sdf_out, beta_out


def sampler_resample(z, sdf, beta, u, refine, add_tiny=0.0):
    """-> samples [R,N] (and, when refining, the sorted union [R,n+N] with its int32 gather order)."""
    lib = _lib.lib()
    z, sdf, beta, u = _f32c(z), _f32c(sdf), _f32c(beta), _f32c(u)
    R, n = z.shape
    N = u.shape[-1]
    stride = N if u.dim() == 2 else 0
    samples = torch.empty(R, N, device=z.device)
    zm = torch.empty(R, n + N, device=z.device) if refine else None
    order = torch.empty(R, n + N, device=z.device, dtype=torch.int32) if refine else None
    _lib.check(lib.neat_sampler_resample(_p(z), _p(sdf), n, R, _p(beta), int(bool(refine)), float(add_tiny), _p(u), stride, N,
                                         _p(samples), _p(zm), _p(order), _stream()), "neat_sampler_resample")
    return samples, zm, order


def sampler_finish(samples, z, pick, near, far, eik_idx):
    """-> z_vals [R, N+2+len(pick)] sorted, z_eik [R,1]."""
    lib = _lib.lib()
    samples, z = _f32c(samples), _f32c(z)
    R, N = samples.shape
    n_extra = 0 if pick is None else pick.numel()
    out = torch.empty(R, N + 2 + n_extra, device=z.device)
    zeik = torch.empty(R, 1, device=z.device)
    _lib.check(lib.neat_sampler_finish(_p(samples), N, _p(z), z.shape[1], _p(pick), n_extra, float(near), float(far), R,
                                       _p(eik_idx), _p(out), _p(zeik), _stream()), "neat_sampler_finish")
    return out, zeik


def sample_pdf(bins, weights, u, z_merge=None):
    """sample_pdf on the device in one launch: bins [R,nb], weights [R,nb-1], u [N] or [R,N] -> samples [R,N]
    (and, with z_merge [R,nz], the sorted union [R,nz+N] of get_z_vals_fine)."""
    bins, weights, u = _f32c(bins.detach()), _f32c(weights.detach()), _f32c(u)
    R, nb = bins.shape
    N = u.shape[-1]
    if weights.shape != (R, nb - 1):
        raise RuntimeError("sample_pdf: weights must be [R, nb - 1]")
    samples = torch.empty(R, N, device=bins.device)
    zm = _f32c(z_merge.detach()) if z_merge is not None else None
    z_out = torch.empty(R, zm.shape[1] + N, device=bins.device) if zm is not None else None
    _lib.check(_lib.lib().neat_sample_pdf(_p(bins), _p(weights), nb, R, _p(u), N if u.dim() == 2 else 0, N, _p(samples), _p(zm),
                                          zm.shape[1] if zm is not None else 0, _p(z_out), _stream()), "neat_sample_pdf")
    return samples, z_out


_LINSPACE = {}


def uniform_depths(R, N, near, far, rnd, device):
    """UniformSampler.get_z_vals in one launch: near / far floats or [R] / [R,1] tensors, rnd [R,N] (training jitter) or None."""
    key = (N, str(device))
    t = _LINSPACE.get(key)
    if t is None:
        t = _LINSPACE[key] = torch.linspace(0.0, 1.0, N, device=device)
    z = torch.empty(R, N, device=device)
    nt = _f32c(near.reshape(-1)) if isinstance(near, torch.Tensor) else None
    ft = _f32c(far.reshape(-1)) if isinstance(far, torch.Tensor) else None
    _lib.check(_lib.lib().neat_uniform_depths(_p(nt), 0.0 if nt is not None else float(near), _p(ft), 0.0 if ft is not None else float(far),
                                              _p(t), _p(_f32c(rnd)) if rnd is not None else None, R, N, _p(z), _stream()),
               "neat_uniform_depths")
    return z


# ---- the same rounds with the control flow on the device (no host sync; see include/neat_hip.h) ------------------------------------
def _ip(t, idx):
    return ctypes.c_void_p(t.data_ptr() + 4 * idx)


def sampler_round_dev(z, sdf_old, sdf_new, order, beta_in, beta0, eps, iters, add_tiny, u_refine, u_final, samples_final, z_final, ctl,
                      rnd, max_rounds):
    """Round `rnd` of Algorithm 1 with the refine/finish decision taken on the device.
    ctl int32 [2*max_rounds + 1] = open[max_rounds] | cont[max_rounds] | n_final (zero-initialised by the caller).
    -> (merged sdf [R,n], beta [R], refine samples [R,Ne], merged grid [R,n+Ne], order) -- garbage once the sampler has finished."""
    lib = _lib.lib()
    z, sdf_new = _f32c(z), _f32c(sdf_new)
    R, n = z.shape
    n_old = 0
    if order is not None:
        sdf_old, n_old = _f32c(sdf_old), sdf_old.shape[1]
    dev = z.device
    sdf_out, beta_out = torch.empty(R, n, device=dev), torch.empty(R, device=dev)
    gate = _ip(ctl, max_rounds + rnd - 1) if rnd > 0 else None
    _lib.check(lib.neat_sampler_bound_dev(_p(z), n, R, _p(sdf_old) if order is not None else None, _p(sdf_new), _p(order), n_old,
                                          _p(_f32c(beta_in)), _p(_f32c(beta0.reshape(1))), float(eps), int(iters), _p(sdf_out),
                                          _p(beta_out), _ip(ctl, rnd), gate, 1, _stream()), "neat_sampler_bound_dev")
    Ne, N = u_refine.shape[-1], u_final.shape[-1]
    fresh = torch.empty(R, Ne, device=dev)
    zm = torch.empty(R, n + Ne, device=dev)
    order_out = torch.empty(R, n + Ne, device=dev, dtype=torch.int32)
    _lib.check(lib.neat_sampler_resample_dev(_p(z), _p(sdf_out), n, R, _p(beta_out), float(add_tiny), _p(u_refine), Ne, _p(fresh), _p(zm),
                                             _p(order_out), _p(u_final), N if u_final.dim() == 2 else 0, N, _p(samples_final),
                                             _p(z_final), z_final.shape[1], _ip(ctl, 2 * max_rounds), _ip(ctl, rnd),
                                             _ip(ctl, max_rounds), rnd, max_rounds, _stream()), "neat_sampler_resample_dev")
    return sdf_out, beta_out, fresh, zm, order_out


def sdf_query_workspace(handle, P, device, fast=False):
    """Workspace of the values-mode SDF kernels for P points + the point stride of its x rows (the first 3 rows, feature-major): the
    sampler's launches write a round's query points there themselves (neat_sampler_init_rays / neat_sampler_round)."""
    lib = _lib.lib()
    prec = 5 if (fast and handle.precision == 4) else handle.precision
    ws = torch.empty(lib.neat_sdf_ws_floats(P, 0, prec), device=device, dtype=torch.float32)
    return ws, int(lib.neat_sdf_ldp(P, prec))


def sdf_values_laid_out(handle, ws, P, radius, scale, gate=None, fast=False):
    """sdf_values on the P points already laid out in `ws` (sdf_query_workspace) -> [P, 1]."""
    lib = _lib.lib()
    sdf = torch.empty(P, 1, device=ws.device)
    packed, netp = handle.packed()
    prec = 5 if (fast and handle.precision == 4) else handle.precision
    gptr, gval = (None, 0) if gate is None else (ctypes.c_void_p(gate[0].data_ptr() + 4 * gate[1]), int(gate[2]))
    _lib.check(lib.neat_sdf_values_laid_out(_p(packed), ctypes.byref(netp), P, prec, float(radius), float(scale), _p(ws), _p(sdf), gptr, gval,
                                            _stream()), "neat_sdf_values_laid_out")
    return sdf


def sampler_init_rays(z, beta_param, beta_min, beta_c, nctl, origins, dirs, x_fm, ldp, keys, n_step, n_cand, n_extra):
    """sampler_init + the first round's query points into x_fm [3, ldp] + (keys given) the training-mode picks of every possible final
    grid size -> (beta0, beta, ctl, pick_all [n_cand, n_extra] or None)."""
    z = _f32c(z.detach())
    R, n = z.shape
    dev = z.device
    beta0, beta, ctl = torch.empty(1, device=dev), torch.empty(R, device=dev), torch.empty(nctl, dtype=torch.int32, device=dev)
    pick_all = torch.empty(n_cand, n_extra, dtype=torch.int32, device=dev) if keys is not None else None
    _lib.check(_lib.lib().neat_sampler_init_rays(_p(z), R, n, _p(_f32c(beta_param.detach().reshape(1))), float(beta_min), float(beta_c),
                                                 _p(beta0), _p(beta), _p(ctl), nctl, _p(_f32c(origins.detach())), _p(_f32c(dirs.detach())),
                                                 _p(x_fm), int(ldp), _p(_f32c(keys)) if keys is not None else None, int(n_step), int(n_cand),
                                                 int(n_extra), _p(pick_all), _stream()), "neat_sampler_init_rays")
    return beta0, beta, ctl, pick_all


def sampler_round(z, sdf_old, sdf_new, order, beta_in, beta0, eps, iters, add_tiny, u_refine, u_final, samples_final, z_final, ctl, rnd,
                  max_rounds, origins, dirs, x_fm, ldp):
    """Round `rnd` of Algorithm 1 as ONE launch (neat_sampler_round): bound + refine resampling + the next round's query points into
    x_fm + (converged rays / last round) the final samples.  -> (merged sdf [R,n], beta [R], refine samples [R,Ne], merged grid
    [R,n+Ne], order) -- the last three None in the last round."""
    lib = _lib.lib()
    z, sdf_new = _f32c(z), _f32c(sdf_new)
    R, n = z.shape
    n_old = 0
    if order is not None:
        sdf_old, n_old = _f32c(sdf_old), sdf_old.shape[1]
    dev = z.device
    sdf_out, beta_out = torch.empty(R, n, device=dev), torch.empty(R, device=dev)
    Ne, N = u_refine.shape[-1], u_final.shape[-1]
    last = rnd + 1 >= max_rounds
    fresh = zm = order_out = None
    if not last:
        fresh, zm = torch.empty(R, Ne, device=dev), torch.empty(R, n + Ne, device=dev)
        order_out = torch.empty(R, n + Ne, device=dev, dtype=torch.int32)
    _lib.check(lib.neat_sampler_round(_p(z), n, R, _p(sdf_old) if order is not None else None, _p(sdf_new), _p(order), n_old,
                                      _p(_f32c(beta_in)), _p(_f32c(beta0.reshape(1))), float(eps), int(iters), _p(sdf_out), _p(beta_out),
                                      _p(ctl), int(rnd), int(max_rounds), float(add_tiny), _p(u_refine), Ne, _p(fresh), _p(zm), _p(order_out),
                                      _p(origins), _p(dirs), _p(x_fm) if not last else None, int(ldp), _p(u_final),
                                      N if u_final.dim() == 2 else 0, N, _p(samples_final), _p(z_final), z_final.shape[1], _stream()),
               "neat_sampler_round")
    return sdf_out, beta_out, fresh, zm, order_out


def sampler_finish_picked(samples_final, z_final, ctl, max_rounds, pick_all, n_step, n_extra, near, far, eik_idx):
    """-> z_vals [R, N+2+n_extra] sorted, z_eik [R,1]; the grid size comes from the control words, the picks from pick_all (training) or
    from the reference's linspace formula (pick_all None, eval)."""
    lib = _lib.lib()
    R, N = samples_final.shape
    dev = samples_final.device
    out, zeik = torch.empty(R, N + 2 + n_extra, device=dev), torch.empty(R, 1, device=dev)
    _lib.check(lib.neat_sampler_finish_picked(_p(samples_final), N, _p(z_final), z_final.shape[1], _ip(ctl, 2 * max_rounds), _p(pick_all),
                                              int(n_step), int(n_extra), float(near), float(far), R, _p(eik_idx), _p(out), _p(zeik), _stream()),
               "neat_sampler_finish_picked")
    return out, zeik


def sampler_finish_dev(samples_final, z_final, ctl, max_rounds, keys, n_extra, near, far, eik_idx):
    """-> z_vals [R, N+2+n_extra] sorted, z_eik [R,1], pick [n_extra] (device-chosen grid indices)."""
    lib = _lib.lib()
    R, N = samples_final.shape
    dev = samples_final.device
    out, zeik = torch.empty(R, N + 2 + n_extra, device=dev), torch.empty(R, 1, device=dev)
    pick = torch.empty(max(n_extra, 1), device=dev, dtype=torch.int32)
    _lib.check(lib.neat_sampler_finish_dev(_p(samples_final), N, _p(z_final), z_final.shape[1], _ip(ctl, 2 * max_rounds),
                                           _p(_f32c(keys)) if keys is not None else None, n_extra, _p(pick), float(near), float(far), R,
                                           _p(eik_idx), _p(out), _p(zeik), _stream()), "neat_sampler_finish_dev")
    return out, zeik, pick[:n_extra]


def linear_sum_assignment(cost, row_mask=None, col_mask=None):
    """scipy.optimize.linear_sum_assignment on the device, no host round trip (neat_lsap).
    cost [nr,nc] float32; row_mask [nr] / col_mask [nc] bool: rows / columns that take part (None = all).
    -> row_ind, col_ind int64 [min(nr,nc)] sorted by row and padded with -1, n_match int32 [1] (on the device)."""
    lib = _lib.lib()
    cost = _f32c(cost.detach())
    nr, nc = cost.shape
    k = min(nr, nc)
    rows = torch.empty(k, device=cost.device, dtype=torch.int64)
    cols = torch.empty(k, device=cost.device, dtype=torch.int64)
    n_match = torch.empty(1, device=cost.device, dtype=torch.int32)
    mask, cmask = _as_u8(row_mask), _as_u8(col_mask)
    ws = torch.empty(max(int(lib.neat_lsap_ws_bytes(nr, nc)), 8), device=cost.device, dtype=torch.uint8)
    _lib.check(lib.neat_lsap(_p(cost), nr, nc, _p(mask), _p(cmask), _p(rows), _p(cols), _p(n_match), _p(ws), _stream()), "neat_lsap")
    return rows, cols, n_match


class Project2DFn(torch.autograd.Function):
    """VolSDFNetwork.project2D as one launch forward, one backward (gradient w.r.t. the points only)."""

    @staticmethod
    def forward(ctx, K3, w2c, X):
        lib = _lib.lib()
        K3, w2c = _f32c(K3.detach()), _f32c(w2c.detach())
        Xc = _f32c(X.detach().reshape(-1, 3))
        uv = torch.empty(Xc.shape[0], 2, device=Xc.device)
        _lib.check(lib.neat_project2d(_p(K3), _p(w2c), _p(Xc), Xc.shape[0], _p(uv), _stream()), "neat_project2d")
        ctx.save_for_backward(K3, w2c, Xc)
        ctx.shape = X.shape
        ctx.set_materialize_grads(False)      # a projection nobody differentiates through costs no backward launch
        return uv.reshape(*X.shape[:-1], 2)

    @staticmethod
    def backward(ctx, d_uv):
        if d_uv is None:
            return None, None, None
        K3, w2c, Xc = ctx.saved_tensors
        d = _f32c(d_uv.reshape(-1, 2))
        dX = torch.empty_like(Xc)
        _lib.check(_lib.lib().neat_project2d_backward(_p(K3), _p(w2c), _p(Xc), Xc.shape[0], _p(d), _p(dX), _stream()),
                   "neat_project2d_backward")
        return None, None, dX.reshape(ctx.shape)


def project2d(K3, w2c, X):
    """K3 [3,3], w2c [3,4] = [R|T] (contiguous), X [...,3] -> [...,2]."""
    return Project2DFn.apply(K3, w2c, X)


class LineLossFn(torch.autograd.Function):
    """VolSDFLoss.get_line_loss in one launch: (loss, per_line, count); d loss / d pred is produced by the same launch."""

    @staticmethod
    def forward(ctx, pred, gt, weight, threshold):
        lib = _lib.lib()
        pred_c, gt_c, w_c = _f32c(pred.detach()), _f32c(gt.detach()), _f32c(weight.detach().reshape(-1))
        R = pred_c.shape[0]
        out2 = torch.empty(2, device=pred_c.device)
        per_line = torch.empty(R, device=pred_c.device)
        d_pred = torch.empty(R, 4, device=pred_c.device)
        _lib.check(lib.neat_line_loss(_p(pred_c), _p(gt_c), _p(w_c), R, float(threshold), _p(out2), _p(per_line), _p(d_pred),
                                      _stream()), "neat_line_loss")
        ctx.save_for_backward(d_pred)
        ctx.set_materialize_grads(False)
        count = out2[1]
        ctx.mark_non_differentiable(per_line, count)
        return out2[0], per_line, count

    @staticmethod
    def backward(ctx, g_loss, g_per_line, g_count):
        (d_pred,) = ctx.saved_tensors
        return (None if g_loss is None else g_loss * d_pred), None, None, None


def line_loss(pred, gt, weight, threshold=100.0):
    return LineLossFn.apply(pred, gt, weight, threshold)


class LineLossesFn(torch.autograd.Function):
    """The two line terms of VolSDFLoss.forward in one launch (neat_line_losses): -> (l2d pixel term, calibrated line loss, count)."""

    @staticmethod
    def forward(ctx, pred_px, pred_calib, gt5, K, threshold):
        pu, pc, g, Kc = _f32c(pred_px.detach()), _f32c(pred_calib.detach()), _f32c(gt5.detach()), _f32c(K.detach())
        R = pc.shape[0]
        if pu.shape != (R, 4) or pc.shape != (R, 4) or g.shape != (R, 5) or Kc.numel() != 9:
            raise RuntimeError("line_losses: pred [R,4] x2, gt [R,5], K [3,3]")
        out3 = torch.empty(3, device=pc.device)
        d_pred = torch.empty(R, 4, device=pc.device)
        _lib.check(_lib.lib().neat_line_losses(_p(pu), _p(pc), _p(g), _p(Kc), R, float(threshold), _p(out3), _p(d_pred), 1.0, _stream()),
                   "neat_line_losses")
        ctx.save_for_backward(d_pred)
        ctx.set_materialize_grads(False)
        l2d, count = out3[0], out3[2]
        ctx.mark_non_differentiable(l2d, count)
        return l2d, out3[1], count

    @staticmethod
    def backward(ctx, g_l2d, g_loss, g_count):
        (d_pred,) = ctx.saved_tensors
        return None, (None if g_loss is None else g_loss * d_pred), None, None, None


def line_losses(pred_px, pred_calib, gt5, K, threshold=100.0):
    return LineLossesFn.apply(pred_px, pred_calib, gt5, K, threshold)


def inv_small(A):
    """Inverse of one [n,n] matrix, n <= 4, in one launch (no host-side singularity check, no sync); no gradient."""
    A = A.detach()
    if not A.is_cuda:
        return torch.linalg.inv_ex(A).inverse
    A = _f32c(A)
    n = A.shape[-1]
    out = torch.empty(n, n, device=A.device)
    _lib.check(_lib.lib().neat_inv_small(_p(A), n, n, _p(out), _stream()), "neat_inv_small")
    return out


class FfnFn(torch.autograd.Function):
    """ffn(latents) = Linear-ReLU-Linear-ReLU-Linear (256 -> 256 -> 256 -> 3) in three launches (forward, data backward,
    weight backward) instead of ~25 latency-bound library GEMM / bias / relu kernels on a few dozen rows."""

    @staticmethod
    def forward(ctx, x, W0, b0, W1, b1, W2, b2):
        lib = _lib.lib()
        args = [_f32c(t.detach()) for t in (x, W0, b0, W1, b1, W2, b2)]
        J = args[0].shape[0]
        h1, h2 = torch.empty(J, 256, device=x.device), torch.empty(J, 256, device=x.device)
        y = torch.empty(J, 3, device=x.device)
        _lib.check(lib.neat_ffn_forward(_p(args[0]), J, *[_p(t) for t in args[1:]], _p(h1), _p(h2), _p(y), _stream()), "neat_ffn_forward")
        ctx.save_for_backward(args[0], args[1], args[3], args[5], h1, h2)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, W0, W1, W2, h1, h2 = ctx.saved_tensors
        J = x.shape[0]
        dy = _f32c(dy)
        ws2 = torch.empty(2 * J * 256, device=x.device)
        dx = torch.empty_like(x)
        flat = torch.empty(2 * (256 * 256 + 256) + 3 * 256 + 3, device=x.device)
        dW0, db0, dW1, db1, dW2, db2 = flat.split([65536, 256, 65536, 256, 768, 3])
        _lib.check(_lib.lib().neat_ffn_backward(_p(x), J, _p(W0), _p(W1), _p(W2), _p(h1), _p(h2), _p(dy), _p(ws2), _p(dx), _p(dW0), _p(db0),
                                               _p(dW1), _p(db1), _p(dW2), _p(db2), _stream()), "neat_ffn_backward")
        return dx, dW0.view(256, 256), db0, dW1.view(256, 256), db1, dW2.view(3, 256), db2


def ffn_junctions(x, linears):
    """linears: the three nn.Linear modules of VolSDFNetwork.ffn."""
    l0, l1, l2 = linears
    return FfnFn.apply(x, l0.weight, l0.bias, l1.weight, l1.bias, l2.weight, l2.bias)


DBSCAN_MAX_POINTS = 8192


def dbscan_means(points, eps):
    """DBSCAN(eps, min_samples=2) + cluster means on the device (neat_dbscan_means).  points [n,3] CUDA float32, n <= 8192.
    -> centres [n//2, 3] (sklearn's cluster order, zero padded), valid [n//2] bool, count int32 [1] -- no host sync."""
    lib = _lib.lib()
    pts = _f32c(points.detach().reshape(-1, 3))
    n = pts.shape[0]
    if n > DBSCAN_MAX_POINTS or n < 2:
        raise RuntimeError(f"dbscan_means handles 2..{DBSCAN_MAX_POINTS} points, got {n}")
    buf = torch.empty(n // 2, 4, device=pts.device)                      # centres followed by n/2 ints of scratch
    valid = torch.empty(n // 2, device=pts.device, dtype=torch.uint8)
    count = torch.empty(1, device=pts.device, dtype=torch.int32)
    ws = torch.empty(int(lib.neat_dbscan_ws_bytes(n)), device=pts.device, dtype=torch.uint8)
    flat = buf.view(-1)
    _lib.check(lib.neat_dbscan_means(_p(pts), n, float(eps), _p(flat), _p(valid), _p(count), _p(ws), _stream()), "neat_dbscan_means")
    return flat[:3 * (n // 2)].view(n // 2, 3), valid.view(torch.bool), count      # the kernel writes 0 / 1: reinterpret, no copy


_GRAD_ONE = {}
_UNIT_SEED = [False]      # True only while neat_amd.train._backward runs `loss.backward(gradient=grad_one(...))`


def grad_one(device):
    """The constant 1.0 to seed `loss.backward(gradient=...)` with (neat_amd.train._backward): no `ones_like` launch per step.  It is
    private to that call: created OUTSIDE any stream capture (a tensor first made inside an aborted capture would stay cached with its
    fill never executed), never handed to user code, value fixed."""
    key = str(device)
    t = _GRAD_ONE.get(key)
    if t is None:
        if torch.device(device).type == "cuda" and torch.cuda.is_current_stream_capturing():
            return torch.ones((), device=device)      # not cached: the capture's private pool owns it (Trainer.capture() makes the real one first)
        t = _GRAD_ONE[key] = torch.ones((), device=device)
    return t


class unit_seed:
    """`with unit_seed(device) as one: loss.backward(gradient=one)`: tells LossTailFn that the upstream gradient of this backward IS the
    constant 1 -- its flat buffer already holds the gradients of the total loss, so nothing is multiplied.  The fast path is taken on
    this flag (and the seed's identity), never on a pointer comparison alone: any other caller's gradient, or a mutated tensor, goes
    through the multiply."""

    def __init__(self, device):
        self.one = grad_one(device)

    def __enter__(self):
        self.prev, _UNIT_SEED[0] = _UNIT_SEED[0], True
        return self.one

    def __exit__(self, *exc):
        _UNIT_SEED[0] = self.prev
        return False


class LossTailFn(torch.autograd.Function):
    """VolSDFLoss.forward after the projections (loss_wfr.py:52-137) in three launches around neat_lsap: both line terms
    (neat_line_losses), rgb L1 + eikonal + the junction pair cost (neat_loss_terms), the matched-pair terms and the weighted total
    (neat_loss_pairs).  Differentiable inputs: rgb, grad_theta, the global junctions (3-D and calibrated 2-D) and the calibrated 2-D
    lines.  Every gradient leaves its kernel as a gradient of the TOTAL loss (the loss weights are applied there) into one flat buffer:
    backward is one multiply by the upstream gradient, or nothing at all inside `unit_seed` (the trainer's backward, which never
    retains the graph: the returned gradients are then views of the saved buffer)."""

    @staticmethod
    def forward(ctx, rgb, gtheta, glo3, glo2c, pred_calib, pred_px, gt5, Kmat, rgb_gt, loc3, loc2c, loc2, glo2, good,
                w_eik, w_line, w_j3, w_j2, threshold, lines3d=None, w2c=None):
        # lines3d / w2c given (the model's own outputs, networks.JunctionOutputs.calib_proj): pred_calib = project2d(I, w2c, lines3d) and
        # glo2c = project2d(I, w2c, glo3); the kernels then carry their gradients through those projections themselves and the node's
        # differentiable inputs are lines3d and glo3 -- no projection backward launch, no accumulation of glo3's two gradients
        lib = _lib.lib()
        dev = rgb.device
        c = lambda t: None if t is None else _f32c(t.detach())
        rgb_c, gt_c, gth_c = c(rgb), c(rgb_gt.reshape(-1, 3)), c(gtheta)
        glo3_c, glo2c_c, glo2_c, loc3_c, loc2c_c, loc2_c = c(glo3), c(glo2c), c(glo2), c(loc3), c(loc2c), c(loc2)
        pc, pu, g5, Kc = c(pred_calib.reshape(-1, 4)), c(pred_px.reshape(-1, 4)), c(gt5), c(Kmat)
        R, E = rgb_c.shape[0], 0 if gth_c is None else gth_c.shape[0]
        L = pc.shape[0]
        if pu.shape != (L, 4) or g5.shape != (L, 5) or Kc.numel() != 9:
            raise RuntimeError("loss_tail: lines [L,4] x2, gt [L,5], K [3,3]")
        K = 0 if loc3_c is None else loc3_c.shape[0]
        J = 0 if glo3_c is None else glo3_c.shape[0]
        have_pairs = bool(K and J)
        # (with junction pairs the launches write all seven scalars the caller reads: no fill launch)
        scal = torch.empty(8, device=dev) if have_pairs else torch.zeros(8, device=dev)
        line3 = torch.empty(3, device=dev)
        fold = lines3d is not None and w2c is not None
        x3_c, w2c_c = (_f32c(lines3d.detach().reshape(-1, 6)), _f32c(w2c.detach())) if fold else (None, None)
        if fold and (x3_c.shape[0] != L or w2c_c.shape != (3, 4)):
            raise RuntimeError("loss_tail: lines3d [L,2,3] and w2c [3,4]")
        sizes = [R * 3, E * 3, J * 3 if have_pairs else 0, J * 2 if have_pairs else 0, L * 4, L * 6 if fold else 0]
        flat = torch.empty(sum(sizes), device=dev)
        d_rgb, d_gth, d_glo3, d_glo2c, d_pred, d_x3 = flat.split(sizes)
        d_gth = d_gth if E else None
        pair_cost = torch.empty(K, J, device=dev) if have_pairs else None
        # both line terms | rgb + eikonal + the junction pair cost: independent of each other, the two workgroups of ONE launch
        _lib.check(lib.neat_loss_lines_terms(_p(pu), _p(pc), _p(g5), _p(Kc), L, float(threshold), _p(line3), _p(d_pred), w_line,
                                             _p(rgb_c), _p(gt_c), R, _p(gth_c), E, _p(loc3_c), _p(loc2c_c), K, _p(glo3_c), _p(glo2c_c), J,
                                             _p(scal), _p(d_rgb), _p(d_gth), _p(pair_cost), w_eik, _p(w2c_c), _p(x3_c), _p(d_x3) if fold else None,
                                             _stream()), "neat_loss_lines_terms")
        if have_pairs:
            ri, ci, n_match = linear_sum_assignment(pair_cost, good)
            loss = torch.empty((), device=dev)
            _lib.check(lib.neat_loss_pairs(_p(ri), _p(ci), _p(n_match), ri.shape[0], _p(loc3_c), _p(loc2c_c), _p(loc2_c), _p(glo3_c),
                                           _p(glo2c_c), _p(glo2_c), J, _p(pair_cost), _p(scal), _p(d_glo3), _p(d_glo2c),
                                           _ip(line3, 1), w_eik, w_line, w_j3, w_j2, 1, _p(loss), _p(w2c_c), _stream()), "neat_loss_pairs")
        else:
            loss = scal[0] + w_eik * scal[1] + w_line * line3[1]
        ctx.save_for_backward(flat)
        ctx.sizes = sizes
        ctx.set_materialize_grads(False)
        ctx.shapes = (rgb.shape, None if gtheta is None else gtheta.shape, None if glo3 is None or not have_pairs else glo3.shape,
                      None if glo2c is None or not have_pairs or fold else glo2c.shape, None if fold else pred_calib.shape,
                      lines3d.shape if fold else None)
        ctx.mark_non_differentiable(scal, line3)
        return loss, scal, line3

    @staticmethod
    def backward(ctx, g_loss, g_scal, g_line3):
        if g_loss is None:
            return (None,) * 21
        (flat,) = ctx.saved_tensors
        one = _GRAD_ONE.get(str(flat.device))
        if not (_UNIT_SEED[0] and one is not None and g_loss.data_ptr() == one.data_ptr()):
            flat = flat * g_loss          # a fresh tensor: the views below never alias the saved buffer (retain_graph callers)
        g = flat.split(ctx.sizes)
        s_rgb, s_gth, s_glo3, s_glo2c, s_pred, s_x3 = ctx.shapes
        return (g[0].view(s_rgb),
                None if s_gth is None else g[1].view(s_gth),
                None if s_glo3 is None else g[2].view(s_glo3),
                None if s_glo2c is None else g[3].view(s_glo2c),
                None if s_pred is None else g[4].view(s_pred)) + (None,) * 14 + (None if s_x3 is None else g[5].view(s_x3), None)


def loss_tail(rgb, gtheta, glo3, glo2c, pred_calib, pred_px, gt5, Kmat, rgb_gt, loc3, loc2c, loc2, glo2, good, w_eik, w_line, w_j3, w_j2,
              threshold=100.0, lines3d=None, w2c=None):
    """-> (total loss, scal [8] = rgb, eikonal, j3d, j2d, j2d pixels, jcount, total, -, line3 [3] = l2d pixel term, line loss, count)."""
    return LossTailFn.apply(rgb, gtheta, glo3, glo2c, pred_calib, pred_px, gt5, Kmat, rgb_gt, loc3, loc2c, loc2, glo2, good,
                            float(w_eik), float(w_line), float(w_j3), float(w_j2), float(threshold), lines3d, w2c)


def camera_setup(uv, uv_proj, pose, intrinsics):
    """Everything the forward derives from the camera alone in ONE launch (neat_camera_setup): rays through uv [1,R,2] (dirs [R,3],
    origins [R,3]), rays through uv_proj (dirs [R,3]; None: not wanted), w2c [3,4] = [R | T] of pose^-1 and the contiguous K3 [3,3].
    Same values as camera_rays x 2 + camera_mats."""
    uv_c, pose_c, Kc = _f32c(uv.detach()), _f32c(pose.detach()), intrinsics.detach()
    if pose_c.shape != (1, 4, 4) or Kc.dtype != torch.float32 or not Kc.is_cuda or Kc.stride(-1) != 1 or Kc.shape[0] != 1:
        raise RuntimeError("camera_setup: one 4x4 pose, intrinsics with unit column stride, CUDA float32")
    R, dev = uv_c.shape[1], uv_c.device
    up = _f32c(uv_proj.detach()) if uv_proj is not None else None
    if up is not None and up.shape != uv_c.shape:
        raise RuntimeError("camera_setup: uv_proj must have uv's shape")
    dirs, origins = torch.empty(R, 3, device=dev), torch.empty(R, 3, device=dev)
    dirs2 = torch.empty(R, 3, device=dev) if up is not None else None
    w2c, K3 = torch.empty(3, 4, device=dev), torch.empty(3, 3, device=dev)
    _lib.check(_lib.lib().neat_camera_setup(_p(uv_c), _p(up), _p(pose_c), _p(Kc), int(Kc.stride(-2)), R, _p(dirs), _p(origins), _p(dirs2),
                                            _p(w2c), _p(K3), _stream()), "neat_camera_setup")
    return dirs, origins, dirs2, w2c, K3


class Project2DPairFn(torch.autograd.Function):
    """VolSDFNetwork.project2D of the same points with two intrinsics (pixels / calibrated) as one launch; backward: one launch per
    output that received a gradient (gradient w.r.t. the points only)."""

    @staticmethod
    def forward(ctx, K3, K3b, w2c, X):
        lib = _lib.lib()
        K3, K3b, w2c = _f32c(K3.detach()), _f32c(K3b.detach()), _f32c(w2c.detach())
        Xc = _f32c(X.detach().reshape(-1, 3))
        uv, uv2 = torch.empty(Xc.shape[0], 2, device=Xc.device), torch.empty(Xc.shape[0], 2, device=Xc.device)
        _lib.check(lib.neat_project2d_pair(_p(K3), _p(K3b), _p(w2c), _p(Xc), Xc.shape[0], _p(uv), _p(uv2), _stream()), "neat_project2d_pair")
        ctx.save_for_backward(K3, K3b, w2c, Xc)
        ctx.shape = X.shape
        ctx.set_materialize_grads(False)
        return uv.reshape(*X.shape[:-1], 2), uv2.reshape(*X.shape[:-1], 2)

    @staticmethod
    def backward(ctx, d_uv, d_uv2):
        if d_uv is None and d_uv2 is None:
            return None, None, None, None
        K3, K3b, w2c, Xc = ctx.saved_tensors
        total = None
        for Kx, d in ((K3, d_uv), (K3b, d_uv2)):
            if d is None:
                continue
            dX = torch.empty_like(Xc)
            _lib.check(_lib.lib().neat_project2d_backward(_p(Kx), _p(w2c), _p(Xc), Xc.shape[0], _p(_f32c(d.reshape(-1, 2))), _p(dX), _stream()),
                       "neat_project2d_backward")
            total = dX if total is None else total + dX
        return None, None, None, total.reshape(ctx.shape)


def project2d_pair(K3, K3b, w2c, X):
    """-> (project2d(K3, w2c, X), project2d(K3b, w2c, X)) from one launch."""
    return Project2DPairFn.apply(K3, K3b, w2c, X)


class SdfNormalsFn(torch.autograd.Function):
    """ImplicitNetwork.get_outputs without the feature rows: (clamped sdf [P,1], d sdf / dx [P,3]), differentiable wrt the 27 SDF parameters
    like SdfOutputsFn -- minus the two row-major copies of lin8's output (one launch) that the junction block never reads (rend_a :441-443)."""

    @staticmethod
    def forward(ctx, handle, x, radius, scale, *params):
        lib = _lib.lib()
        ctx.set_materialize_grads(False)
        x = _f32c(x.detach())
        P = x.shape[0]
        packed, netp = handle.packed()
        prec = handle.precision
        ws = torch.empty(lib.neat_sdf_ws_floats(P, 1, prec), device=x.device, dtype=torch.float32)
        sdf, grad = torch.empty(P, 1, device=x.device), torch.empty(P, 3, device=x.device)
        _lib.check(lib.neat_sdf_forward(_p(packed), ctypes.byref(netp), _p(x), P, 1, prec, float(radius), float(scale), _p(ws),
                                        None, _p(sdf), None, _p(grad), _stream()), "neat_sdf_forward(normals)")
        ctx.handle, ctx.P, ctx.ws, ctx.packed, ctx.netp, ctx.prec = handle, P, ws, packed, netp, prec
        ctx.keep = params
        return sdf, grad

    @staticmethod
    def backward(ctx, d_sdf, d_grad):
        lib = _lib.lib()
        gr, views, _ = _grad_buffers(ctx.handle, 0, N_SDF, ctx.ws.device)
        d_sdf, d_grad = _f32c(d_sdf), _f32c(d_grad)
        _lib.check(lib.neat_sdf_backward(_p(ctx.packed), ctypes.byref(ctx.netp), _p(ctx.ws), ctx.P, ctx.prec, None, _p(d_sdf), None, _p(d_grad),
                                         ctypes.byref(gr), _stream()), "neat_sdf_backward")
        ctx.ws = None
        return (None, None, None, None, *views)


def sdf_point_normals(handle, x, radius, scale):
    """-> (clamped sdf [P,1], d sdf / dx [P,3]): get_outputs as the junction block uses it."""
    if x.shape[0] == 0:
        return x.new_zeros(0, 1), x.new_zeros(0, 3)
    return SdfNormalsFn.apply(handle, x, radius, scale, *handle.tensors(0, N_SDF))


def camera_mats(pose, intrinsics):
    """pose [4,4] cam-to-world, intrinsics [>=3, >=3] -> (w2c [3,4] = the first three rows of pose^-1, K3 [3,3] contiguous): one launch
    (rend_a :424-431, :440; no gradient)."""
    pose, Kc = _f32c(pose.detach()), intrinsics.detach()
    if Kc.dtype != torch.float32 or not Kc.is_cuda or Kc.stride(-1) != 1 or pose.shape != (4, 4):
        raise RuntimeError("camera_mats: pose [4,4] and intrinsics with unit column stride, CUDA float32")
    w2c, K3 = torch.empty(3, 4, device=pose.device), torch.empty(3, 3, device=pose.device)
    _lib.check(_lib.lib().neat_camera_mats(_p(pose), _p(Kc), int(Kc.stride(-2)), _p(w2c), _p(K3), _stream()), "neat_camera_mats")
    return w2c, K3


def l3d_points(x, origins, dirs, normals):
    """l3d = o + t d with t = <x - o, n> / (<d, n> +- 1e-6) per ray (rend_a :441-447), one launch, no gradient."""
    x, o, d, n = (_f32c(t.detach().reshape(-1, 3)) for t in (x, origins, dirs, normals))
    out = torch.empty_like(x)
    _lib.check(_lib.lib().neat_l3d(_p(x), _p(o), _p(d), _p(n), x.shape[0], _p(out), _stream()), "neat_l3d")
    return out


def junction_cost(cand2d, gt2d):
    """cost [V, C] = |cand2d[c] - gt2d[v]|_2 (rend_a :472), one launch."""
    cand2d, gt2d = _f32c(cand2d.detach()), _f32c(gt2d.detach())
    V, C = gt2d.shape[0], cand2d.shape[0]
    cost = torch.empty(V, C, device=cand2d.device)
    _lib.check(_lib.lib().neat_junction_cost(_p(cand2d), _p(gt2d), V, C, _p(cost), _stream()), "neat_junction_cost")
    return cost


def junction_gate(rows, cols, cost, cand3d, cand2d, cand2d_calib, use_median):
    """Matched costs of the pairs, median / 10 px gate and the gathered matched candidates (rend_a :474-489), one launch.
    -> median [1] or None, good [K] bool, j3d [K,3], j2d [K,2], j2d_calib [K,2] (padded: pairs that do not exist are masked)."""
    K = rows.shape[0]
    dev = cost.device
    cand3d, cand2d, cand2d_calib = _f32c(cand3d.detach()), _f32c(cand2d.detach()), _f32c(cand2d_calib.detach())
    median = torch.empty(1, device=dev) if use_median else None
    good = torch.empty(K, device=dev, dtype=torch.uint8)
    j3d, j2d, j2dc = torch.empty(K, 3, device=dev), torch.empty(K, 2, device=dev), torch.empty(K, 2, device=dev)
    _lib.check(_lib.lib().neat_junction_gate(_p(rows), _p(cols), K, _p(cost), cost.shape[1], _p(cand3d), _p(cand2d), _p(cand2d_calib),
                                             1 if use_median else 0, _p(median), _p(good), _p(j3d), _p(j2d), _p(j2dc), _stream()),
               "neat_junction_gate")
    return (median.reshape(()) if use_median else None), good.view(torch.bool), j3d, j2d, j2dc   # 0 / 1 bytes: reinterpret, no copy
