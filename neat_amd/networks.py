"""Model classes with the reference's plug-in API (code/model/networks/neat_wfr_rend_a.py), backed by HIP.

Drop-in: set `train.model_class = neat_amd.networks.VolSDFNetwork` in a reference conf; constructor,
forward(input)->dict, sub-module names and state_dict keys are the reference's.
All P-sized work (PE, the three MLPs, normals, compositing, and their backward incl. the double backward)
runs in neat_amd/csrc; what stays in torch here is the R-sized glue of the attraction-field block.
"""
import math

import numpy as np
import os

import torch
from torch import nn

from . import ops, rend_util
from .conf import ConfTree, from_dict
from .density import LaplaceDensity
from .ray_sampler import ErrorBoundSampler, HierarchicalSampler


def _wn_linear(in_dim, out_dim, weight_norm=True):
    """Parameter container with the reference's names: bias, weight_g [out,1], weight_v [out,in].
    Its own forward is never used -- the HIP kernels read weight_v / weight_g / bias directly."""
    lin = nn.Linear(in_dim, out_dim)
    if not weight_norm:
        raise NotImplementedError("the HIP path implements the weight-normed layers of the shipped confs")
    return lin


DEFAULT_PRECISION = "fp16x3"      # conf key model.hip_precision


def _wrap_wn(lin):
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        return nn.utils.weight_norm(lin)


def _triples(module, n):
    out = []
    for l in range(n):
        lin = getattr(module, f"lin{l}")
        out.append((lin.weight_v, lin.weight_g, lin.bias))
    return out


class _HipModule(nn.Module):
    """Keeps a NetHandle that always points at the module's *current* parameter tensors."""

    def _handle(self):
        h = self.__dict__.get("_neat_handle")
        if h is None:
            from . import _lib
            h = ops.NetHandle()
            h.precision = _lib.PRECISIONS[DEFAULT_PRECISION]      # a sub-module used on its own gets the model's default build
            self.__dict__["_neat_handle"] = h
        return h

    def set_precision(self, precision):
        """One of _lib.PRECISIONS: 'fp16x3' (3-product f16 forward chains + the f16 backward pass: the reference's outputs to 1e-4 and
        gradients to 2e-3 at 8x the speed of 'fp32'; the drop-in default), 'fp32' (exact-f32 MFMA), 'bf16' / 'fp16' (16-bit MFMA with
        fp32 accumulate, fastest, 16-bit-grade parity), 'bf16x3' (split-bf16 products in the fp32 layouts)."""
        from . import _lib
        if precision not in _lib.PRECISIONS:
            raise ValueError(f"precision must be one of {sorted(_lib.PRECISIONS)}")
        self._handle().precision = _lib.PRECISIONS[precision]
        return self


class ImplicitNetwork(_HipModule):
    """SDF MLP: PE-6 -> 8 x 256 softplus(100) with a skip into layer 4 -> [sdf, 256 features]  (rend_a :14-137)."""

    def __init__(self, feature_vector_size, sdf_bounding_sphere, d_in, d_out, dims, geometric_init=True, bias=1.0,
                 skip_in=(), weight_norm=True, multires=0, sphere_scale=1.0, inside_out=False):
        super().__init__()
        self.sdf_bounding_sphere, self.sphere_scale = sdf_bounding_sphere, sphere_scale
        self.skip_in, self.inside_out = tuple(skip_in), inside_out
        pe_dim = d_in * (1 + 2 * multires) if multires > 0 else d_in
        widths = [pe_dim] + list(dims) + [d_out + feature_vector_size]
        if widths != [39] + [256] * 8 + [257] or self.skip_in != (4,) or d_in != 3:
            raise NotImplementedError(f"HIP path implements dims [39,256x8,257] skip (4,), got {widths} {self.skip_in}")
        self.num_layers = len(widths)
        self.multires = multires
        last = self.num_layers - 2
        for l in range(last + 1):
            fan_out = widths[l + 1] - widths[0] if (l + 1) in self.skip_in else widths[l + 1]
            lin = _wn_linear(widths[l], fan_out, weight_norm)
            if geometric_init:          # sphere-like start (rend_a :55-69); init precedes weight_norm, so g = |v|
                with torch.no_grad():
                    if l == last:
                        lin.weight.normal_(math.sqrt(math.pi) / math.sqrt(widths[l]), 0.0001)
                        lin.bias.fill_(-bias)
                    else:
                        lin.bias.zero_()
                        std = math.sqrt(2.0) / math.sqrt(fan_out)
                        if l == 0:
                            lin.weight[:, 3:].zero_()
                            lin.weight[:, :3].normal_(0.0, std)
                        else:
                            lin.weight.normal_(0.0, std)
                            if l in self.skip_in:
                                lin.weight[:, -(widths[0] - 3):].zero_()
            setattr(self, f"lin{l}", _wrap_wn(lin))
        self.softplus = nn.Softplus(beta=100)

    def triples(self):
        """(weight_v, weight_g, bias) of the nine layers as the kernels see them.  inside_out (rend_a :94-95: `x[:, :1] = -x[:, :1]` after
        the last layer) = the sdf row of lin8 with the opposite sign: gain and bias row 0 are negated on the way in (two small launches
        per forward; their gradients come back through the same products)."""
        tr = _triples(self, 9)
        if self.inside_out:
            v, g, b = tr[8]
            sign = self.__dict__.get("_sdf_sign")
            if sign is None or sign.device != g.device:
                sign = torch.ones(g.shape[0], device=g.device, dtype=g.dtype)
                sign[0] = -1.0
                self.__dict__["_sdf_sign"] = sign
            g2, b2 = g * sign.view_as(g), b * sign
            g2._neat_key = (g.data_ptr(), g._version, "inside_out")
            b2._neat_key = (b.data_ptr(), b._version, "inside_out")
            tr[8] = (v, g2, b2)
        return tr

    def handle(self):
        owner = self.__dict__.get("_neat_owner")
        if owner is not None and owner() is not None:
            # attached to a VolSDFNetwork: one handle for all three networks, refreshed as a whole (its heads may be represented by
            # tensors derived per forward, which must not go stale when only this sub-module is called)
            return owner().handle()
        h = self._handle()
        h.set_layers(0, self.triples())
        return h

    def _outputs(self, x, radius):
        return ops.sdf_outputs(self.handle(), x, radius, self.sphere_scale)

    def forward(self, input):
        return self._outputs(input, 0.0)[0]

    def gradient(self, x):
        """d raw-sdf / dx, differentiable wrt the parameters (eikonal term, rend_a :98-109)."""
        return self._outputs(x, 0.0)[3]

    def get_outputs(self, x):
        _, sdf, feat, grad = self._outputs(x, self.sdf_bounding_sphere)
        return sdf, feat, grad

    def get_sdf_vals(self, x, gate=None, fast=False):
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            return self._outputs(x, self.sdf_bounding_sphere)[1]
        return ops.sdf_values(self.handle(), x, self.sdf_bounding_sphere, self.sphere_scale, gate=gate, fast=fast)

    def get_sdf_vals_rays(self, cam_loc, ray_dirs, z, gate=None, fast=False):
        """get_sdf_vals(cam_loc + z * ray_dirs) for R rays x S depths without autograd (what a sampler round asks for,
        ray_sampler.py:146-151): the points never exist as a row-major tensor."""
        return ops.sdf_values_rays(self.handle(), cam_loc, ray_dirs, z, self.sdf_bounding_sphere, self.sphere_scale, gate=gate, fast=fast)


class _Head(_HipModule):
    first_layer = None

    def __init__(self, feature_vector_size, mode, d_in, d_out, dims, weight_norm=True, multires_view=0, expect_in=None):
        super().__init__()
        if mode not in ("idr", "nerf"):
            raise NotImplementedError(f"head mode {mode!r}: the reference knows 'idr' and 'nerf'")
        self.mode = mode
        in_dim = d_in + feature_vector_size + (6 * multires_view if multires_view > 0 else 0)
        widths = [in_dim] + list(dims) + [d_out]
        # mode = 'nerf' (rend_a :180-181,240-241): the input is [view, feature] -- no point, no normal; d_in = 3.  The kernels keep their
        # [p | view | normal | feature] input layout; the parameter's columns are spread into it with zero weights for p and the normal
        if mode == "nerf":
            expect_in -= 6
        if in_dim != expect_in or list(dims) != [256] * 4:
            raise NotImplementedError(f"HIP path implements {expect_in}->256x4->{d_out}, got {widths}")
        self.full_in = expect_in + (6 if mode == "nerf" else 0)
        self.num_layers = len(widths)
        self.multires_view = multires_view
        for l in range(self.num_layers - 1):
            setattr(self, f"lin{l}", _wrap_wn(_wn_linear(widths[l], widths[l + 1], weight_norm)))
        self.relu, self.sigmoid = nn.ReLU(), nn.Sigmoid()

    def triples(self):
        """(weight_v, weight_g, bias) of the five layers as the kernels see them: lin0's weight_v spread over the kernels' input columns
        when mode = 'nerf' (zeros + one index_copy per forward; weight norm divides by the row norm, which the zero columns leave alone)."""
        tr = _triples(self, 5)
        if self.mode == "nerf":
            v, g, b = tr[0]
            nv = v.shape[1] - 256                        # view columns: 27 (PE-4) or 3
            idx = self.__dict__.get("_nerf_cols")
            if idx is None or idx.device != v.device:
                idx = torch.cat([torch.arange(3, 3 + nv), torch.arange(6 + nv, 6 + nv + 256)]).to(v.device)
                self.__dict__["_nerf_cols"] = idx
            v2 = torch.zeros(v.shape[0], self.full_in, device=v.device, dtype=v.dtype).index_copy(1, idx, v)
            v2._neat_key = (v.data_ptr(), v._version, "nerf")
            tr[0] = (v2, g, b)
        return tr

    def _standalone(self, points, normals, view_dirs, feature_vectors):
        owner = self.__dict__.get("_neat_owner")
        if owner is None or owner() is None:
            raise RuntimeError("this head is not attached to a VolSDFNetwork (the HIP kernels pack all three networks)")
        if torch.is_grad_enabled() and (normals.requires_grad or feature_vectors.requires_grad
                                        or any(p.requires_grad for p in self.parameters())):
            raise NotImplementedError("stand-alone head calls are forward-only; training goes through "
                                      "VolSDFNetwork.forward (fused main pass with full backward). Wrap in torch.no_grad().")
        return ops.heads_forward(owner().handle(), points, normals, view_dirs, feature_vectors)


class RenderingNetwork(_Head):
    """[p, PE4(view), normal, feature] -> 4x256 ReLU -> sigmoid rgb   (rend_a :199-255)."""

    def __init__(self, feature_vector_size, mode, d_in, d_out, dims, weight_norm=True, multires_view=0):
        super().__init__(feature_vector_size, mode, d_in, d_out, dims, weight_norm, multires_view, expect_in=289)
        if d_out != 3 or multires_view != 4:
            raise NotImplementedError("rendering head: d_out=3, multires_view=4")

    def forward(self, points, normals, view_dirs, feature_vectors):
        return self._standalone(points, normals, view_dirs, feature_vectors)[0]


class AttractionFieldNetwork(_Head):
    """[p, view, normal, feature] -> 4x256 ReLU -> 6 ; y = p + offsets.reshape(2,3)   (rend_a :139-197)."""

    def __init__(self, feature_vector_size, mode, d_in, d_out, dims, weight_norm=True, multires_view=0):
        super().__init__(feature_vector_size, mode, d_in, d_out, dims, weight_norm, multires_view, expect_in=265)
        if d_out != 6 or multires_view != 0:
            raise NotImplementedError("attraction head: d_out=6, no view encoding")

    def forward(self, points, normals, view_dirs, feature_vectors):
        return self._standalone(points, normals, view_dirs, feature_vectors)[1]


_EYE3 = {}


def _eye3(device):
    key = str(device)
    if key not in _EYE3:
        _EYE3[key] = torch.eye(3, device=device)
    return _EYE3[key]


_STAGING = {}


def _to_device_async(t, device, site=None):
    """CPU-drawn randoms (the reference's RNG stream) -> device without stalling the host behind queued GPU work.
    Staged through a ring of 4 pinned buffers per draw site: `Tensor.pin_memory()` per call costs a hipHostMalloc (measured: 4 ms on
    average for the sampler's 512 KB draws, with 80 ms stalls every other step when the host allocator trims its cache)."""
    if device.type != "cuda":
        return t.to(device)
    key = (site, tuple(t.shape), t.dtype, str(device))
    ring = _STAGING.get(key)
    if ring is None:
        ring = _STAGING[key] = {"bufs": [], "pos": 0}
    if len(ring["bufs"]) < 4:
        ring["bufs"].append((torch.empty_like(t).pin_memory(), torch.cuda.Event()))
        buf, ev = ring["bufs"][-1]
    else:
        buf, ev = ring["bufs"][ring["pos"] % 4]
        ring["pos"] += 1
        ev.synchronize()                # the copy that used this buffer four draws ago is long done
    buf.copy_(t)
    out = buf.to(device, non_blocking=True)
    ev.record()
    return out


def _device_copy(obj, attr, device):
    """Device-resident copy of a per-view CPU tensor attribute, made once (a pageable H2D copy stalls the host)."""
    cache = obj.__dict__.setdefault("_device_cache", {})
    src = getattr(obj, attr)
    hit = cache.get((attr, str(device)))
    if hit is None or hit[0] is not src:
        hit = (src, src.to(device).contiguous())
        cache[(attr, str(device))] = hit
    return hit[1]


class JunctionOutputs(dict):
    """Model output dict whose `j2d_local`, `j3d_local`, `j2d_local_calib` entries (the matched local junctions that
    pass the gate, rend_a :478-489) are compacted lazily: `padded[k]` [K,.] + `good` [K] live on the device, and
    `out[k]` = `padded[k][good]` is only materialised (one host sync for the data-dependent shape) when read."""

    def __init__(self, base, good, padded):
        super().__init__(base)
        self.good, self.padded = good, padded
        # (set by VolSDFNetwork.forward on CUDA) how the calibrated projections were made: {"w2c", "lines3d", "lines2d_calib",
        # "j3d_global", "j2d_global_calib"} with lines2d_calib = project2D(I, w2c, lines3d) etc.; neat_amd.loss folds their backward
        # passes into its own kernels when it is handed exactly these tensors
        self.calib_proj = None

    def __missing__(self, key):
        if key in self.padded:
            self[key] = self.padded[key][self.good]
            return dict.__getitem__(self, key)
        raise KeyError(key)

    def __contains__(self, key):
        return dict.__contains__(self, key) or key in self.padded

    def get(self, key, default=None):
        return self[key] if key in self else default

    def keys(self):
        return list(dict.keys(self)) + [k for k in self.padded if not dict.__contains__(self, k)]

    def items(self):
        return [(k, self[k]) for k in self.keys()]

    def detached(self):
        """The same outputs without their autograd history (what a trainer hands back after it has run backward itself)."""
        det = lambda v: v.detach() if isinstance(v, torch.Tensor) else v
        return JunctionOutputs({k: det(v) for k, v in dict.items(self)}, det(self.good), {k: det(v) for k, v in self.padded.items()})


class VolSDFNetwork(_HipModule):
    def __init__(self, conf):
        super().__init__()
        if isinstance(conf, dict) and not hasattr(conf, "get_config"):
            conf = from_dict(conf)
        self.feature_vector_size = conf.get_int("feature_vector_size")
        self.scene_bounding_sphere = conf.get_float("scene_bounding_sphere", default=1.0)
        self.white_bkgd = conf.get_bool("white_bkgd", default=False)
        self.register_buffer("bg_color", torch.tensor(conf.get_list("bg_color", default=[1.0, 1.0, 1.0])).float(),
                             persistent=False)
        self.implicit_network = ImplicitNetwork(self.feature_vector_size,
                                                0.0 if self.white_bkgd else self.scene_bounding_sphere,
                                                **conf.get_config("implicit_network"))
        self.rendering_network = RenderingNetwork(self.feature_vector_size, **conf.get_config("rendering_network"))
        self.attraction_network = AttractionFieldNetwork(self.feature_vector_size, **conf.get_config("attraction_network"))
        self.density = LaplaceDensity(**conf.get_config("density"))
        self.ray_sampler = ErrorBoundSampler(self.scene_bounding_sphere, **conf.get_config("ray_sampler"))
        self.ray_sampler.sync_free = bool(conf.get_bool("hip_sampler_sync_free", default=False))      # new optional key
        if conf.get_string("hip_sampler", default="error_bound") == "hierarchical":      # new optional key: BASELINE config C5
            rs = conf.get_config("ray_sampler")
            self.ray_sampler = HierarchicalSampler(self.scene_bounding_sphere, rs.get_float("near", default=0.0),
                                                   conf.get_int("hip_sampler_coarse", default=rs.get_int("N_samples")),
                                                   conf.get_int("hip_sampler_fine", default=rs.get_int("N_samples")))
        # global junction MLP on learnable latents (rend_a :272-303)
        cj = conf.get_config("global_junctions", default=ConfTree())
        hidden, depth = cj.get_int("dim_hidden", default=256), cj.get_int("num_layers", default=2)
        self.latents = nn.Parameter(torch.empty(cj.get_int("num_junctions", default=1024), hidden))
        nn.init.normal_(self.latents, mean=0.0, std=1)
        stack = []
        for i in range(depth + 1):
            stack.append(nn.Linear(hidden, hidden if i != depth else 3))
            if i != depth:
                stack.append(nn.ReLU())
        self.ffn = nn.Sequential(*stack)
        self.dbscan_enabled = conf.get_bool("dbscan_enabled", default=True)
        self.use_median = conf.get_bool("use_median", default=False)
        self.junction_eikonal = conf.get_bool("junction_eikonal", default=False)
        self.use_l3d = conf.get_bool("use_l3d", default=False)
        import weakref
        for head in (self.rendering_network, self.attraction_network, self.implicit_network):
            head.__dict__["_neat_owner"] = weakref.ref(self)
        self.static_randoms = None        # see _cpu_random
        self.use_side_stream = int(os.environ.get("NEAT_SIDE_STREAMS", "0"))   # forward(): bit 0: ffn(latents), bit 1: get_outputs(points3d) on a second stream
        self._side = {}
        self.z_vals_override = None       # bench/tests: given depth samples [R,S] bypass the sampler
        # new optional key.  Default = the parity-grade build a maintainer should get from changing `train.model_class` alone: fp16x3
        # passes every reference golden at the fp32 bars (outputs 1e-4, gradients 2e-3) at 8x the speed of the exact-f32 build
        self.set_precision(conf.get_string("hip_precision", default=DEFAULT_PRECISION))
        # new optional key (fp16x3 only): the depth sampler's SDF queries through the one-product f16 chain -- 3x faster queries, the
        # sampled distribution stays the reference's to ~1e-4 of the depth range, the individual depths do not
        self.sampler_fast_values = conf.get_bool("hip_sampler_fast_values", default=False)

    # ---- HIP plumbing ----------------------------------------------------------------------------
    def set_precision(self, precision):
        super().set_precision(precision)
        self.handle()                  # the SDF sub-module shares this handle (and therefore the precision)
        return self

    def handle(self):
        h = self._handle()
        h.set_layers(0, self.implicit_network.triples())
        h.set_layers(9, self.rendering_network.triples())
        h.set_layers(14, self.attraction_network.triples())
        self.implicit_network.__dict__["_neat_handle"] = h      # share the pack cache with the sub-module API
        return h

    def _sphere(self):
        return 0.0 if self.white_bkgd else self.scene_bounding_sphere

    def _render(self, cam_loc, ray_dirs, z_vals, want_normal_map, eik_points=None, with_eik=False):
        if z_vals.is_cuda and type(self.density) is LaplaceDensity:
            # the raw parameter and beta_min: |beta| + beta_min (density.py:29-30) is formed inside the compositing kernels and its
            # backward (sum over the rays, sign) inside neat_render_backward -- five elementwise launches less per step
            beta, beta_min = self.density.beta, self.density.beta_min
        else:
            beta, beta_min = self.density.get_beta(), 0.0
        rgb, lines3d, depth, xyz, eik_grad, weights, sdf, points, nmap = ops.render_rays(
            self.handle(), cam_loc, ray_dirs, z_vals, beta, self._sphere(),
            self.implicit_network.sphere_scale, want_normal_map, eik_points, self.bg_color if self.white_bkgd else None, beta_min)
        if with_eik:
            return rgb, lines3d, depth, xyz, weights, sdf, points, nmap, eik_grad
        return rgb, lines3d, depth, xyz, weights, sdf, points, nmap

    # ---- reference helpers -------------------------------------------------------------------------
    def project2D(self, K, R, T, points3d):
        shape = points3d.shape
        assert shape[-1] == 3
        cam = (K @ (R @ points3d.reshape(-1, 3).t() + T)).t()
        w = cam[:, -1:]
        w = w + torch.where(w.abs() < 1e-8, torch.full_like(w, 1e-8), torch.zeros_like(w)) * torch.where(
            w >= 0, torch.ones_like(w), -torch.ones_like(w))
        return (cam / w).reshape(*shape)[..., :2]

    def cluster_dbscan(self, points, eps=0.01, min_samples=2):
        from sklearn.cluster import DBSCAN
        labels = DBSCAN(eps=eps, min_samples=min_samples).fit(points).labels_
        centres = [points[labels == i].mean(axis=0) for i in range(labels.max() + 1)]
        return torch.tensor(np.array(centres).reshape(-1, 3)).float().to(self.latents.device)

    def volume_rendering(self, z_vals, sdf):
        return ops.volume_weights(z_vals, sdf, self.density.get_beta())

    def _rays(self, input, key="uv"):
        if input[key].is_cuda and input["pose"].shape[1:] == (4, 4):      # directions and the per-ray camera centre in one launch
            dirs, _, origins = ops.camera_rays(input[key], input["pose"], input["intrinsics"], with_origins=True)
            return dirs.reshape(-1, 3), origins
        dirs, cam = rend_util.get_camera_params(input[key], input["pose"], input["intrinsics"])
        n = dirs.shape[1]
        return dirs.reshape(-1, 3), cam.unsqueeze(1).repeat(1, n, 1).reshape(-1, 3)

    def _z_vals(self, ray_dirs, cam_loc):
        if self.z_vals_override is not None:
            z = self.z_vals_override
            idx = self._cpu_random("eik_idx", lambda: torch.randint(z.shape[-1], (z.shape[0],)), z.device)
            if z.is_cuda and self.training and not self.junction_eikonal:
                return z, (z, idx)          # the eikonal-point launch picks z[r, idx[r]] itself (ops.eik_points)
            return z, z.gather(1, idx.unsqueeze(-1))
        return self.ray_sampler.get_z_vals(ray_dirs, cam_loc, self)

    def render_rgb(self, input):
        assert not self.training
        ray_dirs, cam_loc = self._rays(input)
        z_vals, _ = self._z_vals(ray_dirs, cam_loc)
        return self._render(cam_loc, ray_dirs, z_vals, False)[0]

    def forward(self, input):
        intrinsics, uv, pose = input["intrinsics"], input["uv"], input["pose"]
        # camera-only work of the whole forward in one launch: rays through uv and through uv_proj, [R | T] of pose^-1, the contiguous
        # intrinsics (round 6; before: camera_rays twice + camera_mats)
        setup = None
        if (uv.is_cuda and pose.shape == (1, 4, 4) and intrinsics.dtype == torch.float32 and intrinsics.stride(-1) == 1
                and intrinsics.shape[0] == 1 and "uv_proj" in input and input["uv_proj"].shape == uv.shape):
            setup = ops.camera_setup(uv, input["uv_proj"], pose, intrinsics)
            ray_dirs, cam_loc = setup[0], setup[1]
        else:
            ray_dirs, cam_loc = self._rays(input)
        n_rays = ray_dirs.shape[0]
        z_vals, z_eik = self._z_vals(ray_dirs, cam_loc)
        grad_theta = None
        if self.training and not self.junction_eikonal:
            # Eikonal points (rend_a :515-527) ride along with the main pass through the SDF network: no separate
            # latency-bound small launches.  The reference draws them after the junction block, but no random draw
            # happens in between, so the CPU RNG stream order is unchanged.
            eik = self._eikonal_points(n_rays, cam_loc, ray_dirs, z_eik, None)
            rgb, lines3d, depth, xyz, weights, sdf_s, points, nmap, grad_theta = self._render(
                cam_loc, ray_dirs, z_vals, False, eik, with_eik=True)
        else:
            rgb, lines3d, depth, xyz, weights, sdf_s, points, nmap = self._render(cam_loc, ray_dirs, z_vals, not self.training)
        output = {"points": points, "rgb_values": rgb, "sdf": sdf_s, "depth": depth, "xyz": xyz}

        # ---- attraction field / junctions (rend_a :424-513); R-sized, stays in torch -------------------
        # Two pieces of this block do not feed the matching below and run on a side stream, concurrently with it: the
        # global junctions ffn(latents) (depends on parameters only; autograd runs its backward on that stream too, next
        # to the render backward) and get_outputs(points3d) with the l3d geometry (latency-bound R-point launches).
        points3d = xyz
        main = torch.cuda.current_stream() if xyz.is_cuda else None
        side = self._side_stream(xyz.device) if (main is not None and self.use_side_stream) else None
        if setup is not None:
            w2c, K3 = setup[3], setup[4]
        elif xyz.is_cuda and pose.shape[1:] == (4, 4) and intrinsics.dtype == torch.float32 and intrinsics.stride(-1) == 1:
            w2c, K3 = ops.camera_mats(pose[0], intrinsics[0])     # [R | T] of pose^-1 and the contiguous 3x3 intrinsics: one launch, no sync
        else:
            w2c = ops.inv_small(pose[0])[:3]                 # one launch, no host-side singularity check, no sync
            K3 = intrinsics[0, :3, :3]
        Rm, T = w2c[:, :3], w2c[:, 3:]
        eye = _eye3(K3.device)

        def l3d_block():
            if points3d.is_cuda:          # get_outputs(points3d) (rend_a :441-443) without the row-major copies of lin8's output (one
                # launch): the junction block reads the sdf and the normal only; both keep their graph to the parameters
                net = self.implicit_network
                p3_sdf, p3_grad = ops.sdf_point_normals(net.handle(), points3d, net.sdf_bounding_sphere, net.sphere_scale)
            else:
                p3_sdf, _, p3_grad = self.implicit_network.get_outputs(points3d)
            l_dirs, l_orig = (setup[2], cam_loc) if setup is not None else self._rays(input, "uv_proj")
            if points3d.is_cuda:          # plane intersection per ray: one launch
                l3d = ops.l3d_points(points3d, l_orig, l_dirs, p3_grad)
            else:
                den = (l_dirs * p3_grad).sum(-1)
                den = den + torch.where(den >= 0, torch.full_like(den, 1e-6), torch.full_like(den, -1e-6))
                t = (((points3d - l_orig) * p3_grad).sum(-1) / den).detach()
                l3d = l_orig + l_dirs * t.unsqueeze(-1)
            l3d_score = None
            if self.training and self.use_l3d:      # only the l3d candidate filter reads the score (rend_a :455-468)
                with torch.no_grad():
                    a, b = l3d - lines3d[:, 0], l3d - lines3d[:, 1]
                    l3d_score = torch.linalg.cross(a, b).norm(dim=-1) / (lines3d[:, 0] - lines3d[:, 1]).norm(dim=-1)
            return p3_sdf, l3d, l3d_score

        j3d_global = None
        if side is not None:
            side_b = self._side_stream(xyz.device, 1)
            side.wait_stream(main)
            side_b.wait_stream(main)
            if self.training:
                if self.use_side_stream & 1:
                    with torch.cuda.stream(side):
                        j3d_global = self._global_junctions()
                    j3d_global.record_stream(main)
                else:
                    j3d_global = self._global_junctions()
            if self.use_side_stream & 2:
                with torch.cuda.stream(side_b):
                    p3_sdf, l3d, l3d_score = l3d_block()
                for tns in (p3_sdf, l3d, l3d_score):
                    if tns is not None:
                        tns.record_stream(main)
            else:
                p3_sdf, l3d, l3d_score = l3d_block()
                side_b = None
            if self.training and self.use_l3d and side_b is not None:
                main.wait_stream(side_b)
                side_b = None
        else:
            side_b = None
            if self.training:
                j3d_global = self._global_junctions()
            p3_sdf, l3d, l3d_score = l3d_block()
        if xyz.is_cuda:        # one launch per projection (forward / backward) instead of ~16 R-sized torch kernels
            K3c, w2c3 = K3.contiguous(), w2c.contiguous()
            K3 = K3c           # the loss inverts output["K"]: hand it the contiguous copy
            proj = lambda Kc, X: ops.project2d(Kc, w2c3, X)
            # the same points with K (pixels) and with the identity (calibrated): one launch for both (round 6)
            proj_pair = lambda X: ops.project2d_pair(K3c, eye, w2c3, X)
        else:
            K3c = K3
            proj = lambda Kc, X: self.project2D(Kc, Rm, T, X)
            proj_pair = lambda X: (proj(K3c, X), proj(eye, X))
        if xyz.is_cuda:
            lines2d, lines2d_calib = proj_pair(lines3d)
            lines2d = lines2d.detach()      # (rend_a :436: the pixel projection is taken of lines3d.detach())
        else:
            lines2d = proj(K3c, lines3d.detach())
            lines2d_calib = proj(eye, lines3d)
        if self.training:
            cand_valid = None
            if self.dbscan_enabled:
                if lines3d.is_cuda and 2 <= 2 * n_rays <= ops.DBSCAN_MAX_POINTS:
                    # DBSCAN(eps = 0.01, min_samples = 2) + cluster means on the device: padded centres + validity mask,
                    # no host round trip (reference: sklearn on the host, :328-339, :460)
                    cand3d, cand_valid, _ = ops.dbscan_means(lines3d.detach().reshape(-1, 3), 0.01)
                else:
                    cand3d = self.cluster_dbscan(lines3d.detach().cpu().numpy().reshape(-1, 3), eps=0.01, min_samples=2)
            elif self.use_l3d:
                thr = l3d_score.median().clamp_min(0.01)
                keep = l3d_score < thr          # data-dependent shape: one host sync, as the reference (:465-468)
                cand3d = torch.cat([lines3d[keep].detach().reshape(-1, 3), l3d[keep]], 0)
            else:
                cand3d = lines3d.detach().reshape(-1, 3)
            if self.dbscan_enabled or self.use_l3d:
                cand2d, cand2d_calib = proj_pair(cand3d)
            else:      # the candidates ARE the line end points: their projections were just computed (same arithmetic, same values)
                cand2d = lines2d.reshape(-1, 2)
                cand2d_calib = lines2d_calib.detach().reshape(-1, 2)
            gt2d = _device_copy(input["wireframe"][0], "vertices", cand2d.device)
            fused_gate = cand2d.is_cuda and min(gt2d.shape[0], cand2d.shape[0]) <= 2048
            cost = ops.junction_cost(cand2d, gt2d) if fused_gate else ((cand2d[None] - gt2d[:, None]) ** 2).sum(-1).sqrt()
            # Hungarian matching on the device (reference: scipy on the host, :473).  Without a candidate mask every gt
            # junction / candidate of the smaller side is matched, so the pair count min(V, C) is static; with the padded
            # DBSCAN centres the pairs beyond the device-side count come back as -1 and are masked out
            rows, cols, _ = ops.linear_sum_assignment(cost, None, cand_valid)
            if fused_gate:
                # matched costs, median / 10 px gate and the gathered matched candidates: one launch (:474-489)
                median, good, j3_pad, j2_pad, j2c_pad = ops.junction_gate(rows, cols, cost, cand3d, cand2d, cand2d_calib, self.use_median)
                if self.use_median:
                    output["median"] = median
            else:
                if cand_valid is None:
                    matched = cost[rows, cols]
                    pair_ok = None
                else:
                    pair_ok = rows >= 0
                    rows, cols = rows.clamp_min(0), cols.clamp_min(0)
                    matched = torch.where(pair_ok, cost[rows, cols], torch.full_like(cost[rows, cols], float("nan")))
                if self.use_median:
                    if matched.numel() == 0:
                        median = matched.new_tensor(10.0)
                    elif pair_ok is None:
                        median = matched.detach().median()
                    else:
                        median = torch.nanmedian(matched.detach())
                        median = torch.where(torch.isnan(median), torch.full_like(median, 10.0), median)
                    good = matched < median
                    output["median"] = median
                else:
                    good = matched < 10
                if pair_ok is not None:
                    good = good & pair_ok
                j2_pad, j3_pad, j2c_pad = cand2d[cols], cand3d[cols], cand2d_calib[cols]
            if side is not None:
                main.wait_stream(side)          # join: the global junctions are needed from here on
            # the reference compacts with `[good]` (:478-489), a data-dependent shape; here the matched candidates stay
            # padded + mask (what neat_amd.loss reads) and the compact tensors are built only if somebody asks for them
            output = JunctionOutputs(output, good, {"j2d_local": j2_pad, "j3d_local": j3_pad, "j2d_local_calib": j2c_pad})
            output["j3d_global"] = j3d_global
            output["j2d_global"], output["j2d_global_calib"] = proj_pair(j3d_global)
            if xyz.is_cuda:
                output.calib_proj = {"w2c": w2c3, "lines3d": lines3d, "lines2d_calib": lines2d_calib, "j3d_global": j3d_global,
                                     "j2d_global_calib": output["j2d_global_calib"]}
        if side_b is not None:
            main.wait_stream(side_b)            # join: get_outputs(points3d) / l3d ran next to the matching above
        elif side is not None and not self.training:
            main.wait_stream(side)
        output["l3d"] = l3d
        output["points3d"] = points3d
        output["lines3d"] = lines3d
        output["lines2d_calib"] = lines2d_calib
        output["lines2d"] = lines2d
        output["sdf"] = p3_sdf.flatten()
        output["wireframe-gt"] = input["wireframe"]
        output["K"] = K3

        if self.training:
            if grad_theta is None:
                grad_theta = self._eikonal(n_rays, cam_loc, ray_dirs, z_eik, output["j3d_global"].detach())
            output["grad_theta"] = grad_theta
        else:
            output["normal_map"] = nmap
        return output

    def _global_junctions(self):
        """ffn(latents) (rend_a :491).  The shipped architecture (Linear-ReLU-Linear-ReLU-Linear, 256 wide) on CUDA goes
        through three HIP launches; anything else through the torch modules."""
        lin = [m for m in self.ffn if isinstance(m, nn.Linear)]
        if (self.latents.is_cuda and len(self.ffn) == 5 and len(lin) == 3 and self.latents.shape[1] == 256
                and tuple(lin[0].weight.shape) == (256, 256) and tuple(lin[1].weight.shape) == (256, 256)
                and tuple(lin[2].weight.shape) == (3, 256)):
            return ops.ffn_junctions(self.latents, lin)
        return self.ffn(self.latents)

    def _side_stream(self, device, idx=0):
        key = (str(device), idx)
        if key not in self._side:
            self._side[key] = torch.cuda.Stream(device=device)
        return self._side[key]

    def _cpu_random(self, name, draw, device):
        """Randoms are drawn on the CPU (the reference's RNG stream).  Normally: draw + asynchronous pinned copy.  With
        `static_randoms` set (a dict; HIP-graph capture / replay, neat_amd.train.Trainer): the forward reads a persistent
        device tensor per draw site, which the trainer refills (same draw order) before every replay."""
        if self.static_randoms is None:
            return _to_device_async(draw(), device, name)
        slot = self.static_randoms.get(name)
        if slot is None:
            t = draw()
            slot = {"draw": draw, "dev": t.to(device), "order": len(self.static_randoms)}
            self.static_randoms[name] = slot
        return slot["dev"]

    def _eikonal_points(self, n_rays, cam_loc, ray_dirs, z_eik, junctions):
        """Eikonal points: uniform in the bounding cube + one near-surface sample per ray (rend_a :515-527)."""
        r = self.scene_bounding_sphere
        eik = self._cpu_random("eik_uniform", lambda: torch.empty(n_rays, 3).uniform_(-r, r), ray_dirs.device)
        if isinstance(z_eik, tuple):          # (depths [R,S], drawn index per ray): given depth samples (_z_vals)
            if eik.is_cuda:
                return ops.eik_points(eik, cam_loc, ray_dirs, None, junctions, z=z_eik[0], idx=z_eik[1])
            z_eik = z_eik[0].gather(1, z_eik[1].unsqueeze(-1))
        if eik.is_cuda:
            return ops.eik_points(eik, cam_loc, ray_dirs, z_eik, junctions)       # [uniform | o + z d | junctions]: one launch
        eik = torch.cat([eik, torch.addcmul(cam_loc, z_eik, ray_dirs)], 0)
        if junctions is not None:
            eik = torch.cat([eik, junctions], 0)
        return eik

    def _eikonal(self, n_rays, cam_loc, ray_dirs, z_eik, junctions):
        return self.implicit_network.gradient(self._eikonal_points(n_rays, cam_loc, ray_dirs, z_eik, junctions))
