"""Depth samplers with the reference's names (code/model/ray_sampler.py).

UniformSampler, sample_pdf / get_z_vals_fine and the ErrorBoundSampler (VolSDF Algorithm 1) run in HIP on CUDA tensors:
neat_uniform_depths, neat_sample_pdf, and for Algorithm 1 the fused SDF values kernel (128..640 SDF queries per ray, the dominant
cost) plus the per-ray bound / resample / finish kernels (neat_sampler_*), driven either with the reference's control flow
(one host sync per round) or with the decision kept on the device (get_z_vals_device).  The torch formulations in this file are
the CPU path and the cross-checks of the gpu tests.
Random draws are made on the CPU generator in the reference's order and moved to the device,
so a seeded run consumes the RNG stream exactly as the reference does.
"""
import math
import os

import torch


def _lerp_inverse_cdf(bins, cdf, u):
    idx = torch.searchsorted(cdf, u, right=True)
    lo = (idx - 1).clamp_(min=0)
    hi = idx.clamp_(max=cdf.shape[-1] - 1)
    c_lo, c_hi = cdf.gather(1, lo), cdf.gather(1, hi)
    b_lo, b_hi = bins.gather(1, lo), bins.gather(1, hi)
    span = c_hi - c_lo
    span = torch.where(span < 1e-5, torch.ones_like(span), span)
    return b_lo + (u - c_lo) / span * (b_hi - b_lo)


def _pdf_to_cdf(pdf):
    pdf = pdf / pdf.sum(-1, keepdim=True)
    # torch-CPU (where the reference's sampler math was defined) accumulates a float cumsum in double and rounds each knot once;
    # do the same on the device -- an fp32 scan moves knots by a few 1e-8, enough to flip the `denom < 1e-5` rule below for the
    # empty bins of `weights + 1e-5` (see tests/test_oracle_golden.py::test_sampler_is_ill_conditioned)
    return torch.cat([torch.zeros_like(pdf[:, :1]), pdf.double().cumsum(-1).float()], -1)


_DET_U = {}


def sample_pdf(bins, weights, N_samples, det=False, pytest=False, merge_with=None, model=None):
    """NeRF inverse-CDF sampling (ray_sampler.py:16-59).  CUDA tensors: one HIP launch (neat_sample_pdf), which also produces the
    sorted union with `merge_with` (get_z_vals_fine); returns samples, or (samples, sorted union) when merge_with is given."""
    if pytest:
        raise NotImplementedError("the reference's pytest branch is broken (NameError on np); not reproduced")
    shape = list(weights.shape[:-1]) + [N_samples]
    if weights.is_cuda:
        from . import ops
        if det:                       # the CPU linspace of the reference, uploaded once per (N, device)
            key = (N_samples, str(weights.device))
            if key not in _DET_U:
                _DET_U[key] = torch.linspace(0.0, 1.0, N_samples).to(weights.device)
            u = _DET_U[key]
        else:
            u = _draw(model, "sample_pdf_u", lambda: torch.rand(shape), weights.device)
        samples, merged = ops.sample_pdf(bins, weights, u, merge_with)
        return samples if merge_with is None else (samples, merged)
    u = torch.linspace(0.0, 1.0, N_samples) if det else torch.rand(shape)
    cdf = _pdf_to_cdf(weights + 1e-5)
    samples = _lerp_inverse_cdf(bins, cdf, u.expand(shape).contiguous().to(weights.device))
    if merge_with is None:
        return samples
    return samples, torch.sort(torch.cat([merge_with, samples], -1), -1)[0]


def _draw(model, name, fn, dev):
    """A CPU draw moved to the device; through the model's draw-site registry when it has one (HIP-graph replay refills the
    persistent device tensor of every site, networks.VolSDFNetwork._cpu_random)."""
    hook = getattr(model, "_cpu_random", None) if model is not None else None
    return hook(name, fn, dev) if hook is not None else fn().to(dev)


_UNIT_GRIDS = {}


def _unit_grid(n, device):
    """torch.linspace(0, 1, n) on the device, made once per (n, device): the refine rounds' u (ray_sampler.py:226) is a constant."""
    key = (int(n), str(device))
    g = _UNIT_GRIDS.get(key)
    if g is None:
        g = _UNIT_GRIDS[key] = torch.linspace(0.0, 1.0, n, device=device)
    return g


def _fast(model):
    """Keyword for get_sdf_vals: the model's opt-in to one-product f16 SDF queries in the sampler (conf key
    model.hip_sampler_fast_values, fp16x3 only).  A reference-shaped model object without the attribute gets none."""
    return {"fast": True} if getattr(model, "sampler_fast_values", False) else {}


class RaySampler:
    def __init__(self, near, far):
        self.near, self.far = near, far

    def get_z_vals(self, ray_dirs, cam_loc, model):
        raise NotImplementedError


class UniformSampler(RaySampler):
    def __init__(self, scene_bounding_sphere, near, N_samples, N_important=0, take_sphere_intersection=False, far=-1):
        super().__init__(near, 2.0 * scene_bounding_sphere if far == -1 else far)
        self.N_samples = N_samples
        self.N_important = N_important
        self.scene_bounding_sphere = scene_bounding_sphere
        self.take_sphere_intersection = take_sphere_intersection

    def get_z_vals(self, ray_dirs, cam_loc, model):
        dev, R = ray_dirs.device, ray_dirs.shape[0]
        far = None
        if self.take_sphere_intersection:
            from . import rend_util
            far = rend_util.get_sphere_intersections(cam_loc, ray_dirs, r=self.scene_bounding_sphere)[:, 1:]
        rnd = None
        if model.training:                                   # stratified jitter inside each bin (:81-89)
            shape = (R, self.N_samples)
            rnd = _draw(model, "uniform_jitter", lambda: torch.rand(shape), dev)
        if ray_dirs.is_cuda:                                 # one launch (neat_uniform_depths), bit-identical to the op chain below
            from . import ops
            z = ops.uniform_depths(R, self.N_samples, float(self.near), far if far is not None else float(self.far), rnd, dev)
        else:
            near = torch.full((R, 1), float(self.near), device=dev)
            if far is None:
                far = torch.full((R, 1), float(self.far), device=dev)
            t = torch.linspace(0.0, 1.0, self.N_samples, device=dev)
            z = near * (1.0 - t) + far * t
            if rnd is not None:
                mid = 0.5 * (z[:, 1:] + z[:, :-1])
                hi = torch.cat([mid, z[:, -1:]], -1)
                lo = torch.cat([z[:, :1], mid], -1)
                z = lo + (hi - lo) * rnd
        if getattr(model, "static_randoms", None) is None:
            torch.randint(z.shape[-1], (R,))                 # the reference draws (and discards) an index here (:91)
        else:                                                # graph replay: keep the draw in the refill sequence
            n = z.shape[-1]
            _draw(model, "uniform_dropped_idx", lambda: torch.randint(n, (R,)), dev)
        return z

    def get_z_vals_fine(self, z_vals, weights, model):
        assert self.N_important > 0
        mid = 0.5 * (z_vals[..., 1:] + z_vals[..., :-1])
        _, merged = sample_pdf(mid, weights[..., 1:-1], self.N_important, det=model.training, merge_with=z_vals, model=model)   # det is inverted vs NeRF
        return merged.detach()


class HierarchicalSampler(RaySampler):
    """BASELINE config C5: NeRF-style hierarchical sampling, N_coarse stratified depths + N_fine importance samples, feeding the
    main pass.  The reference ships the two pieces (`UniformSampler.get_z_vals`, `get_z_vals_fine` / `sample_pdf`,
    ray_sampler.py:16-106) but no model calls them; they are composed here the way SURVEY 8(d) defines C5 and the way
    tests/golden/make_golden.py composes the reference's own functions for fixture G12:
        coarse depths -> no-grad SDF values (fused HIP chain) -> volume_rendering weights (HIP scan) -> fine depths by inverse CDF
        (`det = model.training`, the reference's inverted flag) -> sorted union; z_eik = one of the depths per ray (randint).
    CPU random draws in the reference's order: rand [R, N_coarse] (training only), randint (drawn and dropped by get_z_vals),
    [rand [R, N_fine] in eval mode], randint (eikonal index).  Selected with the optional conf key model.hip_sampler = hierarchical."""

    def __init__(self, scene_bounding_sphere, near, N_coarse, N_fine):
        super().__init__(near, 2.0 * scene_bounding_sphere)
        self.uniform_sampler = UniformSampler(scene_bounding_sphere, near, N_coarse, N_important=N_fine)
        self.N_samples, self.N_fine = N_coarse, N_fine
        self.last_rounds = 0

    def get_z_vals(self, ray_dirs, cam_loc, model):
        zc = self.uniform_sampler.get_z_vals(ray_dirs, cam_loc, model)
        pts = cam_loc.unsqueeze(1) + zc.unsqueeze(2) * ray_dirs.unsqueeze(1)
        with torch.no_grad():
            sdf = model.implicit_network.get_sdf_vals(pts.reshape(-1, 3), **_fast(model))
            w = model.volume_rendering(zc, sdf)
        z = self.uniform_sampler.get_z_vals_fine(zc, w, model)
        n, R = z.shape[-1], z.shape[0]
        idx = _draw(model, "eik_idx", lambda: torch.randint(n, (R,)), z.device)
        return z, torch.gather(z, 1, idx.unsqueeze(-1))


class ErrorBoundSampler(RaySampler):
    def __init__(self, scene_bounding_sphere, near, N_samples, N_samples_eval, N_samples_extra, eps, beta_iters,
                 max_total_iters, inverse_sphere_bg=False, N_samples_inverse_sphere=0, add_tiny=0.0):
        super().__init__(near, 2.0 * scene_bounding_sphere)
        if inverse_sphere_bg:
            raise NotImplementedError("inverse_sphere_bg is off in every shipped conf and is not implemented")
        self.N_samples, self.N_samples_eval, self.N_samples_extra = N_samples, N_samples_eval, N_samples_extra
        self.eps, self.beta_iters, self.max_total_iters = eps, beta_iters, max_total_iters
        self.scene_bounding_sphere, self.add_tiny = scene_bounding_sphere, add_tiny
        self.inverse_sphere_bg = False
        self.uniform_sampler = UniformSampler(scene_bounding_sphere, near, N_samples_eval)
        self.last_rounds = 0          # refinement rounds taken by the last call (reported by bench.py)
        # (1 / (4 log(1 + eps))) of the initial beta (:143), evaluated once in fp32 on the host: a device-side torch.tensor(eps + 1)
        # would be a pageable upload per call (a sync, and not capturable)
        self._beta_c = float(1.0 / (4.0 * torch.log(torch.tensor(self.eps + 1.0))))
        self.sync_free = False        # True: get_z_vals_device (control flow on the device, HIP-graph capturable)
        self._ctl = None              # device control words of the last sync-free call
        self._pick_all = self._pick_direct = None

    # ----- pieces of Algorithm 1 --------------------------------------------------------------
    @staticmethod
    def _interval_bound(d, gap):
        """d* of Theorem 1 per interval (:161-173)."""
        b, c = d[:, :-1].abs(), d[:, 1:].abs()
        near_left = gap * gap + b * b <= c * c
        near_right = gap * gap + c * c <= b * b
        s = (gap + b + c) / 2.0
        heron = 2.0 * torch.sqrt(s * (s - gap) * (s - b) * (s - c)) / gap
        inside = ~near_left & ~near_right & (b + c - gap > 0)
        out = torch.where(near_left, b, torch.zeros_like(gap))
        out = torch.where(near_right, c, out)
        out = torch.where(inside, heron, out)
        return (d[:, 1:].sign() * d[:, :-1].sign() == 1) * out

    def get_error_bound(self, beta, model, sdf, z_vals, dists, d_star):
        """max_i of the opacity error bound (:285-293)."""
        sigma = model.density(sdf.reshape(z_vals.shape), beta=beta)
        opt_depth = torch.cat([torch.zeros_like(dists[:, :1]), dists * sigma[:, :-1]], -1).cumsum(-1)
        err = (torch.exp(-d_star / beta) * dists ** 2.0 / (4.0 * beta ** 2)).cumsum(-1)
        return ((torch.exp(err).clamp(max=1.0e6) - 1.0) * torch.exp(-opt_depth[:, :-1])).max(-1)[0]

    def get_z_vals_device(self, ray_dirs, cam_loc, model):
        """Algorithm 1 without host synchronisation: always `max_total_iters` rounds of launches, the batch-global test of :200 kept
        in device memory (include/neat_hip.h "a3 without host synchronisation"); the launches after the final round return at once.
        Same arithmetic as get_z_vals.  Differences, all in the random draws: the final u [R, N] is drawn before the rounds instead of
        in the last one (same position in the CPU stream: the refine rounds draw nothing), and the training-mode
        `randperm(n)[:N_extra]` -- whose n is only known on the device -- becomes "the N_extra smallest of n uniform keys", the same
        distribution from a different stretch of the stream.  Eval mode draws nothing here and is bit-identical to get_z_vals.
        Round 6: with the HIP SDF network a round is TWO launches -- the fused SDF query and neat_sampler_round (bound + resampling + the
        next round's query points, written straight into the query's workspace) -- instead of four (`_get_z_vals_device_unfused`,
        kept for model objects without the HIP entry points); the training-mode picks are computed beside the prologue."""
        from . import ops
        net = getattr(model, "implicit_network", None)
        density = getattr(model, "density", None)
        if not (hasattr(net, "get_sdf_vals_rays") and hasattr(net, "handle") and type(density).__name__ == "LaplaceDensity"
                and hasattr(density, "beta_min") and os.environ.get("NEAT_SAMPLER_UNFUSED") != "1"):
            return self._get_z_vals_device_unfused(ray_dirs, cam_loc, model)
        dev, R = ray_dirs.device, ray_dirs.shape[0]
        K, Ne, N = self.max_total_iters, self.N_samples_eval, self.N_samples
        z = self.uniform_sampler.get_z_vals(ray_dirs, cam_loc, model)
        u_refine = _unit_grid(Ne, dev)
        u_final = _draw(model, "sampler_u", lambda: torch.rand(R, N), dev) if model.training else _unit_grid(N, dev)
        n_extra = max(self.N_samples_extra, 0)
        keys = _draw(model, "sampler_keys", lambda: torch.rand(Ne * K), dev) if (n_extra and model.training) else None
        handle, fast = net.handle(), bool(getattr(model, "sampler_fast_values", False))
        ws, ldp = ops.sdf_query_workspace(handle, R * Ne, dev, fast)      # one workspace for all rounds: every query has R * Ne points
        o, d = cam_loc.detach().float().contiguous(), ray_dirs.detach().float().contiguous()
        beta0, beta, ctl, pick_all = ops.sampler_init_rays(z, density.beta, density.beta_min, self._beta_c, 2 * K + 1, o, d, ws, ldp,
                                                           keys, Ne, K, n_extra)
        samples, z_final = torch.empty(R, N, device=dev), torch.empty(R, Ne * K, device=dev)
        order, sdf = None, None
        for k in range(K):
            gate = (ctl, k - 1, 1) if k > 0 else None                     # open[k-1]: the previous round left a ray above beta0
            new_sdf = ops.sdf_values_laid_out(handle, ws, R * Ne, net.sdf_bounding_sphere, net.sphere_scale, gate=gate, fast=fast).reshape(R, Ne)
            sdf, beta, _, z_next, order_next = ops.sampler_round(z, sdf, new_sdf, order, beta, beta0, self.eps, self.beta_iters, self.add_tiny,
                                                                 u_refine, u_final, samples, z_final, ctl, k, K, o, d, ws, ldp)
            z, order = z_next, order_next
        self._ctl, self._pick_all, self._pick_direct = ctl, pick_all, None
        self.last_rounds = None                              # read lazily: rounds_taken()
        n_out = N + 2 + n_extra
        eik_idx = _draw(model, "sampler_eik_idx", lambda: torch.randint(n_out, (R,)).to(torch.int32), dev)
        return ops.sampler_finish_picked(samples, z_final, ctl, K, pick_all, Ne, n_extra, self.near, self.far, eik_idx)

    @property
    def last_pick(self):
        """Grid indices the last device-decided call took as its extra samples (one device read; tests)."""
        if getattr(self, "_pick_direct", None) is not None:
            return self._pick_direct
        K, Ne = self.max_total_iters, self.N_samples_eval
        n = int(self._ctl[2 * K].item())
        if self._pick_all is None:
            return torch.linspace(0, n - 1, self.N_samples_extra).long()
        return self._pick_all[n // Ne - 1]

    def _get_z_vals_device_unfused(self, ray_dirs, cam_loc, model):
        """The round-2..5 form of get_z_vals_device: per round the SDF query, neat_sampler_bound_dev and neat_sampler_resample_dev."""
        from . import ops
        dev, R = ray_dirs.device, ray_dirs.shape[0]
        K, Ne, N = self.max_total_iters, self.N_samples_eval, self.N_samples
        z = self.uniform_sampler.get_z_vals(ray_dirs, cam_loc, model)
        density = getattr(model, "density", None)
        if type(density).__name__ == "LaplaceDensity" and hasattr(density, "beta_min"):
            # beta0 = |beta| + beta_min, the rays' initial beta (Lemma 2) and the zeroed control words: one launch (neat_sampler_init)
            beta0, beta, ctl = ops.sampler_init(z, density.beta, density.beta_min, self._beta_c, 2 * K + 1)
        else:
            beta0 = model.density.get_beta().detach()
            gap = z[:, 1:] - z[:, :-1]
            beta = torch.sqrt(self._beta_c * (gap ** 2.0).sum(-1))
            ctl = torch.zeros(2 * K + 1, dtype=torch.int32, device=dev)
        u_refine = _unit_grid(Ne, dev)
        u_final = _draw(model, "sampler_u", lambda: torch.rand(R, N), dev) if model.training else _unit_grid(N, dev)
        samples, z_final = torch.empty(R, N, device=dev), torch.empty(R, Ne * K, device=dev)
        fresh, order, sdf = z, None, None
        on_rays = getattr(model.implicit_network, "get_sdf_vals_rays", None)
        cam, dirs = cam_loc.unsqueeze(1), ray_dirs.unsqueeze(1)
        for k in range(K):
            gate = (ctl, K + k - 1, 1) if k > 0 else None
            with torch.no_grad():
                if on_rays is not None:
                    new_sdf = on_rays(cam_loc, ray_dirs, fresh, gate=gate, **_fast(model)).reshape(R, -1)
                else:
                    pts = torch.addcmul(cam, fresh.unsqueeze(2), dirs).reshape(-1, 3)
                    new_sdf = model.implicit_network.get_sdf_vals(pts, gate=gate, **_fast(model)).reshape(R, -1)
            sdf, beta, fresh, z_next, order_next = ops.sampler_round_dev(z, sdf, new_sdf, order, beta, beta0, self.eps, self.beta_iters,
                                                                         self.add_tiny, u_refine, u_final, samples, z_final, ctl, k, K)
            z, order = z_next, order_next
        self._ctl = ctl
        self.last_rounds = None                              # read lazily: rounds_taken()
        n_extra = max(self.N_samples_extra, 0)
        keys = _draw(model, "sampler_keys", lambda: torch.rand(Ne * K), dev) if (n_extra and model.training) else None
        n_out = N + 2 + n_extra
        eik_idx = _draw(model, "sampler_eik_idx", lambda: torch.randint(n_out, (R,)).to(torch.int32), dev)
        self._pick_all = None
        z_vals, z_eik, self._pick_direct = ops.sampler_finish_dev(samples, z_final, ctl, K, keys, n_extra, self.near, self.far, eik_idx)
        return z_vals, z_eik

    @staticmethod
    def _eval_on_device(model):
        """Inference takes the device-decided rounds by default: in eval mode they draw nothing and are bit-identical to the
        host-decided ones, minus one host round trip per round.  (bf16 build only: the fp32 build's SDF kernels are not gated, a
        sampler that converged early would still evaluate all rounds.)"""
        if model.training:
            return False
        net = getattr(model, "implicit_network", None)
        handle = getattr(net, "handle", None)
        return handle is not None and handle().precision in (1, 3, 4)      # bf16 / f16 / f16x3 builds

    def rounds_taken(self):
        """Rounds of Algorithm 1 the last call ran (one device read when the last call was sync-free)."""
        if self.last_rounds is None and self._ctl is not None:
            K = self.max_total_iters
            self.last_rounds = int((self._ctl[K:2 * K] != 0).sum().item())
        return self.last_rounds

    def get_z_vals(self, ray_dirs, cam_loc, model):
        """Algorithm 1 with the per-ray work in HIP (neat_sampler_* kernels, one wavefront per ray) and the MLP queries in
        the SDF kernels.  Control flow, the one host sync per round and the CPU random draws follow the reference."""
        from . import ops
        if ray_dirs.is_cuda and (self.sync_free or self._eval_on_device(model)):
            return self.get_z_vals_device(ray_dirs, cam_loc, model)
        dev, R = ray_dirs.device, ray_dirs.shape[0]
        z = self.uniform_sampler.get_z_vals(ray_dirs, cam_loc, model)
        fresh, order, sdf = z, None, None
        density = getattr(model, "density", None)
        on_rays = getattr(model.implicit_network, "get_sdf_vals_rays", None) if ray_dirs.is_cuda else None
        if ray_dirs.is_cuda and type(density).__name__ == "LaplaceDensity" and hasattr(density, "beta_min"):
            beta0, beta, _ = ops.sampler_init(z, density.beta, density.beta_min, self._beta_c, 0)      # (as get_z_vals_device: same bits)
        else:
            beta0 = model.density.get_beta().detach()
            gap = z[:, 1:] - z[:, :-1]
            beta = torch.sqrt(self._beta_c * (gap ** 2.0).sum(-1))
        flag = torch.zeros(1, dtype=torch.int32, device=dev)
        rounds, open_ = 0, True
        while open_ and rounds < self.max_total_iters:
            with torch.no_grad():
                if on_rays is not None:
                    new_sdf = on_rays(cam_loc, ray_dirs, fresh, **_fast(model)).reshape(R, -1)
                else:
                    pts = (cam_loc.unsqueeze(1) + fresh.unsqueeze(2) * ray_dirs.unsqueeze(1)).reshape(-1, 3)
                    new_sdf = model.implicit_network.get_sdf_vals(pts, **_fast(model)).reshape(R, -1)
            flag.zero_()
            sdf, beta = ops.sampler_bound(z, sdf, new_sdf, order, beta, beta0, self.eps, self.beta_iters, flag)
            rounds += 1
            open_ = bool(flag.item())                       # batch-global `beta.max() > beta0`, one host sync per round (:200)
            refine = open_ and rounds < self.max_total_iters
            n = self.N_samples_eval if refine else self.N_samples
            if refine or not model.training:
                u = torch.linspace(0.0, 1.0, n, device=dev)
            else:
                u = _draw(model, "sampler_u", lambda: torch.rand(R, n), dev)
            fresh, zm, order = ops.sampler_resample(z, sdf, beta, u, refine, self.add_tiny)
            if refine:
                z = zm
        self.last_rounds = rounds
        pick = None
        if self.N_samples_extra > 0:
            if model.training:
                pick = torch.randperm(z.shape[1])[:self.N_samples_extra]
            else:
                pick = torch.linspace(0, z.shape[1] - 1, self.N_samples_extra).long()
            pick = _draw(model, "sampler_pick", lambda: pick.to(torch.int32), dev)
        n_out = self.N_samples + 2 + (self.N_samples_extra if self.N_samples_extra > 0 else 0)
        eik_idx = _draw(model, "sampler_eik_idx", lambda: torch.randint(n_out, (R,)).to(torch.int32), dev)
        return ops.sampler_finish(fresh, z, pick, self.near, self.far, eik_idx)

    # ---- torch-on-device formulation of the same algorithm (kept for cross-checking the kernels in the gpu tests) ----
    def get_z_vals_torch(self, ray_dirs, cam_loc, model):
        dev, R = ray_dirs.device, ray_dirs.shape[0]
        beta0 = model.density.get_beta().detach()
        z = self.uniform_sampler.get_z_vals(ray_dirs, cam_loc, model)
        fresh, order, sdf = z, None, None
        gap = z[:, 1:] - z[:, :-1]
        beta = torch.sqrt((1.0 / (4.0 * torch.log(torch.tensor(self.eps + 1.0, device=dev)))) * (gap ** 2.0).sum(-1))
        rounds, open_ = 0, True
        while open_ and rounds < self.max_total_iters:
            pts = (cam_loc.unsqueeze(1) + fresh.unsqueeze(2) * ray_dirs.unsqueeze(1)).reshape(-1, 3)
            with torch.no_grad():
                new_sdf = model.implicit_network.get_sdf_vals(pts, **_fast(model)).reshape(R, -1)
            sdf = new_sdf if order is None else torch.cat([sdf, new_sdf], -1).gather(1, order)
            gap = z[:, 1:] - z[:, :-1]
            d_star = self._interval_bound(sdf, gap)
            # per-ray bisection of beta on [beta0, beta] (:177-185)
            err = self.get_error_bound(beta0, model, sdf, z, gap, d_star)
            hi = torch.where(err <= self.eps, beta0.expand_as(beta), beta)
            lo = beta0.expand(R).clone()
            for _ in range(self.beta_iters):
                mid = (lo + hi) / 2.0
                err = self.get_error_bound(mid.unsqueeze(-1), model, sdf, z, gap, d_star)
                hi = torch.where(err <= self.eps, mid, hi)
                lo = torch.where(err > self.eps, mid, lo)
            beta = hi
            sigma = model.density(sdf, beta=beta.unsqueeze(-1))
            gap_inf = torch.cat([gap, torch.full((R, 1), 1e10, device=dev)], -1)
            energy = gap_inf * sigma
            trans = torch.exp(-torch.cat([torch.zeros(R, 1, device=dev), energy[:, :-1]], -1).cumsum(-1))
            weights = (1.0 - torch.exp(-energy)) * trans
            rounds += 1
            open_ = bool(beta.max() > beta0)               # batch-global test, one host sync per round (:200)
            refine = open_ and rounds < self.max_total_iters
            if refine:      # more samples where the error bound is large (:205-215)
                n = self.N_samples_eval
                err = (torch.exp(-d_star / beta.unsqueeze(-1)) * gap ** 2.0 / (4.0 * beta.unsqueeze(-1) ** 2)).cumsum(-1)
                cdf = _pdf_to_cdf((torch.exp(err).clamp(max=1.0e6) - 1.0) * trans[:, :-1] + self.add_tiny)
            else:           # final set from the rendering weights (:217-226)
                n = self.N_samples
                cdf = _pdf_to_cdf(weights[:, :-1] + 1e-5)
            if refine or not model.training:
                u = torch.linspace(0.0, 1.0, n, device=dev).unsqueeze(0).repeat(R, 1)
            else:
                u = torch.rand(R, n).to(dev)
            fresh = _lerp_inverse_cdf(z, cdf, u.contiguous())
            if refine:
                z, order = torch.sort(torch.cat([z, fresh], -1), -1)
        self.last_rounds = rounds
        ends = torch.tensor([float(self.near), float(self.far)], device=dev).expand(R, 2)
        if self.N_samples_extra > 0:
            if model.training:
                pick = torch.randperm(z.shape[1])[:self.N_samples_extra]
            else:
                pick = torch.linspace(0, z.shape[1] - 1, self.N_samples_extra).long()
            ends = torch.cat([ends, z[:, pick.to(dev)]], -1)
        z_out = torch.sort(torch.cat([fresh, ends], -1), -1)[0]
        eik_idx = torch.randint(z_out.shape[-1], (R,)).to(dev)
        return z_out, z_out.gather(1, eik_idx.unsqueeze(-1))
