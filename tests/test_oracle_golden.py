"""Pin the CPU oracle (oracle/neat_oracle.py) to golden vectors produced by the reference itself
(tests/golden/make_golden.py).  CPU only."""
import numpy as np
import pytest
import torch

from neat_amd import synth
from oracle import neat_oracle as O

T = torch.tensor


def params(variant, grad=False):
    return O.params_from_numpy(synth.synth_state_dict(42, variant), requires_grad=grad)


def close(a, b, tol, what=""):
    a = a.detach().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    err = np.abs(a - b).max() if a.size else 0.0
    assert a.shape == b.shape, (what, a.shape, b.shape)
    tol = tol * max(1.0, float(np.abs(b).max()) if b.size else 1.0)      # relative to the tensor's scale
    assert err <= tol, f"{what}: max abs err {err:.3e} > {tol}"


def test_posenc(golden):
    g = golden("g1_posenc")
    close(O.posenc(T(g["x"]), 6), g["pe6"], 0, "pe6")
    close(O.posenc(T(g["x"]), 4), g["pe4"], 0, "pe4")


@pytest.mark.parametrize("variant", ["init", "rough"])
def test_networks(golden, variant):
    g = golden(f"g2g3_networks_{variant}")
    p = params(variant)
    x, view = T(g["x"]), T(g["view"])
    close(O.sdf_forward(p, x), g["forward"], 2e-6, "forward")
    close(O.sdf_values(p, x), g["sdf_vals"], 2e-6, "sdf_vals")
    s, f, gr = O.sdf_outputs(p, x)
    close(s, g["out_sdf"], 2e-6, "sdf")
    close(f, g["out_feat"], 2e-6, "feat")
    close(gr, g["out_grad"], 2e-5, "grad")
    close(O.sdf_gradient(p, x), g["grad_raw"], 2e-5, "grad_raw")
    # sphere clamp must be active on some points and inactive on others
    assert (g["out_sdf"] != g["forward"][:, :1]).any() and (g["out_sdf"] == g["forward"][:, :1]).any()
    close(O.render_head(p, x, T(g["out_grad"]), view, T(g["out_feat"])), g["rgb"], 2e-6, "rgb")
    close(O.attraction_head(p, x, T(g["out_grad"]), view, T(g["out_feat"])), g["lines"], 5e-6, "lines")


def test_density_and_weights(golden):
    g = golden("g4_density")
    for b in (1e-3, 1e-2, 0.1):
        close(O.laplace_density(T(g["sdf"]), T(b)), g[f"sigma_{b:g}"], 0, f"sigma {b}")
    g = golden("g5_volume_rendering")
    sig = O.laplace_density(T(g["sdf"]).reshape(g["z"].shape), T(g["beta"]))
    close(O.free_energy_weights(T(g["z"]), sig)[0], g["weights"], 0, "weights")


def test_camera(golden):
    g = golden("g10_camera")
    for tag, K in (("", "K"), ("_skew", "K_skew")):
        d, c = O.camera_rays(T(g["uv"]), T(g["pose"]), T(g[K]))
        close(d, g["dirs" + tag], 1e-7, "dirs" + tag)
        close(c, g["cam" + tag], 0, "cam" + tag)


def test_hierarchical(golden):
    g = golden("g9_hierarchical")
    z, w = T(g["z_coarse"]), T(g["weights"])
    close(O.uniform_z(16, 64, 0.0, 6.0), g["z_coarse"], 0, "coarse")
    u = torch.linspace(0.0, 1.0, 64)[None].expand(16, 64)
    close(O.z_vals_fine(z, w, u), g["z_fine_det"], 0, "fine det")
    close(O.z_vals_fine(z, w, T(g["u_rand"])), g["z_fine_rand"], 0, "fine rand")


def _rays(g):
    d, o = O.camera_rays(T(g["uv"]), T(g["pose"]), T(g["intrinsics"]))
    d = d.reshape(-1, 3)
    return d, o.expand(d.shape[0], 3)


@pytest.mark.parametrize("variant", ["init", "rough"])
def test_sampler_eval(golden, variant):
    g = golden(f"g6_sampler_eval_{variant}")
    p = params(variant)
    d, o = _rays(g)
    tr = {}
    z, ze = O.error_bound_sampler(lambda x: O.sdf_values(p, x), O.beta_of(p), d, o, training=False,
                                  rand={"eik_idx": T(g["eik_idx"])}, trace=tr)
    close(z, g["z_vals"], 2e-5, "z_vals")
    close(ze, g["z_eik"], 2e-5, "z_eik")
    if variant == "rough":
        assert tr["rounds"] >= 2


@pytest.mark.parametrize("variant", ["init", "rough"])
def test_sampler_train(golden, variant):
    g = golden(f"g6_sampler_train_{variant}")
    p = params(variant)
    d, o = _rays(g)
    rand = {k: T(g[k]) for k in ("t_rand", "u_final", "perm", "eik_idx")}
    z, ze = O.error_bound_sampler(lambda x: O.sdf_values(p, x), O.beta_of(p), d, o, training=True, rand=rand)
    close(z, g["z_vals"], 2e-5, "z_vals")
    close(ze, g["z_eik"], 2e-5, "z_eik")


def _inp(g):
    return {k: T(g[k]) for k in ("intrinsics", "pose", "uv", "uv_proj")}


def _wf(g):
    v, e, w = T(g["wf_vertices"]), T(g["wf_edges"]), T(g["wf_weights"])
    ok = w > 0.97
    return torch.cat([v[e[ok, 0]], v[e[ok, 1]], w[ok, None]], -1), v


@pytest.mark.parametrize("variant", ["init", "rough"])
def test_full_forward_eval(golden, variant):
    g = golden(f"g7_forward_eval_{variant}")
    p = params(variant)
    lines, verts = _wf(g)
    out = O.full_forward(p, _inp(g), lines, verts, training=False, rand={"eik_idx": T(g["eik_idx"])})
    for k in ("points", "rgb_values", "depth", "xyz", "l3d", "points3d", "lines3d", "lines2d_calib", "sdf",
              "normal_map"):
        close(out[k], g["out_" + k], 5e-5, k)
    close(out["lines2d"], g["out_lines2d"], 2e-2, "lines2d (pixels)")


def test_train_step(golden):
    g = golden("g8_train_step_rough")
    p = params("rough", grad=True)
    lines, verts = _wf(g)
    rand = {k: T(g[k]) for k in ("t_rand", "u_final", "perm", "eik_idx", "eik_uniform")}
    out = O.full_forward(p, _inp(g), lines, verts, training=True, rand=rand)
    for k in ("rgb_values", "depth", "xyz", "l3d", "points3d", "lines3d", "lines2d_calib", "sdf", "grad_theta",
              "j3d_local", "j3d_global", "j2d_global_calib", "j2d_local_calib", "median"):
        close(out[k], g["out_" + k], 5e-5, k)
    lo = O.neat_loss(out, T(g["gt_rgb"]), T(g["gt_lines2d"]))
    for k in ("loss", "rgb_loss", "eikonal_loss", "line_loss", "j3d_loss", "j2d_loss"):
        close(lo[k].float(), g["loss_" + k], 2e-5, "loss " + k)
    assert int(lo["count"]) == int(g["loss_count"]) and int(lo["jcount"]) == int(g["loss_jcount"])
    lo["loss"].backward()
    from tests.golden.make_golden import GRAD_STRIDE
    worst = 0.0
    for k, v in p.items():
        gr = v.grad.reshape(-1).numpy()
        ref, (nrm, _) = g["grad_" + k], g["gradnorm_" + k]
        err = np.abs(gr[::GRAD_STRIDE] - ref).max()
        scale = max(np.abs(ref).max(), 1e-6)
        worst = max(worst, err / scale)
        assert err <= 2e-4 * scale + 1e-7, (k, err, scale)
        assert abs(np.sqrt((gr.astype(np.float64) ** 2).sum()) - nrm) <= 2e-4 * nrm + 1e-7, k
    print("worst relative grad err", worst)


def _check_train_step(g, p, out, lo, keys):
    for k in keys:
        close(out[k], g["out_" + k], 5e-5, k)
    for k in ("loss", "rgb_loss", "eikonal_loss", "line_loss", "j3d_loss", "j2d_loss"):
        close(lo[k].float(), g["loss_" + k], 2e-5, "loss " + k)
    assert int(lo["count"]) == int(g["loss_count"]) and int(lo["jcount"]) == int(g["loss_jcount"])
    lo["loss"].backward()
    from tests.golden.make_golden import GRAD_STRIDE
    for k, v in p.items():
        if "grad_" + k not in g:
            assert v.grad is None or float(v.grad.abs().max()) == 0.0, k
            continue
        gr = v.grad.reshape(-1).numpy()
        ref, (nrm, _) = g["grad_" + k], g["gradnorm_" + k]
        scale = max(np.abs(ref).max(), 1e-6)
        assert np.abs(gr[::GRAD_STRIDE] - ref).max() <= 2e-4 * scale + 1e-7, k
        assert abs(np.sqrt((gr.astype(np.float64) ** 2).sum()) - nrm) <= 2e-4 * nrm + 1e-7, k


def test_train_step_dtu_switches(golden):
    """G11: dbscan_enabled = True, use_median = False, 1024 junction latents (confs/dtu.conf, bmvs.conf; rend_a :333-342,460,475-482)."""
    g = golden("g11_train_step_dtu_switches")
    p = O.params_from_numpy(synth.synth_state_dict(42, "rough", num_junctions=1024), requires_grad=True)
    lines, verts = _wf(g)
    rand = {k: T(g[k]) for k in ("t_rand", "u_final", "perm", "eik_idx", "eik_uniform")}
    out = O.full_forward(p, _inp(g), lines, verts, training=True, rand=rand, use_median=False, dbscan_enabled=True)
    assert "median" not in out and out["j3d_global"].shape == (1024, 3)
    lo = O.neat_loss(out, T(g["gt_rgb"]), T(g["gt_lines2d"]))
    _check_train_step(g, p, out, lo, ("rgb_values", "depth", "xyz", "l3d", "points3d", "lines3d", "lines2d_calib", "sdf", "grad_theta",
                                      "j3d_local", "j3d_global", "j2d_global_calib", "j2d_local_calib"))


SWITCH_GOLDENS = [("g14_train_step_white_bkgd", dict(white_bkgd=True, bg_color=(1.0, 0.9, 0.8))),
                  ("g15_train_step_use_l3d", dict(use_l3d=True)),
                  ("g16_train_step_junction_eikonal", dict(junction_eikonal=True)),
                  ("g17_train_step_nerf_heads", dict(render_mode="nerf", attraction_mode="nerf")),
                  ("g18_train_step_inside_out", dict(inside_out=True))]


@pytest.mark.parametrize("name,switch", SWITCH_GOLDENS)
def test_train_step_model_switches(golden, name, switch):
    """G14-G16 (round 5): white_bkgd (rend_a :263-265,411-413), use_l3d (:461-465), junction_eikonal (:524-525) -- reference train steps.
    G17 / G18: heads with mode = 'nerf' (:180-181,240-241), SDF network with inside_out (:94-95)."""
    g = golden(name)
    sd = synth.synth_state_dict(42, "rough")
    if "render_mode" in switch:
        sd = synth.nerf_heads_state_dict(sd)
    p = O.params_from_numpy(sd, requires_grad=True)
    lines, verts = _wf(g)
    rand = {k: T(g[k]) for k in ("t_rand", "u_final", "perm", "eik_idx", "eik_uniform")}
    out = O.full_forward(p, _inp(g), lines, verts, training=True, rand=rand, **switch)
    close(out["z_vals"], g["z_vals"], 2e-5, "z_vals")
    if "junction_eikonal" in switch:
        assert out["grad_theta"].shape[0] == 2 * 64 + 64
    lo = O.neat_loss(out, T(g["gt_rgb"]), T(g["gt_lines2d"]))
    _check_train_step(g, p, out, lo, [k for k in ("rgb_values", "depth", "xyz", "l3d", "points3d", "lines3d", "lines2d_calib", "sdf", "grad_theta",
                                                  "j3d_local", "j3d_global", "j2d_global_calib", "j2d_local_calib", "median") if "out_" + k in g])


def test_white_bkgd_sums_are_ill_conditioned(golden):
    """Why the GPU test of G14 holds lin8.bias and density.beta to 5e-2 / 8e-2 instead of 2e-3: with white_bkgd the cotangent of every
    weight is d_rgb . (rgb_i - bg), and the two gradients that are plain sums of the sdf cotangents over all samples cancel to a small
    remainder.  The SAME algorithm (the oracle, pinned to the reference at 2e-4 in fp32 above) evaluated in fp64 moves them by 1.8 % and
    3.1 %: that is the reference's own rounding, not a property of any build."""
    g = golden("g14_train_step_white_bkgd")
    vals = {}
    for dt in (torch.float32, torch.float64):
      try:
        torch.set_default_dtype(dt)          # (the oracle creates its constants -- eye(3), epsilons -- in the default dtype)
        sd = synth.synth_state_dict(42, "rough")
        p = {k: torch.tensor(np.asarray(v), dtype=dt, requires_grad=True) for k, v in sd.items()}
        lines, verts = _wf(g)
        inp = {k: v.to(dt) for k, v in _inp(g).items()}
        out = O.full_forward(p, inp, lines.to(dt), verts.to(dt), training=True,
                             rand={"eik_idx": T(g["eik_idx"]), "eik_uniform": T(g["eik_uniform"]).to(dt)}, z_vals=T(g["z_vals"]).to(dt),
                             white_bkgd=True, bg_color=(1.0, 0.9, 0.8))
        lo = O.neat_loss(out, T(g["gt_rgb"]).to(dt), T(g["gt_lines2d"]).to(dt))
        lo["loss"].backward()
      finally:
        torch.set_default_dtype(torch.float32)
      if True:
        vals[dt] = (float(p["implicit_network.lin8.bias"].grad[0]), float(p["density.beta"].grad))
    ref = (float(g["grad_implicit_network.lin8.bias"][0]), float(g["grad_density.beta"].reshape(-1)[0]))
    for i, name in enumerate(("lin8.bias[0]", "density.beta")):
        assert abs(vals[torch.float32][i] - ref[i]) <= 2e-3 * abs(ref[i]), name          # fp32 oracle = reference
        rel = abs(vals[torch.float64][i] - ref[i]) / abs(ref[i])
        print(f"{name}: reference fp32 {ref[i]:.6e}, same algorithm in fp64 {vals[torch.float64][i]:.6e} ({rel:.2%})")
        assert 5e-3 < rel < 8e-2, (name, rel)


def test_train_step_hierarchical(golden):
    """G12: C5, hierarchical 64 coarse + 64 fine depths feeding the main pass."""
    g = golden("g12_train_step_hierarchical")
    p = params("rough", grad=True)
    lines, verts = _wf(g)
    rand = {k: T(g[k]) for k in ("t_rand", "eik_idx", "eik_uniform")}
    out = O.full_forward(p, _inp(g), lines, verts, training=True, rand=rand, sampler="hierarchical")
    close(out["z_vals"], g["z_vals"], 2e-5, "z_vals")
    lo = O.neat_loss(out, T(g["gt_rgb"]), T(g["gt_lines2d"]))
    _check_train_step(g, p, out, lo, ("rgb_values", "depth", "xyz", "l3d", "points3d", "lines3d", "lines2d_calib", "sdf", "grad_theta",
                                      "j3d_local", "j3d_global", "j2d_global_calib", "j2d_local_calib", "median"))


def test_sampler_is_ill_conditioned(golden):
    """Why the GPU sampler tests carry an allowance (tests/test_gpu_parity.py::close_sampler).  The inverse-CDF step of the
    reference (ray_sampler.py:237-249) replaces a bin's CDF span by 1 when it is below 1e-5; the final pdf is `weights + 1e-5`,
    normalised by ~1.0006, so EVERY empty bin has a span of 0.9999e-5 -- within float rounding of the threshold.  One ulp of noise on
    the SDF values the sampler reads (any implementation that is not bit-identical to torch-CPU has more) therefore moves some
    samples by a whole bin, while all others move by < 2e-4.  Shown here on the reference algorithm itself (the oracle is
    bit-identical to the golden), so the flips are a property of the algorithm, not of a summation order."""
    g = golden("g6_sampler_train_rough")
    p = params("rough")
    dirs, orig = O.camera_rays(T(g["uv"]), T(g["pose"]), T(g["intrinsics"]))
    dirs = dirs.reshape(-1, 3)
    o = orig[:, None, :].expand(1, dirs.shape[0], 3).reshape(-1, 3)
    rand = {k: T(g[k]) for k in ("t_rand", "u_final", "perm", "eik_idx")}
    z0, _ = O.error_bound_sampler(lambda x: O.sdf_values(p, x), O.beta_of(p), dirs, o, training=True, rand=rand)
    assert float((z0 - T(g["z_vals"])).abs().max()) == 0.0          # bit-identical to the reference
    flips, total = 0, 0
    for seed in range(4):
        gen = torch.Generator().manual_seed(seed)

        def noisy(x):
            s = O.sdf_values(p, x)
            return s * (1.0 + 1.2e-7 * torch.randn(s.shape, generator=gen).sign())      # +- one ulp
        z1, _ = O.error_bound_sampler(noisy, O.beta_of(p), dirs, o, training=True, rand=rand)
        err = (z1 - z0).abs()
        flips += int((err > 2e-4).sum())
        total += err.numel()
        assert float(err[err <= 2e-4].max()) < 2e-4 and float(err.max()) <= 2 * 6.0 / 127 + 1e-3
    assert flips >= 1                                               # at least one whole-bin move from one ulp of input noise
    assert flips / total <= 0.003


def test_real_scene_abc_00075213(golden):
    """G13: view 0 of the scene BASELINE configs 1 / 2 name (K / pose from the reference's cameras.npz, its HAWP wireframe through
    the reference's own WireframeGraph.load_json): eval forward with all keys, then a train step with all gradients."""
    g = golden("g13_real_scene_abc_00075213")
    lines, verts = _wf(g)
    # (HAWP scores this view's 13 edges 0.77 .. 1e-4: none passes the 0.97 default of line_segments() that the model's forward uses,
    #  rend_a :428 -- the junction block runs on an EMPTY ground-truth segment set here, the dataset's 0.05 threshold keeps 9)
    assert [int(v) for v in g["frame"]] == [512, 512] and tuple(verts.shape) == (8, 2) and lines.shape[0] == 0
    p = params("rough")
    out = O.full_forward(p, _inp(g), lines, verts, training=False, rand={"eik_idx": T(g["eval_eik_idx"])})
    for k in ("points", "rgb_values", "depth", "xyz", "l3d", "points3d", "lines3d", "lines2d_calib", "sdf", "normal_map"):
        close(out[k], g["eval_" + k], 5e-5, k)
    close(out["lines2d"], g["eval_lines2d"], 2e-2, "lines2d (pixels)")
    p = params("rough", grad=True)
    rand = {k: T(g[k]) for k in ("t_rand", "u_final", "perm", "eik_idx", "eik_uniform")}
    out = O.full_forward(p, _inp(g), lines, verts, training=True, rand=rand)
    lo = O.neat_loss(out, T(g["gt_rgb"]), T(g["gt_lines2d"]))
    _check_train_step(g, p, out, lo, ("rgb_values", "depth", "xyz", "l3d", "points3d", "lines3d", "lines2d_calib", "sdf", "grad_theta",
                                      "j3d_local", "j3d_global", "j2d_global_calib", "j2d_local_calib", "median"))
