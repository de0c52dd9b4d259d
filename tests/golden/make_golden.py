"""Generate golden vectors by running the REFERENCE implementation (authoring container only).

    python tests/golden/make_golden.py        # needs /root/reference ; writes tests/golden/*.npz

The reference is imported from /root/reference/code with three shims:
  * sys.modules stubs for open3d/trimesh/imageio/skimage/cv2 (imported, never called on the path);
  * a ConfigTree stand-in for pyhocon (the abc-neat-a model block is restated in neat_amd/synth.py);
  * Tensor.cuda / Module.cuda = identity (this container has no GPU).
The reference's random draws are recorded (torch.rand/randint/randperm/Tensor.uniform_)
so the oracle / HIP path can replay them.  Only DATA (inputs + outputs) is written;
no reference source travels to the GPU box.
"""
import importlib.util
import copy
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference/code"
GRAD_STRIDE = 13
sys.path.insert(0, REPO)

from neat_amd import synth  # noqa: E402


# ------------------------------------------------------------------ shims
class ConfigTree(dict):
    def _get(self, key, default):
        if key in self:
            return self[key]
        if default is _MISSING:
            raise KeyError(key)
        return default

    def get_int(self, k, default=None):
        return int(self._get(k, _MISSING if default is None else default))

    def get_float(self, k, default=None):
        return float(self._get(k, _MISSING if default is None else default))

    def get_bool(self, k, default=None):
        return bool(self._get(k, _MISSING if default is None else default))

    def get_list(self, k, default=None):
        return list(self._get(k, _MISSING if default is None else default))

    def get_config(self, k, default=None):
        v = self._get(k, _MISSING if default is None else default)
        return v if isinstance(v, ConfigTree) else to_tree(v)


_MISSING = object()


def to_tree(d):
    t = ConfigTree()
    for k, v in d.items():
        t[k] = to_tree(v) if isinstance(v, dict) else v
    return t


def install_shims():
    for name in ("open3d", "trimesh", "imageio", "skimage", "cv2"):
        sys.modules[name] = types.ModuleType(name)
    ph = types.ModuleType("pyhocon")
    ph.ConfigTree = ConfigTree
    sys.modules["pyhocon"] = ph
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.nn.Module.cuda = lambda self, *a, **k: self
    sys.path.insert(0, REF)


def load_wireframe_cls():
    # site-packages has HuggingFace `datasets`, which shadows the reference's namespace package
    spec = importlib.util.spec_from_file_location("ref_wireframe", os.path.join(REF, "datasets/utils/wireframe.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m.WireframeGraph


class RngTape:
    """Record torch.rand / randint / randperm / Tensor.uniform_ results in call order."""

    def __init__(self):
        self.tape = []
        self._orig = {}

    def __enter__(self):
        for name in ("rand", "randint", "randperm"):
            self._orig[name] = getattr(torch, name)
            setattr(torch, name, self._wrap(name, self._orig[name]))
        self._orig["uniform_"] = torch.Tensor.uniform_
        orig_u = self._orig["uniform_"]
        tape = self.tape

        def uniform_(t, *a, **k):
            r = orig_u(t, *a, **k)
            tape.append(("uniform_", r.clone()))
            return r
        torch.Tensor.uniform_ = uniform_
        return self

    def _wrap(self, name, fn):
        def f(*a, **k):
            r = fn(*a, **k)
            self.tape.append((name, r.clone()))
            return r
        return f

    def __exit__(self, *exc):
        for name in ("rand", "randint", "randperm"):
            setattr(torch, name, self._orig[name])
        torch.Tensor.uniform_ = self._orig["uniform_"]


def build_model(variant, seed=42, conf=None, num_junctions=64, sd_fn=None):
    from model.networks.neat_wfr_rend_a import VolSDFNetwork
    torch.manual_seed(0)
    net = VolSDFNetwork(to_tree(conf or synth.ABC_NEAT_A_MODEL_CONF))
    sd_np = synth.synth_state_dict(seed, variant, num_junctions=num_junctions)
    if sd_fn is not None:
        sd_np = sd_fn(sd_np)
    sd = {k: torch.tensor(v) for k, v in sd_np.items()}
    net.load_state_dict(sd, strict=True)
    return net


def scene_inputs(WG, seed, n_rays, view=0):
    sc = synth.synth_scene(seed=seed, n_rays=n_rays, view=view)
    wf = WG(torch.tensor(sc["wf_vertices"]), torch.tensor(sc["wf_vconf"]), torch.tensor(sc["wf_edges"]),
            torch.tensor(sc["wf_weights"]), 512, 512)
    inp = {"intrinsics": torch.tensor(sc["intrinsics"]), "pose": torch.tensor(sc["pose"]),
           "uv": torch.tensor(sc["uv"]), "uv_proj": torch.tensor(sc["uv_proj"]), "wireframe": [wf]}
    gt = {"rgb": torch.tensor(sc["gt_rgb"]), "lines2d": torch.tensor(sc["gt_lines2d"])}
    return sc, inp, gt


def np_(t):
    return t.detach().cpu().numpy()


def save(name, **arrs):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrs)
    print(f"  {name}.npz  {os.path.getsize(path) / 1024:.1f} KiB")


class DupRows:
    """While active, per-ray random draws (first dimension = rays) are made identical for rays r and r + R/2."""

    def __enter__(self):
        self._rand, self._randint, self._uni = torch.rand, torch.randint, torch.Tensor.uniform_

        def dup(t):
            if t.dim() >= 1 and t.shape[0] % 2 == 0 and t.shape[0] >= 2:
                h = t.shape[0] // 2
                t[h:] = t[:h]
            return t
        torch.rand = lambda *a, **k: dup(self._rand(*a, **k))
        torch.randint = lambda *a, **k: dup(self._randint(*a, **k))
        uni = self._uni
        torch.Tensor.uniform_ = lambda t, *a, **k: dup(uni(t, *a, **k))
        return self

    def __exit__(self, *exc):
        torch.rand, torch.randint, torch.Tensor.uniform_ = self._rand, self._randint, self._uni


def train_step_arrays(net, sc, inp, gt, tape_names, loss_conf=None):
    """One reference train step (forward + VolSDFLoss + backward) with its random draws recorded: inputs, draws, outputs, the loss
    scalars and a strided subsample + norm of every gradient."""
    from model.networks.loss_wfr import VolSDFLoss
    loss_fn = VolSDFLoss(**(loss_conf or synth.ABC_NEAT_A_LOSS_CONF))
    with RngTape() as tp:
        out = net(inp)
    names = [n for n, _ in tp.tape]
    assert names == tape_names, names
    lo = loss_fn(out, gt)
    lo["loss"].backward()
    arrs = {k: sc[k] for k in ("uv", "uv_proj", "pose", "intrinsics", "wf_vertices", "wf_vconf", "wf_edges",
                               "wf_weights", "gt_rgb", "gt_lines2d")}
    for k in ("rgb_values", "depth", "xyz", "l3d", "points3d", "lines3d", "lines2d_calib", "lines2d", "sdf",
              "grad_theta", "j3d_local", "j2d_local", "j2d_local_calib", "j3d_global", "j2d_global",
              "j2d_global_calib", "median"):
        if k in out:
            arrs["out_" + k] = np_(out[k])
    for k, v in lo.items():
        arrs["loss_" + k] = np_(torch.as_tensor(v).float())
    for k, prm in net.named_parameters():
        if prm.grad is None:
            continue
        gflat = np_(prm.grad).reshape(-1)
        arrs["grad_" + k] = gflat[::GRAD_STRIDE].copy()
        arrs["gradnorm_" + k] = np.array([np.sqrt((gflat.astype(np.float64) ** 2).sum()), gflat.astype(np.float64).sum()])
    return arrs, tp.tape


def extra_goldens():
    """G11 / G12 (round 2): the DTU / BlendedMVS model switches and the hierarchical 64 + 64 sampler as train steps of the reference."""
    install_shims()
    torch.set_default_dtype(torch.float32)
    from model import ray_sampler as ref_rs
    WG = load_wireframe_cls()

    # ---------------- G11: dtu.conf / bmvs.conf model switches (dbscan_enabled = True, use_median = False, 1024 latents;
    # rend_a :333-342,460,475-482), train step, rough weights.  32 distinct rays, each twice: DBSCAN(eps = 0.01, min_samples = 2)
    # then finds one cluster per line end point (a random batch has no two end points within 1 cm and the reference would stop
    # on an empty candidate set).
    conf = dict(synth.ABC_NEAT_A_MODEL_CONF)
    conf.update(dbscan_enabled=True, use_median=False)
    conf["global_junctions"] = dict(conf["global_junctions"], num_junctions=1024)
    net = build_model("rough", conf=conf, num_junctions=1024)
    net.train()
    sc, inp, gt = scene_inputs(WG, seed=21, n_rays=32, view=2)
    for k in ("uv", "uv_proj", "gt_rgb", "gt_lines2d"):
        sc[k] = np.concatenate([sc[k], sc[k]], axis=1)
    inp["uv"], inp["uv_proj"] = torch.tensor(sc["uv"]), torch.tensor(sc["uv_proj"])
    gt = {"rgb": torch.tensor(sc["gt_rgb"]), "lines2d": torch.tensor(sc["gt_lines2d"])}
    # the sampler's draws are per ray: the two copies of a ray get identical draws, so their samples and line end points coincide
    # exactly and the clustering does not hinge on distances near eps
    rec = {}
    inner = net.ray_sampler.get_z_vals

    def recording(*a, _inner=inner, _rec=rec, **k):
        _rec["z"], _rec["z_eik"] = _inner(*a, **k)
        return _rec["z"], _rec["z_eik"]
    net.ray_sampler.get_z_vals = recording
    with DupRows():
        arrs, tape = train_step_arrays(net, sc, inp, gt, ["rand", "randint", "rand", "randperm", "randint", "uniform_"])
    arrs.update(z_vals=np_(rec["z"]), z_eik=np_(rec["z_eik"]))
    arrs.update(t_rand=np_(tape[0][1]), u_final=np_(tape[2][1]), perm=np_(tape[3][1]), eik_idx=np_(tape[4][1]), eik_uniform=np_(tape[5][1]))
    save("g11_train_step_dtu_switches", **arrs)

    # ---------------- G12: C5 = hierarchical sampling 64 coarse + 64 fine feeding the main pass.  The reference ships the pieces
    # (UniformSampler.get_z_vals, get_z_vals_fine / sample_pdf, ray_sampler.py:16-106) but no model calls them; composed here exactly
    # as SURVEY 8(d) defines C5: coarse depths -> no-grad SDF -> volume_rendering weights -> fine depths -> sorted union, z_eik by randint.
    net = build_model("rough")
    net.train()
    sc, inp, gt = scene_inputs(WG, seed=23, n_rays=64, view=1)
    us = ref_rs.UniformSampler(net.scene_bounding_sphere, 0.0, 64, N_important=64)

    def hierarchical(ray_dirs, cam_loc, model):
        zc = us.get_z_vals(ray_dirs, cam_loc, model)
        pts = cam_loc.unsqueeze(1) + zc.unsqueeze(2) * ray_dirs.unsqueeze(1)
        with torch.no_grad():
            sdf = model.implicit_network.get_sdf_vals(pts.reshape(-1, 3))
            w = model.volume_rendering(zc, sdf)
        z = us.get_z_vals_fine(zc, w, model)
        idx = torch.randint(z.shape[-1], (z.shape[0],))
        rec["z"], rec["z_coarse"], rec["w_coarse"] = z, zc, w
        return z, torch.gather(z, 1, idx.unsqueeze(-1))
    rec = {}
    net.ray_sampler.get_z_vals = hierarchical
    arrs, tape = train_step_arrays(net, sc, inp, gt, ["rand", "randint", "randint", "uniform_"])
    arrs.update(t_rand=np_(tape[0][1]), eik_idx=np_(tape[2][1]), eik_uniform=np_(tape[3][1]), z_vals=np_(rec["z"]),
                z_coarse=np_(rec["z_coarse"]), w_coarse=np_(rec["w_coarse"]))
    save("g12_train_step_hierarchical", **arrs)



def switch_goldens():
    """G14 / G15 / G16 (round 5): the three model switches that neat_amd implements but no shipped conf sets --
    white_bkgd (rend_a :263-265,411-413), use_l3d (:461-465), junction_eikonal (:524-525) -- each as ONE train step of the reference
    (rough weights, 64 rays, its own random draws recorded), like G8.  G17 / G18: the architecture switches mode = 'nerf' of the two
    heads and inside_out of the SDF network."""
    install_shims()
    torch.set_default_dtype(torch.float32)
    WG = load_wireframe_cls()
    for tag, switch, seed, view in (("g14_train_step_white_bkgd", dict(white_bkgd=True, bg_color=[1.0, 0.9, 0.8]), 31, 1),
                                    ("g15_train_step_use_l3d", dict(use_l3d=True), 33, 2),
                                    ("g16_train_step_junction_eikonal", dict(junction_eikonal=True), 35, 3),
                                    # G17 / G18: architecture switches -- both heads with mode = 'nerf' (:180-181,240-241: input = [view, feature]),
                                    # and the SDF network with inside_out (:94-95)
                                    ("g17_train_step_nerf_heads", "nerf", 37, 0),
                                    ("g18_train_step_inside_out", "inside_out", 39, 1)):
        conf = copy.deepcopy(synth.ABC_NEAT_A_MODEL_CONF)
        sd_fn = None
        if switch == "nerf":
            conf["rendering_network"].update(mode="nerf", d_in=3)
            conf["attraction_network"].update(mode="nerf", d_in=3)
            sd_fn = synth.nerf_heads_state_dict
        elif switch == "inside_out":
            conf["implicit_network"]["inside_out"] = True
        else:
            conf.update(switch)
        net = build_model("rough", conf=conf, sd_fn=sd_fn)
        net.train()
        sc, inp, gt = scene_inputs(WG, seed=seed, n_rays=64, view=view)
        rec = {}
        inner = net.ray_sampler.get_z_vals

        def recording(*a, _inner=inner, _rec=rec, **k):
            _rec["z"], _rec["z_eik"] = _inner(*a, **k)
            return _rec["z"], _rec["z_eik"]
        net.ray_sampler.get_z_vals = recording
        arrs, tape = train_step_arrays(net, sc, inp, gt, ["rand", "randint", "rand", "randperm", "randint", "uniform_"])
        arrs.update(z_vals=np_(rec["z"]), z_eik=np_(rec["z_eik"]))
        arrs.update(t_rand=np_(tape[0][1]), u_final=np_(tape[2][1]), perm=np_(tape[3][1]), eik_idx=np_(tape[4][1]), eik_uniform=np_(tape[5][1]))
        save(tag, **arrs)


def real_scene_golden():
    """G13 (round 3): view 0 of the scene BASELINE configs 1 / 2 name, ABC 00075213, as bundled with the reference:
    K / pose from data/abc/00075213/cameras.npz, the HAWP wireframe from hawp/image_0000.json through the reference's own
    WireframeGraph.load_json (datasets/utils/wireframe.py:51-68), colours from images/image_0000.png.  The rays are 64 supporting
    pixels of the wireframe's attraction field (blender_hawp_dataset.py:93-198; hawp's `encodels` is absent, the field comes from
    oracle/attraction_oracle.py -- it only SELECTS inputs here: uv, uv_proj and the per-ray ground-truth segment are stored in the
    fixture as data).  One eval forward and one train step of the reference, rough weights."""
    install_shims()
    torch.set_default_dtype(torch.float32)
    from PIL import Image
    from oracle import attraction_oracle as AO
    WG = load_wireframe_cls()
    root = "/root/reference/data/abc/00075213"
    cams = np.load(os.path.join(root, "cameras.npz"))
    K = cams["intrinsics"][0].astype(np.float32)
    pose = cams["extrinsics"][0].astype(np.float32)
    wf = WG.load_json(os.path.join(root, "hawp", "image_0000.json"))
    lines = wf.line_segments(0.05)                       # [N, 5] (x1, y1, x2, y2, score), dataset ctor :69
    H, W = wf.frame_height, wf.frame_width
    lmap, label, _ = AO.encode_lines(np_(lines)[:, :4], H, W)
    mask, foot = AO.support(lmap, 10.0)
    img = np.asarray(Image.open(os.path.join(root, "images", "image_0000.png")).convert("RGB"), dtype=np.float32) / 255.0
    rng = np.random.RandomState(13)
    R = 64
    pix = rng.choice(np.flatnonzero(mask.reshape(-1)), R)          # with replacement, like __getitem__ (:190)
    ys, xs = pix // W, pix % W
    sc = {"uv": np.stack([xs, ys], -1).astype(np.float32)[None], "uv_proj": foot.reshape(-1, 2)[pix].astype(np.float32)[None],
          "pose": pose[None], "intrinsics": K[None],
          "wf_vertices": np_(wf.vertices), "wf_vconf": np_(wf.v_confidences), "wf_edges": np_(wf.edges), "wf_weights": np_(wf.weights),
          "gt_rgb": img.reshape(-1, 3)[pix][None], "gt_lines2d": np_(lines)[label.reshape(-1)[pix]][None]}
    inp = {"intrinsics": torch.tensor(sc["intrinsics"]), "pose": torch.tensor(sc["pose"]), "uv": torch.tensor(sc["uv"]),
           "uv_proj": torch.tensor(sc["uv_proj"]), "wireframe": [wf]}
    gt = {"rgb": torch.tensor(sc["gt_rgb"]), "lines2d": torch.tensor(sc["gt_lines2d"])}
    # eval forward (all keys) with the sampler's depths recorded
    net = build_model("rough")
    net.eval()
    rec = {}
    inner = net.ray_sampler.get_z_vals

    def recording(*a, _inner=inner, _rec=rec, **k):
        _rec["z"], _rec["z_eik"] = _inner(*a, **k)
        return _rec["z"], _rec["z_eik"]
    net.ray_sampler.get_z_vals = recording
    with RngTape() as tp:
        out = net(inp)
    eik_idx = [t for n, t in tp.tape if n == "randint"][-1]
    keys = ["points", "rgb_values", "depth", "xyz", "l3d", "points3d", "lines3d", "lines2d_calib", "lines2d", "sdf", "normal_map"]
    arrs = {"eval_" + k: np_(out[k]) for k in keys}
    arrs.update(eval_z_vals=np_(rec["z"]), eval_eik_idx=np_(eik_idx), frame=np.array([H, W]))
    # train step on the same rays
    net = build_model("rough")
    net.train()
    net.ray_sampler.get_z_vals = (lambda *a, _inner=net.ray_sampler.get_z_vals, _rec=rec, **k: _rec.__setitem__("zt", _inner(*a, **k)) or _rec["zt"])
    tarrs, tape = train_step_arrays(net, sc, inp, gt, ["rand", "randint", "rand", "randperm", "randint", "uniform_"])
    arrs.update(tarrs)
    arrs.update(z_vals=np_(rec["zt"][0]), z_eik=np_(rec["zt"][1]))
    arrs.update(t_rand=np_(tape[0][1]), u_final=np_(tape[2][1]), perm=np_(tape[3][1]), eik_idx=np_(tape[4][1]), eik_uniform=np_(tape[5][1]))
    save("g13_real_scene_abc_00075213", **arrs)


def main():
    install_shims()
    torch.set_default_dtype(torch.float32)
    from model.embedder import get_embedder
    from model.density import LaplaceDensity
    from model import ray_sampler as ref_rs
    from model.networks.loss_wfr import VolSDFLoss
    from utils import rend_util
    WG = load_wireframe_cls()
    g = np.random.Generator(np.random.PCG64(123))

    # ---------------- G1 positional encoding
    x = torch.tensor(g.uniform(-3, 3, size=(64, 3)), dtype=torch.float32)
    e6, _ = get_embedder(6, 3)
    e4, _ = get_embedder(4, 3)
    save("g1_posenc", x=np_(x), pe6=np_(e6(x)), pe4=np_(e4(x)))

    for variant in ("init", "rough"):
        net = build_model(variant)
        net.eval()
        imp = net.implicit_network
        # ------------- G2 implicit network on points spanning |x| in [0,3.5]
        d = g.standard_normal((256, 3))
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        r = np.linspace(0.0, 3.5, 256)[:, None]
        x = torch.tensor(d * r, dtype=torch.float32)
        fwd = imp.forward(x.clone())
        sdfv = imp.get_sdf_vals(x.clone())
        s, f, gr = imp.get_outputs(x.clone())
        graw = imp.gradient(x.clone())
        # G3 heads on the same points
        vd = g.standard_normal((256, 3))
        vd /= np.linalg.norm(vd, axis=1, keepdims=True)
        vd = torch.tensor(vd, dtype=torch.float32)
        rgb = net.rendering_network(x, gr, vd, f)
        lines = net.attraction_network(x, gr, vd, f)
        save(f"g2g3_networks_{variant}", x=np_(x), view=np_(vd), forward=np_(fwd), sdf_vals=np_(sdfv),
             out_sdf=np_(s), out_feat=np_(f), out_grad=np_(gr), grad_raw=np_(graw), rgb=np_(rgb), lines=np_(lines))

        # ------------- G6 sampler, eval mode, 64 rays (records per-round beta via instrumentation)
        sc, inp, gt = scene_inputs(WG, seed=5, n_rays=64)
        dirs, cam = rend_util.get_camera_params(inp["uv"], inp["pose"], inp["intrinsics"])
        dirs = dirs.reshape(-1, 3)
        camr = cam.unsqueeze(1).repeat(1, 64, 1).reshape(-1, 3)
        with RngTape() as tp:
            z, zeik = net.ray_sampler.get_z_vals(dirs, camr, net)
        eik_idx = [t for n, t in tp.tape if n == "randint"][-1]
        save(f"g6_sampler_eval_{variant}", uv=sc["uv"], pose=sc["pose"], intrinsics=sc["intrinsics"],
             dirs=np_(dirs), z_vals=np_(z), z_eik=np_(zeik), eik_idx=np_(eik_idx))
        # training-mode sampler with recorded randoms
        net.train()
        with RngTape() as tp:
            z, zeik = net.ray_sampler.get_z_vals(dirs, camr, net)
        names = [n for n, _ in tp.tape]
        assert names == ["rand", "randint", "rand", "randperm", "randint"], names
        save(f"g6_sampler_train_{variant}", uv=sc["uv"], pose=sc["pose"], intrinsics=sc["intrinsics"],
             z_vals=np_(z), z_eik=np_(zeik), t_rand=np_(tp.tape[0][1]), u_final=np_(tp.tape[2][1]),
             perm=np_(tp.tape[3][1]), eik_idx=np_(tp.tape[4][1]))
        net.eval()

        # ------------- G7 full forward, eval, R=64
        rec = {}
        inner = net.ray_sampler.get_z_vals

        def recording(*a, _inner=inner, _rec=rec, **k):
            _rec["z"], _rec["z_eik"] = _inner(*a, **k)
            return _rec["z"], _rec["z_eik"]
        net.ray_sampler.get_z_vals = recording
        with RngTape() as tp:
            out = net(inp)
        net.ray_sampler.get_z_vals = inner
        eik_idx = [t for n, t in tp.tape if n == "randint"][-1]
        keys = ["points", "rgb_values", "depth", "xyz", "l3d", "points3d", "lines3d", "lines2d_calib", "lines2d",
                "sdf", "normal_map"]
        save(f"g7_forward_eval_{variant}", eik_idx=np_(eik_idx), z_vals=np_(rec["z"]), **{k: sc[k] for k in
             ("uv", "uv_proj", "pose", "intrinsics", "wf_vertices", "wf_vconf", "wf_edges", "wf_weights")},
             **{"out_" + k: np_(out[k]) for k in keys})

    # ---------------- G8 full train step (rough weights), R=64: outputs, losses, all 65 grads
    net = build_model("rough")
    net.train()
    sc, inp, gt = scene_inputs(WG, seed=9, n_rays=64, view=3)
    loss_fn = VolSDFLoss(**synth.ABC_NEAT_A_LOSS_CONF)
    with RngTape() as tp:
        out = net(inp)
    names = [n for n, _ in tp.tape]
    assert names == ["rand", "randint", "rand", "randperm", "randint", "uniform_"], names
    lo = loss_fn(out, gt)
    lo["loss"].backward()
    arrs = {k: sc[k] for k in ("uv", "uv_proj", "pose", "intrinsics", "wf_vertices", "wf_vconf", "wf_edges",
                               "wf_weights", "gt_rgb", "gt_lines2d")}
    arrs.update(t_rand=np_(tp.tape[0][1]), u_final=np_(tp.tape[2][1]), perm=np_(tp.tape[3][1]),
                eik_idx=np_(tp.tape[4][1]), eik_uniform=np_(tp.tape[5][1]))
    for k in ("rgb_values", "depth", "xyz", "l3d", "points3d", "lines3d", "lines2d_calib", "lines2d", "sdf",
              "grad_theta", "j3d_local", "j2d_local", "j2d_local_calib", "j3d_global", "j2d_global",
              "j2d_global_calib", "median"):
        arrs["out_" + k] = np_(out[k])
    for k, v in lo.items():
        arrs["loss_" + k] = np_(torch.as_tensor(v).float())
    # grads of all 65 parameter tensors: strided subsample (GRAD_STRIDE) + L2 norm + sum, to keep the fixture small
    for k, prm in net.named_parameters():
        gflat = np_(prm.grad).reshape(-1)
        arrs["grad_" + k] = gflat[::GRAD_STRIDE].copy()
        arrs["gradnorm_" + k] = np.array([np.sqrt((gflat.astype(np.float64) ** 2).sum()), gflat.astype(np.float64).sum()])
    save("g8_train_step_rough", **arrs)

    # ---------------- G4 density, G5 volume rendering
    dens = LaplaceDensity(params_init={"beta": 0.1})
    s = torch.linspace(-1, 1, 201)
    arrs = {"sdf": np_(s)}
    for b in (1e-3, 1e-2, 0.1):
        arrs[f"sigma_{b:g}"] = np_(dens(s, beta=torch.tensor(b)))
    save("g4_density", **arrs)
    net = build_model("rough")
    net.eval()
    z = torch.tensor(np.sort(g.uniform(0, 6, size=(32, 98)), axis=1), dtype=torch.float32)
    sdf = torch.tensor(g.uniform(-0.3, 0.5, size=(32 * 98, 1)), dtype=torch.float32)
    save("g5_volume_rendering", z=np_(z), sdf=np_(sdf), weights=np_(net.volume_rendering(z, sdf)),
         beta=np_(net.density.get_beta()))

    # ---------------- G9 hierarchical sampling (a13)
    us = ref_rs.UniformSampler(3.0, 0.0, 64, N_important=64)

    class M:
        training = False
    zc = us.get_z_vals(torch.zeros(16, 3), torch.zeros(16, 3), M)
    w = torch.tensor(g.uniform(0, 1, size=(16, 64)) ** 4, dtype=torch.float32)
    M.training = True      # det=model.training (inverted vs NeRF): linspace u
    zf_det = us.get_z_vals_fine(zc, w, M)
    M.training = False
    with RngTape() as tp:
        zf_rnd = us.get_z_vals_fine(zc, w, M)
    save("g9_hierarchical", z_coarse=np_(zc), weights=np_(w), z_fine_det=np_(zf_det), z_fine_rand=np_(zf_rnd),
         u_rand=np_(tp.tape[0][1]))

    # ---------------- G10 camera rays incl. a skewed K
    sc = synth.synth_scene(seed=11, n_rays=32)
    Ks = sc["intrinsics"].copy()
    Ks[0, 0, 1] = 3.5
    Ks[0, 1, 1] = 540.0
    arrs = {"uv": sc["uv"], "pose": sc["pose"], "K": sc["intrinsics"], "K_skew": Ks}
    for tag, K in (("", sc["intrinsics"]), ("_skew", Ks)):
        dd, cc = rend_util.get_camera_params(torch.tensor(sc["uv"]), torch.tensor(sc["pose"]), torch.tensor(K))
        arrs["dirs" + tag], arrs["cam" + tag] = np_(dd), np_(cc)
    save("g10_camera", **arrs)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "extra":      # only the round-2 fixtures (G11, G12); the others are left untouched
        extra_goldens()
    elif len(sys.argv) > 1 and sys.argv[1] == "switches":   # only the round-5 fixtures (G14-G16)
        switch_goldens()
    elif len(sys.argv) > 1 and sys.argv[1] == "real":     # only the round-3 fixture (G13)
        real_scene_golden()
    else:
        main()
        extra_goldens()
        real_scene_golden()
        switch_goldens()
