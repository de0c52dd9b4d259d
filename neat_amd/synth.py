"""Seeded synthetic weights and scene inputs for tests, goldens and bench.py.

Nothing here is read from the reference at run time.  The recipes restate, in
NumPy, the *distributions* the reference initialises its networks with
(geometric init: code/model/networks/neat_wfr_rend_a.py:55-69; torch's default
nn.Linear init for the two heads and the junction ffn), so that the same
weights can be re-created on the GPU box without committing 4.9 MB of floats.
The model block is confs/abc-neat-a.conf:29-88 restated as a dict.
"""
import math

import numpy as np

# code/confs/abc-neat-a.conf:29-88
ABC_NEAT_A_MODEL_CONF = {
    "feature_vector_size": 256,
    "scene_bounding_sphere": 3.0,
    "dbscan_enabled": False,
    "use_l3d": False,
    "use_median": True,
    "global_junctions": {"num_junctions": 64, "num_layers": 2, "dim_out": 3, "dim_hidden": 256},
    "implicit_network": {
        "d_in": 3, "d_out": 1, "dims": [256] * 8, "geometric_init": True, "bias": 0.6,
        "skip_in": [4], "weight_norm": True, "multires": 6, "sphere_scale": 20.0,
    },
    "attraction_network": {"d_in": 9, "d_out": 6, "dims": [256] * 4, "mode": "idr", "weight_norm": True},
    "rendering_network": {"mode": "idr", "d_in": 9, "d_out": 3, "dims": [256] * 4,
                          "weight_norm": True, "multires_view": 4},
    "density": {"params_init": {"beta": 0.1}, "beta_min": 0.0001},
    "ray_sampler": {"near": 0.0, "N_samples": 64, "N_samples_eval": 128, "N_samples_extra": 32,
                    "eps": 0.1, "beta_iters": 10, "max_total_iters": 5},
}
# code/confs/abc-neat-a.conf:15-21
ABC_NEAT_A_LOSS_CONF = {"eikonal_weight": 0.1, "line_weight": 0.01, "rgb_loss": "torch.nn.L1Loss"}

SDF_DIMS = [39, 256, 256, 256, 256, 256, 256, 256, 256, 257]   # dims after PE-6 / +feature
SDF_SKIP = 4
RENDER_IN = 3 + 27 + 3 + 256     # p, PE4(view), normal, feature
ATTR_IN = 3 + 3 + 3 + 256        # p, view, normal, feature


def _linear_default(rng, out_dim, in_dim):
    bound = 1.0 / math.sqrt(in_dim)
    w = rng.uniform(-bound, bound, size=(out_dim, in_dim))
    b = rng.uniform(-bound, bound, size=(out_dim,))
    return w, b


def synth_state_dict(seed=42, variant="init", num_junctions=64):
    """Return {state_dict key: float32 ndarray} with the reference's key names/shapes.

    variant "init": geometric init (a sphere of radius `bias`).
    variant "rough": geometric init plus seeded perturbations of the SDF-MLP
    weights/biases/gains so the level set is bumpy (the sampler then needs
    several refinement rounds and the normals are non-trivial).
    """
    rng = np.random.Generator(np.random.PCG64(seed))
    sd = {}
    sd["latents"] = rng.standard_normal((num_junctions, 256))
    # --- implicit network -------------------------------------------------
    n_lin = len(SDF_DIMS) - 1
    for l in range(n_lin):
        in_dim = SDF_DIMS[l]
        out_dim = SDF_DIMS[l + 1] - SDF_DIMS[0] if (l + 1) == SDF_SKIP else SDF_DIMS[l + 1]
        if l == n_lin - 1:
            w = rng.normal(math.sqrt(math.pi) / math.sqrt(in_dim), 1e-4, size=(out_dim, in_dim))
            b = np.full((out_dim,), -0.6)
        elif l == 0:
            w = np.zeros((out_dim, in_dim))
            w[:, :3] = rng.normal(0.0, math.sqrt(2) / math.sqrt(out_dim), size=(out_dim, 3))
            b = np.zeros((out_dim,))
        elif l == SDF_SKIP:
            w = rng.normal(0.0, math.sqrt(2) / math.sqrt(out_dim), size=(out_dim, in_dim))
            w[:, -(SDF_DIMS[0] - 3):] = 0.0
            b = np.zeros((out_dim,))
        else:
            w = rng.normal(0.0, math.sqrt(2) / math.sqrt(out_dim), size=(out_dim, in_dim))
            b = np.zeros((out_dim,))
        g = np.linalg.norm(w, axis=1, keepdims=True)
        if variant == "rough":
            w = w + rng.normal(0.0, 0.35 / math.sqrt(in_dim), size=w.shape)
            if l == 0:
                w[:, 3:] = rng.normal(0.0, 0.08 / math.sqrt(out_dim), size=(out_dim, in_dim - 3)) * (
                    1.0 / (1.0 + np.arange(in_dim - 3) // 6))
            b = b + rng.normal(0.0, 0.02, size=b.shape)
            g = g * rng.uniform(0.85, 1.15, size=g.shape)
        sd[f"implicit_network.lin{l}.bias"] = b
        sd[f"implicit_network.lin{l}.weight_g"] = g
        sd[f"implicit_network.lin{l}.weight_v"] = w
    # --- heads (torch default Linear init, then weight_norm => g = |v|) -----
    for name, d_in0, d_out in (("rendering_network", RENDER_IN, 3), ("attraction_network", ATTR_IN, 6)):
        dims = [d_in0, 256, 256, 256, 256, d_out]
        for l in range(5):
            w, b = _linear_default(rng, dims[l + 1], dims[l])
            g = np.linalg.norm(w, axis=1, keepdims=True)
            if variant == "rough":
                g = g * rng.uniform(0.8, 1.25, size=g.shape)
            sd[f"{name}.lin{l}.bias"] = b
            sd[f"{name}.lin{l}.weight_g"] = g
            sd[f"{name}.lin{l}.weight_v"] = w
    sd["density.beta"] = np.array(0.1 if variant == "init" else 0.03)
    for i, (o, k) in zip((0, 2, 4), ((256, 256), (256, 256), (3, 256))):
        w, b = _linear_default(rng, o, k)
        sd[f"ffn.{i}.weight"] = w
        sd[f"ffn.{i}.bias"] = b
    return {k: np.ascontiguousarray(v, dtype=np.float32) for k, v in sd.items()}


def nerf_heads_state_dict(sd):
    """The same weights for heads built with mode = 'nerf' (rend_a :180-181,240-241: the input is [view, feature] without the point and
    the normal): the point / normal columns of the two input layers are dropped ([256, 289] -> [256, 283], [256, 265] -> [256, 259])."""
    out = dict(sd)
    v = sd["rendering_network.lin0.weight_v"]
    out["rendering_network.lin0.weight_v"] = np.ascontiguousarray(np.concatenate([v[:, 3:30], v[:, 33:]], axis=1))
    v = sd["attraction_network.lin0.weight_v"]
    out["attraction_network.lin0.weight_v"] = np.ascontiguousarray(np.concatenate([v[:, 3:6], v[:, 9:]], axis=1))
    return out


def look_at_pose(cam_pos, target=(0.0, 0.0, 0.0), up=(0.0, 0.0, 1.0)):
    """Camera-to-world 4x4 (OpenCV convention: +z forward, +y down), float32."""
    c = np.asarray(cam_pos, dtype=np.float64)
    fwd = np.asarray(target, dtype=np.float64) - c
    fwd /= np.linalg.norm(fwd)
    right = np.cross(fwd, np.asarray(up, dtype=np.float64))
    right /= np.linalg.norm(right)
    down = np.cross(fwd, right)
    pose = np.eye(4)
    pose[:3, 0], pose[:3, 1], pose[:3, 2], pose[:3, 3] = right, down, fwd, c
    return pose.astype(np.float32)


def synth_scene(seed=42, n_rays=1024, res=512, radius=2.0, view=0):
    """Synthetic per-step model input in the layout the reference trainer hands the model
    (code/training/volsdf_train.py:361-366; dataset: code/datasets/blender_hawp_dataset.py:186-216).

    K=[[560,0,256],[0,560,256],[0,0,1]], camera on the radius-2 sphere looking at the origin
    (as data/abc/00075213/cameras.npz), uv ~ U[0,res)^2, uv_proj = uv + N(0,1),
    a cube-like wireframe of 8 vertices / 12 edges, gt rgb ~ U[0,1), gt 2-D segments by random label.
    """
    rng = np.random.Generator(np.random.PCG64(seed + 1000 * (view + 1)))
    ang = 2.0 * math.pi * (view * 0.61803398875 % 1.0)
    elev = 0.6
    cam = (radius * math.cos(elev) * math.cos(ang), radius * math.cos(elev) * math.sin(ang), radius * math.sin(elev))
    pose = look_at_pose(cam)
    K = np.eye(4, dtype=np.float32)
    K[0, 0] = K[1, 1] = 560.0
    K[0, 2] = K[1, 2] = res / 2.0
    uv = rng.uniform(0.0, res, size=(n_rays, 2)).astype(np.float32)
    uv_proj = (uv + rng.standard_normal((n_rays, 2))).astype(np.float32)
    # wireframe: project a cube of half-size 0.5
    corners = np.array([[x, y, z] for x in (-.5, .5) for y in (-.5, .5) for z in (-.5, .5)], dtype=np.float64)
    w2c = np.linalg.inv(pose.astype(np.float64))
    pc = (w2c[:3, :3] @ corners.T + w2c[:3, 3:]).T
    verts = (K[:3, :3].astype(np.float64) @ pc.T).T
    verts = (verts[:, :2] / verts[:, 2:]).astype(np.float32)
    edges = np.array([[0, 1], [0, 2], [0, 4], [1, 3], [1, 5], [2, 3], [2, 6], [3, 7], [4, 5], [4, 6], [5, 7], [6, 7]],
                     dtype=np.int64)
    edge_w = rng.uniform(0.975, 1.0, size=(len(edges),)).astype(np.float32)
    v_conf = rng.uniform(0.5, 1.0, size=(8,)).astype(np.float32)
    labels = rng.integers(0, len(edges), size=(n_rays,))
    seg = np.concatenate([verts[edges[labels, 0]], verts[edges[labels, 1]], edge_w[labels, None]], axis=1)
    gt_rgb = rng.uniform(0.0, 1.0, size=(n_rays, 3)).astype(np.float32)
    return {
        "intrinsics": K[None], "pose": pose[None], "uv": uv[None], "uv_proj": uv_proj[None],
        "wf_vertices": verts, "wf_vconf": v_conf, "wf_edges": edges, "wf_weights": edge_w,
        "gt_rgb": gt_rgb[None], "gt_lines2d": seg.astype(np.float32)[None], "res": res,
    }


def synth_z_vals(seed, n_rays, n_samples, near=0.0, far=6.0):
    """Headline C2 depth samples: sorted stratified U[near,far) per ray."""
    rng = np.random.Generator(np.random.PCG64(seed + 7))
    edges = np.linspace(near, far, n_samples + 1)
    u = rng.uniform(0.0, 1.0, size=(n_rays, n_samples))
    return (edges[:-1] + (edges[1:] - edges[:-1]) * u).astype(np.float32)


def write_scene_fixture(npz_path, root):
    """tests/golden/scene_abc_00075213_8views.npz (made by tests/golden/make_scene_fixture.py: down-sampled views of the ABC scene the
    reference ships, as arrays) -> the directory layout the dataset class reads: images/image_%04d.png, cameras.npz, hawp/image_%04d.json.
    Returns the image resolution."""
    import json
    import os
    from PIL import Image
    d = np.load(npz_path)
    os.makedirs(os.path.join(root, "images"), exist_ok=True)
    os.makedirs(os.path.join(root, "hawp"), exist_ok=True)
    res = int(d["images"].shape[1])
    for i, v in enumerate(d["view_ids"].tolist()):
        Image.fromarray(d["images"][i]).save(os.path.join(root, "images", f"image_{i:04d}.png"))
        json.dump({"vertices": d[f"wf_{v}_vertices"].tolist(), "vertices-score": d[f"wf_{v}_scores"].tolist(),
                   "edges": d[f"wf_{v}_edges"].tolist(), "edges-weights": d[f"wf_{v}_weights"].tolist(), "height": res, "width": res},
                  open(os.path.join(root, "hawp", f"image_{i:04d}.json"), "w"))
    np.savez(os.path.join(root, "cameras.npz"), intrinsics=d["intrinsics"], extrinsics=d["extrinsics"])
    return res


def hocon_text(d, indent=0):
    """A nested dict as HOCON text (what neat_amd.conf reads back): conf files for tests and scripts that drive the runner."""
    pad = "    " * indent
    out = []
    for k, v in d.items():
        if isinstance(v, dict):
            out.append(f"{pad}{k}{{\n{hocon_text(v, indent + 1)}{pad}}}")
        elif isinstance(v, bool):
            out.append(f"{pad}{k} = {'True' if v else 'False'}")
        elif isinstance(v, (list, tuple)):
            out.append(f"{pad}{k} = [{', '.join(str(x) for x in v)}]")
        else:
            out.append(f"{pad}{k} = {v}")
    return "\n".join(out) + "\n"
