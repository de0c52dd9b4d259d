"""Data fixture for the convergence check (VERDICT r3 "next" #6): a few down-sampled views of the scene the reference ships,
/root/reference/data/abc/00075213 (the scene of BASELINE configs 1 / 2) -- images, cameras and the HAWP wireframes as ARRAYS.

    python tests/golden/make_scene_fixture.py        (in the container that holds /root/reference)

writes tests/golden/scene_abc_00075213_8views.npz:
  images      uint8 [V, R, R, 3]   every 12th view, 512 x 512 -> R x R by area averaging (PIL BOX filter)
  intrinsics  float64 [V, 3, 3]    cameras.npz intrinsics with fx, fy, cx, cy scaled by R / 512
  extrinsics  float64 [V, 4, 4]    cameras.npz extrinsics, unchanged
  view_ids    int64 [V]            indices of the views in the reference's directory
  wf_<v>_vertices / _scores / _edges / _weights   the wireframe of view v (hawp/image_%04d.json), vertices scaled by R / 512
Nothing of the reference's source text is stored: the fixture is inputs only (no expected outputs -- the check compares the
builds of THIS repo with one another while they learn the scene).  `neat_amd.synth.write_scene_fixture` turns the arrays back
into the directory layout the dataset class reads (images/*.png, cameras.npz, hawp/*.json)."""
import json
import os

import numpy as np
from PIL import Image

ROOT = "/root/reference/data/abc/00075213"
RES, STEP = 128, 12


def main():
    cams = np.load(os.path.join(ROOT, "cameras.npz"))
    ids = list(range(0, 100, STEP))[:8]
    out = {"view_ids": np.array(ids, dtype=np.int64)}
    imgs, Ks, Es = [], [], []
    for v in ids:
        im = Image.open(os.path.join(ROOT, "images", f"image_{v:04d}.png")).convert("RGB")
        s = RES / im.size[0]
        imgs.append(np.asarray(im.resize((RES, RES), Image.BOX), dtype=np.uint8))
        K = cams["intrinsics"][v].astype(np.float64).copy()
        K[:2, :] *= s
        Ks.append(K[:3, :3])
        Es.append(cams["extrinsics"][v].astype(np.float64))
        wf = json.load(open(os.path.join(ROOT, "hawp", f"image_{v:04d}.json")))
        assert wf["height"] == im.size[1] and wf["width"] == im.size[0]
        out[f"wf_{v}_vertices"] = np.asarray(wf["vertices"], dtype=np.float64).reshape(-1, 2) * s
        out[f"wf_{v}_scores"] = np.asarray(wf["vertices-score"], dtype=np.float64)
        out[f"wf_{v}_edges"] = np.asarray(wf["edges"], dtype=np.int64).reshape(-1, 2)
        out[f"wf_{v}_weights"] = np.asarray(wf["edges-weights"], dtype=np.float64)
    out.update(images=np.stack(imgs), intrinsics=np.stack(Ks), extrinsics=np.stack(Es))
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "scene_abc_00075213_8views.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path), "bytes;", {k: v.shape for k, v in out.items() if not k.startswith("wf_")})


if __name__ == "__main__":
    main()
