"""dataset attraction field (HIP kernel through the C ABI) vs the brute-force CPU oracle, and the dataset
class end to end on a synthetic scene directory.  GPU tests; the oracle's own sanity checks run on CPU."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import attraction_oracle as A


def test_oracle_basics():
    lines = np.array([[2.0, 2.0, 10.0, 2.0], [4.0, 8.0, 4.0, 14.0]], np.float32)
    lmap, label, d2 = A.encode_lines(lines, 16, 16)
    assert label[2, 5] == 0 and abs(d2[2, 5]) < 1e-12           # on segment 0
    assert label[11, 4] == 1 and np.allclose(lmap[:2, 11, 6], [-2.0, 0.0])
    assert np.allclose(lmap[2:4, 0, 0], [2.0, 2.0]) and np.allclose(lmap[4:6, 0, 0], [10.0, 2.0])
    mask, foot = A.support(lmap, 3.0)
    # after the reference's clamps (:125-128) the two angle tests are always true: support == within `distance`
    assert mask[4, 6] and mask[2, 12] and not mask[2, 15] and (mask == (np.sqrt(d2) <= 3.0)).all()
    assert np.allclose(foot[4, 6], [6.0, 2.0])


@pytest.mark.gpu
@pytest.mark.parametrize("H,W,N,seed", [(64, 96, 13, 0), (120, 160, 300, 1), (33, 17, 1, 2)])
def test_encode_lines_vs_oracle(H, W, N, seed):
    from neat_amd import datasets
    rng = np.random.default_rng(seed)
    lines = np.concatenate([rng.uniform(0, W, (N, 1)), rng.uniform(0, H, (N, 1)), rng.uniform(0, W, (N, 1)),
                            rng.uniform(0, H, (N, 1))], 1).astype(np.float32)
    if N > 2:
        lines[2, 2:] = lines[2, :2]                              # a degenerate (zero-length) segment
    lmap, label = datasets.encode_lines(torch.tensor(lines).cuda(), H, W)
    ref_lmap, ref_label, ref_d2 = A.encode_lines(lines, H, W)
    d2 = (lmap[0] ** 2 + lmap[1] ** 2).cpu().numpy()
    assert np.abs(d2 - ref_d2).max() <= 1e-3                      # same nearest distance everywhere
    agree = label.cpu().numpy() == ref_label
    assert agree.mean() > 0.999                                   # labels may differ only on exact ties
    assert np.abs(lmap.cpu().numpy() - ref_lmap)[:, agree].max() <= 1e-3
    mask, lab, foot = datasets.compute_point_line_attraction(torch.tensor(lines), (H, W), distance=10.0)
    ref_mask, ref_foot = A.support(ref_lmap, 10.0)
    m = mask.numpy().reshape(H, W)
    assert (m != ref_mask).mean() < 2e-3                          # boundary pixels of the support may flip by rounding
    both = m & ref_mask & agree
    assert np.abs(foot.cpu().numpy().reshape(H, W, 2) - ref_foot)[both].max() <= 1e-3


def test_oracle_validity_map():
    """zero segments: no pixel has a nearest segment; a zero-length segment is a segment; a non-finite segment is none."""
    lmap, label, best = A.encode_lines(np.zeros((0, 4), np.float32), 8, 8)
    assert not A.valid_map(best).any() and (lmap == 0).all() and (label == 0).all()
    assert not A.support(lmap, 10.0, A.valid_map(best))[0].any()
    lmap, label, best = A.encode_lines(np.array([[3.0, 3.0, 3.0, 3.0]], np.float32), 8, 8)
    assert A.valid_map(best).all() and np.allclose(lmap[:2, 3, 3], 0.0) and np.allclose(lmap[:2, 0, 0], [3.0, 3.0])
    with np.errstate(invalid="ignore"):
        lmap, label, best = A.encode_lines(np.array([[np.nan, 1.0, 2.0, 2.0]], np.float32), 8, 8)
    assert not A.valid_map(best).any()


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["empty", "point", "nan", "mixed"])
def test_encode_lines_validity_vs_oracle(case):
    """The `labels_onehot.max(dim=0)[0]` output of the encodels replacement on the degenerate inputs, against the
    oracle, and multiplied into the support mask as blender_hawp_dataset.py:98,130 do."""
    from neat_amd import datasets
    H, W = 24, 40
    lines = {"empty": np.zeros((0, 4), np.float32), "point": np.array([[7.0, 5.0, 7.0, 5.0]], np.float32),
             "nan": np.array([[np.nan, 1.0, 2.0, 2.0]], np.float32),
             "mixed": np.array([[np.nan, 1.0, 2.0, 2.0], [3.0, 4.0, 30.0, 20.0], [9.0, 9.0, 9.0, 9.0]], np.float32)}[case]
    lmap, label, valid = datasets.encode_lines(torch.tensor(lines).reshape(-1, 4).cuda(), H, W, return_valid=True)
    with np.errstate(invalid="ignore"):
        ref_lmap, ref_label, ref_best = A.encode_lines(lines, H, W)
    ref_valid = A.valid_map(ref_best)
    assert (valid.cpu().numpy() == ref_valid).all()
    assert valid.any().item() == (case in ("point", "mixed"))
    ok = ref_valid & (label.cpu().numpy() == ref_label)
    assert ok.sum() == ref_valid.sum()
    if ok.any():
        assert np.abs(lmap.cpu().numpy() - ref_lmap)[:, ok].max() <= 1e-4
    else:
        assert (lmap == 0).all() and (label == 0).all()
    lines5 = np.concatenate([lines.reshape(-1, 4), np.ones((lines.reshape(-1, 4).shape[0], 1), np.float32)], 1)
    mask, lab, foot = datasets.compute_point_line_attraction(torch.tensor(lines5), (H, W), distance=10.0)
    with np.errstate(invalid="ignore"):
        ref_mask, _ = A.support(ref_lmap, 10.0, ref_valid)
    assert (mask.numpy().reshape(H, W) == ref_mask).mean() > 0.995 and (mask.numpy().reshape(H, W) & ~ref_valid).sum() == 0


@pytest.mark.gpu
def test_blender_dataset_end_to_end(tmp_path):
    """Synthetic scene directory in the reference's layout -> dataset -> one model/loss step."""
    from PIL import Image
    from neat_amd import datasets, networks, synth
    from neat_amd.loss import VolSDFLoss
    res, n_views = 64, 3
    root = tmp_path / "abc" / "toy"
    (root / "images").mkdir(parents=True)
    (root / "hawp").mkdir()
    intr, extr = [], []
    rng = np.random.default_rng(0)
    for v in range(n_views):
        sc = synth.synth_scene(seed=v, n_rays=4, res=res, view=v)
        K = sc["intrinsics"][0, :3, :3].copy()
        K[0, 0] = K[1, 1] = 70.0
        intr.append(K.astype(np.float64))
        extr.append(sc["pose"][0])
        Image.fromarray(rng.integers(0, 255, (res, res, 3), dtype=np.uint8)).save(root / "images" / f"image_{v:04d}.png")
        verts = rng.uniform(8, res - 8, (8, 2)).round(2).tolist()
        edges = [[0, 1], [1, 2], [2, 3], [3, 0], [4, 5], [5, 6]]
        json.dump({"vertices": verts, "vertices-score": [0.9] * 8, "edges": edges, "edges-weights": [0.99] * len(edges),
                   "height": res, "width": res}, open(root / "hawp" / f"image_{v:04d}.json", "w"))
    np.savez(root / "cameras.npz", intrinsics=np.stack(intr), extrinsics=np.stack(extr))
    ds = datasets.BlenderDataset("abc/toy", [res, res], data_root=str(tmp_path))
    assert len(ds) == n_views and ds.total_pixels == res * res
    assert all(m.any() for m in ds.masks)
    ds.change_sampling_idx(128)
    loader = torch.utils.data.DataLoader(ds, batch_size=1, shuffle=False, collate_fn=ds.collate_fn)
    idx, inp, gt = next(iter(loader))
    assert inp["uv"].shape == (1, 128, 2) and gt["lines2d"].shape == (1, 128, 5) and gt["rgb"].shape == (1, 128, 3)
    dev = torch.device("cuda:0")
    m = networks.VolSDFNetwork(synth.ABC_NEAT_A_MODEL_CONF).to(dev).train()
    for k in ("intrinsics", "uv", "pose", "uv_proj"):
        inp[k] = inp[k].to(dev)
    K4 = torch.eye(4, device=dev)[None].clone()
    K4[:, :3, :3] = inp["intrinsics"]
    inp["intrinsics"] = K4
    out = m(inp)
    lo = VolSDFLoss(**synth.ABC_NEAT_A_LOSS_CONF)(out, gt)
    lo["loss"].backward()
    assert torch.isfinite(lo["loss"])
