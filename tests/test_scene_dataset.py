"""SURVEY 8f-1 / VERDICT r1 missing #2: the DTU / BlendedMVS dataset class (reference: code/datasets/scene_hawp_dataset.py).
CPU part: the projection-matrix decomposition against the numpy oracle (Givens RQ, OpenCV's published algorithm) and against
K, R, C -> P -> K, R, C round trips.  GPU part: the dataset on a synthetic DTU-style scan directory, and the runner on it."""
import json

import numpy as np
import pytest
import torch


def _random_camera(rng):
    K = np.array([[rng.uniform(500, 3000), rng.uniform(-2, 2), rng.uniform(300, 900)], [0, rng.uniform(500, 3000), rng.uniform(200, 700)], [0, 0, 1.0]])
    A = rng.normal(size=(3, 3))
    R, _ = np.linalg.qr(A)
    if np.linalg.det(R) < 0:
        R[:, 0] = -R[:, 0]
    C = rng.uniform(-3, 3, 3)
    return K, R, C


def test_projection_decomposition_round_trip_and_oracle():
    from neat_amd.datasets import load_K_Rt_from_P
    from oracle import camera_oracle
    rng = np.random.default_rng(0)
    for i in range(50):
        K, R, C = _random_camera(rng)
        P = K @ np.concatenate([R, (-R @ C)[:, None]], 1) * rng.uniform(0.1, 10.0)          # arbitrary positive scale
        intr, pose = load_K_Rt_from_P(P)
        assert np.allclose(intr[:3, :3], K, rtol=1e-9, atol=1e-7), i
        assert np.allclose(pose[:3, :3], R.T, atol=1e-6) and np.allclose(pose[:3, 3], C, atol=1e-5), i
        o_intr, o_pose = camera_oracle.load_K_Rt_from_P(P)
        assert np.allclose(intr, o_intr, rtol=1e-8, atol=1e-6) and np.allclose(pose, o_pose, atol=1e-5), i
    # a DTU-style pair: world_mat (K [R|t] padded to 4x4) times scale_mat (uniform scale + shift), float32 like the dataset loads it
    K, R, C = _random_camera(rng)
    world = np.eye(4); world[:3] = K @ np.concatenate([R, (-R @ C)[:, None]], 1)
    scale = np.eye(4); scale[:3, :3] *= 250.0; scale[:3, 3] = [10.0, -20.0, 600.0]
    P = (world.astype(np.float32) @ scale.astype(np.float32))[:3, :4]
    intr, pose = load_K_Rt_from_P(P)
    o_intr, o_pose = camera_oracle.load_K_Rt_from_P(P)
    assert np.allclose(intr, o_intr, rtol=1e-5, atol=1e-3) and np.allclose(pose, o_pose, atol=1e-4)
    assert np.allclose(intr[:3, :3], K, rtol=1e-4, atol=1e-2)          # scaling the world does not change K
    assert np.allclose(pose[:3, 3], (C - scale[:3, 3]) / 250.0, atol=1e-3)


def _toy_scan(root, res=(48, 64), n_views=3):
    from PIL import Image
    scan = root / "DTU" / "scan7"
    (scan / "image").mkdir(parents=True)
    (scan / "hawp").mkdir()
    rng = np.random.default_rng(1)
    cams = {}
    H, W = res
    for v in range(n_views):
        ang = 0.7 * v
        C = np.array([2.2 * np.cos(ang), 2.2 * np.sin(ang), 0.4])
        z = -C / np.linalg.norm(C)
        x = np.cross([0, 0, 1.0], z); x /= np.linalg.norm(x)
        y = np.cross(z, x)
        R = np.stack([x, y, z])                                    # world -> camera
        K = np.array([[60.0, 0, W / 2], [0, 60.0, H / 2], [0, 0, 1]])
        world = np.eye(4); world[:3] = K @ np.concatenate([R, (-R @ C)[:, None]], 1)
        cams[f"world_mat_{v}"] = world
        cams[f"scale_mat_{v}"] = np.eye(4)
        Image.fromarray(rng.integers(0, 255, (H, W, 3), dtype=np.uint8)).save(scan / "image" / f"{v:06d}.png")
        verts = np.stack([rng.uniform(6, W - 6, 8), rng.uniform(6, H - 6, 8)], 1).round(2).tolist()
        edges = [[0, 1], [1, 2], [2, 3], [3, 0], [4, 5], [5, 6]]
        json.dump({"vertices": verts, "vertices-score": [0.9] * 8, "edges": edges, "edges-weights": [0.99] * len(edges),
                   "height": H, "width": W}, open(scan / "hawp" / f"{v:06d}.json", "w"))
    np.savez(scan / "cameras.npz", **cams)
    return cams


@pytest.mark.gpu
def test_scene_dataset_and_runner(tmp_path):
    from neat_amd import synth
    from neat_amd.datasets import SceneDataset
    from neat_amd.runner import TrainRunner
    from neat_amd.synth import hocon_text as _hocon
    cams = _toy_scan(tmp_path / "data")
    ds = SceneDataset("DTU", [48, 64], scan_id=7, data_root=str(tmp_path / "data"))
    assert len(ds) == 3 and ds.total_pixels == 48 * 64
    idx, sample, gt = ds[1]
    assert sample["uv"].shape == (48 * 64, 2) and sample["intrinsics"].shape == (4, 4) and sample["pose"].shape == (4, 4)
    assert torch.allclose(sample["intrinsics"][:3, :3], torch.tensor([[60.0, 0, 32], [0, 60.0, 24], [0, 0, 1]]), atol=1e-3)
    assert abs(float(sample["pose"][:3, 3].norm()) - np.sqrt(2.2 ** 2 + 0.4 ** 2)) < 1e-3
    ds.change_sampling_idx(96)
    _, s2, g2 = ds[1]
    assert s2["uv"].shape == (96, 2) and g2["rgb"].shape == (96, 3) and g2["lines2d"].shape[0] == 96
    pix = (s2["uv"][:, 1] * 64 + s2["uv"][:, 0]).long()
    assert pix.unique().numel() == 96 and bool(ds.masks[1][pix].all())          # without replacement, inside the line support
    assert np.allclose(ds.get_scale_mat(), cams["scale_mat_0"])
    # the device path: n distinct pixels of the support, every field gathered from the same pixel
    db = ds.device_batches(torch.device("cuda:0"))
    _, s3, g3 = db.batch(1, 96)
    pix = s3["pixels"][0].cpu()
    assert pix.unique().numel() == 96 and bool(ds.masks[1][pix].all())
    assert torch.equal(s3["uv"][0].cpu(), torch.stack([pix % 64, pix // 64], -1).float())
    assert torch.equal(g3["rgb"][0].cpu(), ds.rgb_images[1][pix]) and torch.equal(s3["uv_proj"][0].cpu(), ds.att_points[1].cpu()[pix])
    assert torch.equal(g3["lines2d"][0].cpu(), ds.lines[1][ds.labels[1][pix]]) and torch.equal(s3["labels"][0].cpu(), ds.labels[1][pix])
    assert torch.equal(s3["intrinsics"][0].cpu(), ds.intrinsics_all[1]) and torch.equal(s3["pose"][0].cpu(), ds.pose_all[1])
    assert db.batch(1, 10 ** 6)[1]["uv"].shape[1] == int(ds.masks[1].sum())      # more rays than support pixels: all of them, once
    # the reference's dtu.conf names this dataset class; the runner maps it and trains on the scan
    model_conf = dict(synth.ABC_NEAT_A_MODEL_CONF)
    model_conf.update(dbscan_enabled=True, use_median=False)
    conf = {"train": {"expname": "toy_dtu", "dataset_class": "datasets.scene_hawp_dataset.SceneDataset",
                      "model_class": "model.networks.neat_wfr_rend_a.VolSDFNetwork", "loss_class": "model.networks.loss_wfr.VolSDFLoss",
                      "learning_rate": 5.0e-4, "num_pixels": 64, "checkpoint_freq": 1},
            "loss": dict(synth.ABC_NEAT_A_LOSS_CONF),
            "dataset": {"data_dir": "DTU", "img_res": [48, 64], "scan_id": 7},
            "model": model_conf}
    path = tmp_path / "toy_dtu.conf"
    path.write_text(_hocon(conf))
    runner = TrainRunner(str(path), nepochs=1, exps_folder=str(tmp_path / "exps"), data_root=str(tmp_path / "data"), log_freq=1)
    assert type(runner.train_dataset).__name__ == "SceneDataset"
    hist = runner.run()
    assert len(hist) >= 3 and all(np.isfinite(h[2]) for h in hist)


def test_draw_rules_of_the_device_batches():
    """The host side of datasets.DeviceBatches (no GPU needed): the ABC class draws the pixels np.random.choice would draw from the same
    numpy state; the DTU class draws n DISTINCT pool entries, every n-subset order equally likely, and never more than the pool holds."""
    from collections import Counter
    from neat_amd.datasets import BlenderDataset, SceneDataset
    b, s = BlenderDataset.__new__(BlenderDataset), SceneDataset.__new__(SceneDataset)
    pool = torch.arange(100, 1000, 3)
    for seed, n in ((0, 1), (1, 64), (2, 777)):
        np.random.seed(seed)
        ref = np.random.choice(pool, n)
        np.random.seed(seed)
        got = pool[b.draw_rays(pool.numel(), n)]
        assert got.dtype == torch.int64 and np.array_equal(got.numpy(), ref)
    torch.manual_seed(0)
    for npool, n in ((50, 50), (50, 40), (100000, 2048), (5, 9), (1, 1)):
        r = s.draw_rays(npool, n)
        assert r.dtype == torch.int64 and r.numel() == min(n, npool) and r.unique().numel() == r.numel()
        assert int(r.min()) >= 0 and int(r.max()) < npool
    counts = Counter(tuple(s.draw_rays(4, 2).tolist()) for _ in range(12000))
    assert len(counts) == 12 and min(counts.values()) > 800 and max(counts.values()) < 1200      # 12 ordered pairs, 1000 expected each


def test_quat_to_rot_matches_scipy():
    """rend_util.quat_to_rot (reference: utils/rend_util.py:111-128, (w, x, y, z) order, normalised first) against scipy's Rotation
    (an independent implementation; (x, y, z, w) order) on random, un-normalised quaternions."""
    import numpy as np
    import torch
    from scipy.spatial.transform import Rotation
    from neat_amd import rend_util
    rng = np.random.default_rng(3)
    q = rng.normal(size=(64, 4)) * rng.uniform(0.1, 5.0, size=(64, 1))
    got = rend_util.quat_to_rot(torch.tensor(q)).numpy()
    ref = Rotation.from_quat(q[:, [1, 2, 3, 0]]).as_matrix()
    assert np.abs(got - ref).max() < 1e-12
    assert np.abs(got @ got.transpose(0, 2, 1) - np.eye(3)).max() < 1e-12


@pytest.mark.gpu
def test_quaternion_pose_gives_the_matrix_pose_rays():
    """get_camera_params with the reference's quaternion pose form [1,7] (rend_util.py:56-61) = the same rays as the 4x4 matrix of the
    same camera through the HIP ray kernel."""
    import numpy as np
    import torch
    from scipy.spatial.transform import Rotation
    from neat_amd import rend_util
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(5)
    rot = Rotation.from_rotvec(rng.normal(size=3))
    c = rng.uniform(-2, 2, size=3)
    pose = np.eye(4, dtype=np.float32)
    pose[:3, :3] = rot.as_matrix()
    pose[:3, 3] = c
    qx = rot.as_quat()                                   # (x, y, z, w)
    pose7 = np.concatenate([[qx[3]], qx[:3], c]).astype(np.float32)[None]
    K = np.array([[500.0, 0.5, 256.0, 0], [0, 480.0, 250.0, 0], [0, 0, 1, 0], [0, 0, 0, 1]], dtype=np.float32)[None]
    uv = torch.tensor(rng.uniform(0, 512, size=(1, 300, 2)).astype(np.float32)).to(dev)
    d0, c0 = rend_util.get_camera_params(uv, torch.tensor(pose[None]).to(dev), torch.tensor(K).to(dev))
    d1, c1 = rend_util.get_camera_params(uv, torch.tensor(pose7).to(dev), torch.tensor(K).to(dev))
    assert float((d0 - d1).abs().max()) < 2e-6 and float((c0 - c1).abs().max()) < 1e-6
