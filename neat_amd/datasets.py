"""Dataset side of the hot path: per-pixel 2-D attraction field and the ABC/Blender dataset class with the
reference's constructor and sample layout (code/datasets/blender_hawp_dataset.py).  The per-pixel nearest-segment search
that the reference delegates to the un-vendored CUDA op `hawp.base._C.encodels` is a HIP kernel here
(`neat_encode_lines`); images are read with PIL (imageio / skimage / cv2 are not needed).

PARITY UNPINNED at the encodels boundary: the hawp submodule is empty in the reference tree, so its exact tie-breaking
cannot be checked; the semantics implemented are the ones the call sites depend on (see include/neat_hip.h) and are
tested against a brute-force numpy oracle (oracle/attraction_oracle.py).  The validity map `labels_onehot.max(dim=0)[0]` that
encodels returns and the reference multiplies into the support mask (blender_hawp_dataset.py:98,130) is the kernel's `valid`
output: 1 where a pixel has a nearest segment (all ones as soon as one finite segment exists, all zeros for an empty set).

`SceneDataset` is the DTU / BlendedMVS counterpart (code/datasets/scene_hawp_dataset.py): cameras come as projection
matrices `world_mat_i @ scale_mat_i` and are decomposed into K and pose; the reference uses
`cv2.decomposeProjectionMatrix` (opencv-python, unpinned in requirements.txt, absent here), restated below as an RQ
decomposition with positive diagonal."""
import ctypes
import json
import os
from glob import glob

import numpy as np
import torch

from . import _lib
from .wireframe import WireframeGraph


def encode_lines(lines, height, width, return_valid=False):
    """lines [N,4] (x1,y1,x2,y2) on the GPU -> lmap [6,H,W] float32, labels [H,W] int64 (, valid [H,W] bool: the pixel has a
    nearest segment = the reference's labels_onehot.max(dim=0)[0])."""
    lib = _lib.lib()
    if not lines.is_cuda:
        raise RuntimeError("encode_lines needs CUDA tensors (no CPU path)")
    lines = lines.detach().float().contiguous()
    lmap = torch.empty(6, height, width, device=lines.device)
    label = torch.empty(height, width, device=lines.device, dtype=torch.int32)
    valid = torch.empty(height, width, device=lines.device, dtype=torch.uint8)
    _lib.check(lib.neat_encode_lines(ctypes.c_void_p(lines.data_ptr()) if lines.shape[0] else None, lines.shape[0], height, width,
                                     ctypes.c_void_p(lmap.data_ptr()), ctypes.c_void_p(label.data_ptr()), ctypes.c_void_p(valid.data_ptr()),
                                     ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "neat_encode_lines")
    return (lmap, label.long(), valid.bool()) if return_valid else (lmap, label.long())


def _unit(v):
    return v / (torch.sqrt(v[0] * v[0] + v[1] * v[1]) + 1e-6)


def compute_point_line_attraction(lines, img_res, distance=10.0):
    """Support mask, nearest-segment label and foot point per pixel (blender_hawp_dataset.py:93-146).
    A pixel supports its nearest segment when it is within `distance` px; the reference's two angle tests on the
    endpoints rotated into the foot-point frame are kept for fidelity, but its clamps (:125-128) make them always pass."""
    H, W = img_res
    lmap, labels, valid = encode_lines(lines[:, :4].cuda(), H, W, return_valid=True)
    dist = torch.sqrt(lmap[0] ** 2 + lmap[1] ** 2)
    md = _unit(lmap[:2]).reshape(2, -1)
    st, ed = lmap[2:4].reshape(2, -1), lmap[4:6].reshape(2, -1)
    # rotate both endpoint vectors by R^T, R = [[md_x, -md_y], [md_y, md_x]]
    st_r = torch.stack([md[0] * st[0] + md[1] * st[1], -md[1] * st[0] + md[0] * st[1]])
    ed_r = torch.stack([md[0] * ed[0] + md[1] * ed[1], -md[1] * ed[0] + md[0] * ed[1]])
    swap = (st_r[1] < 0) & (ed_r[1] > 0)
    pos = torch.where(swap, ed_r, st_r)
    neg = torch.where(swap, st_r, ed_r)
    pos = torch.stack([pos[0].clamp(min=1e-9), pos[1].clamp(min=1e-9)])
    neg = torch.stack([neg[0].clamp(min=1e-9), neg[1].clamp(max=-1e-9)])
    mask = (dist <= distance).reshape(-1) & valid.reshape(-1)      # `mask, labels = labels_onehot.max(dim=0)`; `mask = (dismap <= d) * mask` (:98,130)
    mask &= torch.atan2(pos[1], pos[0]) > 0
    mask &= torch.atan2(neg[1], neg[0]) < 0
    ys, xs = torch.meshgrid(torch.arange(H, device=lmap.device), torch.arange(W, device=lmap.device), indexing="ij")
    foot = torch.stack([lmap[0] + xs, lmap[1] + ys], -1).reshape(-1, 2)
    foot = torch.where(mask[:, None], foot, torch.zeros_like(foot))
    return mask.cpu(), labels.reshape(-1).cpu(), foot.float()


def load_K_Rt_from_P(P):
    """P [3,4] -> (intrinsics [4,4] with K / K[2,2], pose [4,4] camera-to-world)  (utils/rend_util.py:31-52, which calls
    cv2.decomposeProjectionMatrix).  M = P[:, :3] = K R with K upper triangular, positive diagonal (RQ decomposition, signs
    fixed the way OpenCV's RQDecomp3x3 fixes them), camera centre C = -M^-1 p4; pose = [R^T | C]."""
    P = np.asarray(P, dtype=np.float64)
    M = P[:3, :3]
    # RQ via QR of the row-reversed transpose:  J M^T J... written out: M = K R  <=>  (J M)^T = (J R)^T (J K J)^T ...
    J = np.eye(3)[::-1]
    q, r = np.linalg.qr((J @ M).T)              # (J M)^T = q r  ->  M = J r^T q^T = (J r^T J) (J q^T)
    K = J @ r.T @ J
    R = J @ q.T
    D = np.diag(np.where(np.diag(K) < 0, -1.0, 1.0))
    K, R = K @ D, D @ R                         # positive diagonal; D D = I keeps the product
    if np.linalg.det(R) < 0:                    # det(M) < 0: OpenCV leaves the sign in K[2,2]; keep R a rotation
        R = -R
        K = -K
    C = -np.linalg.solve(M, P[:3, 3])
    intrinsics = np.eye(4)
    intrinsics[:3, :3] = K / K[2, 2]
    pose = np.eye(4, dtype=np.float32)
    pose[:3, :3] = R.T
    pose[:3, 3] = C
    return intrinsics, pose


def load_rgb(path):
    """[3,H,W] float32 in [0,1] (rend_util.load_rgb without imageio/skimage)."""
    from PIL import Image
    img = np.asarray(Image.open(path).convert("RGB"), dtype=np.float32) / 255.0
    return img.transpose(2, 0, 1)


class BlenderDataset(torch.utils.data.Dataset):
    """ABC / Blender scenes: images/*.png, cameras.npz (intrinsics, extrinsics), <line_detector>/*.json wireframes."""

    def __init__(self, data_dir, img_res, reverse_coordinate=False, line_detector="hawp", distance_threshold=10.0,
                 data_root="../data"):
        self.instance_dir = os.path.join(data_root, data_dir)
        assert os.path.exists(self.instance_dir), "Data directory is empty"
        self.img_res = list(img_res)
        self.total_pixels = img_res[0] * img_res[1]
        self.sampling_idx = None
        self.distance = distance_threshold
        self.score_threshold = 0.05
        paths = []
        for ext in ("*.png", "*.jpg", "*.JPEG", "*.JPG"):
            paths += glob(os.path.join(self.instance_dir, "images", ext))
        paths = [p for p in sorted(paths) if "mask" not in p]
        cams = np.load(os.path.join(self.instance_dir, "cameras.npz"))
        intr, pose = torch.from_numpy(cams["intrinsics"]).float(), torch.from_numpy(cams["extrinsics"]).float()
        self.rgb_images, self.wireframes, self.lines = [], [], []
        keep = []
        for i, path in enumerate(paths):
            wf = WireframeGraph.load_json(os.path.join(self.instance_dir, line_detector,
                                                       os.path.splitext(os.path.basename(path))[0] + ".json"))
            if wf.vertices.shape[0] == 0 or wf.edges.shape[0] == 0 or wf.line_segments(self.score_threshold).shape[0] == 0:
                continue
            assert wf.frame_height == img_res[0] and wf.frame_width == img_res[1]
            keep.append(i)
            self.rgb_images.append(torch.from_numpy(load_rgb(path).reshape(3, -1).transpose(1, 0).copy()).float())
            self.wireframes.append(wf)
            self.lines.append(wf.line_segments(self.score_threshold))
        self.intrinsics_all, self.pose_all = intr[keep], pose[keep]
        self.n_images = len(keep)
        sign = [1, -1, -1, 1] if reverse_coordinate else [1, 1, 1, 1]
        self.normalization = torch.diag(torch.tensor(sign)).float()
        self.masks, self.labels, self.att_points = [], [], []
        for lines in self.lines:          # precompute the support regions of the 2-D attraction fields (HIP kernel)
            m, l, a = compute_point_line_attraction(lines, self.img_res, self.distance)
            self.masks.append(m)
            self.labels.append(l)
            self.att_points.append(a)          # stays on the device, like the reference's `uv_proj` (trainers do not move it)

    def __len__(self):
        return self.n_images

    def __getitem__(self, idx):
        H, W = self.img_res
        ys, xs = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
        uv = torch.stack([xs, ys], -1).reshape(-1, 2).float()
        lines, mask, labels = self.lines[idx], self.masks[idx], self.labels[idx]
        sample = {"uv": uv, "uv_proj": self.att_points[idx], "juncs2d": self.wireframes[idx].vertices,
                  "intrinsics": self.intrinsics_all[idx], "pose": self.pose_all[idx], "wireframe": self.wireframes[idx],
                  "mask": mask, "labels": labels, "lines": lines[labels], "lines_uniq": lines}
        gt = {"rgb": self.rgb_images[idx]}
        if self.sampling_idx is not None:      # rays only inside the line support, with replacement (:186-198)
            pool = mask.nonzero().flatten()
            pick = np.random.choice(pool, len(self.sampling_idx))
            gt["rgb"] = self.rgb_images[idx][pick, :]
            gt["lines2d"] = lines[labels[pick]]
            sample.update(lines=lines[labels[pick]], labels=labels[pick], uv=uv[pick, :], uv_proj=self.att_points[idx][pick])
        return idx, sample, gt

    def collate_fn(self, batch_list):
        parsed = []
        for entry in zip(*batch_list):
            if isinstance(entry[0], dict):
                parsed.append({k: torch.stack([o[k] for o in entry]) if isinstance(entry[0][k], torch.Tensor)
                               else [o[k] for o in entry] for k in entry[0]})
            else:
                parsed.append(torch.LongTensor(entry))
        return tuple(parsed)

    def change_sampling_idx(self, sampling_size):
        self.sampling_idx = None if sampling_size == -1 else torch.randperm(self.total_pixels)[:sampling_size]

    def draw_rays(self, npool, n):
        """The step's n draws into a view's support pool, as int64 on the host.  `np.random.choice(pool, n)` of __getitem__ IS
        `pool[np.random.randint(0, len(pool), n)]` on numpy's global stream: the device path picks the same pixels."""
        return torch.from_numpy(np.random.randint(0, npool, n).astype(np.int64))

    def device_batches(self, device):
        return DeviceBatches(self, device)

    def get_scale_mat(self):
        return np.eye(4)


class DeviceBatches:
    """Training batches assembled on the device ("a GPU path for the __getitem__ sampling").

    `Dataset.__getitem__` (datasets/blender_hawp_dataset.py:159-198) rebuilds an [HW,2] pixel grid, runs `mask.nonzero()`, gathers
    `lines[labels]` for EVERY pixel and then the n sampled rows, all on the host, every step: 1 ms at 512 x 512, 6+ ms at DTU's
    1200 x 1600 -- longer than the whole train step takes on the GPU (3.4 ms).  Here a view's maps (support pool, foot points,
    colours, labels, segments, camera) are uploaded once and stay in HBM (23 MB per DTU image); per step the host makes the n draws
    (`Dataset.draw_rays`: the same RNG stream and therefore the same pixels as __getitem__), sends 8 KB, and ONE launch
    (`neat_gather_batch`) writes uv, uv_proj, rgb, lines2d and labels.  `batch()` returns what the DataLoader's collate returns for
    batch size 1 -- same keys, shapes and values (tests/test_gpu_parity.py::test_device_batches_equal_getitem), tensors on the device."""

    def __init__(self, dataset, device):
        self.ds, self.device = dataset, torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("DeviceBatches needs a CUDA device (no CPU path; use the Dataset's __getitem__)")
        self._views = {}

    def _view(self, idx):
        v = self._views.get(idx)
        if v is None:
            ds, dev = self.ds, self.device
            pool = ds.masks[idx].nonzero().flatten()
            if pool.numel() == 0:
                raise RuntimeError(f"view {idx}: empty line support")
            lines = ds.lines[idx].float()
            v = self._views[idx] = {
                "npool": int(pool.numel()), "pool": pool.to(torch.int32).to(dev), "att": ds.att_points[idx].to(dev).float().contiguous(),
                "rgb": ds.rgb_images[idx].to(dev).float().contiguous(), "labels": ds.labels[idx].to(torch.int32).to(dev),
                "lines": lines.to(dev).contiguous(), "intrinsics": ds.intrinsics_all[idx].to(dev)[None], "pose": ds.pose_all[idx].to(dev)[None],
                "juncs2d": ds.wireframes[idx].vertices[None], "mask": ds.masks[idx][None], "lines_uniq": ds.lines[idx][None]}
        return v

    def batch(self, idx, n):
        """-> (indices [1], model_input, ground_truth) for view `idx` with n rays drawn from its line support."""
        from .networks import _to_device_async
        v = self._view(idx)
        dev = self.device
        draw = self.ds.draw_rays(v["npool"], n)
        n = draw.numel()                                   # (drawing without replacement cannot exceed the pool)
        draw = _to_device_async(draw, dev, site="dataset.draw")
        uv, uv_proj = torch.empty(1, n, 2, device=dev), torch.empty(1, n, 2, device=dev)
        rgb, lines = torch.empty(1, n, 3, device=dev), torch.empty(1, n, 5, device=dev)
        labels, pixels = torch.empty(1, n, device=dev, dtype=torch.int64), torch.empty(1, n, device=dev, dtype=torch.int64)
        P = lambda t: ctypes.c_void_p(t.data_ptr())
        _lib.check(_lib.lib().neat_gather_batch(P(v["pool"]), v["npool"], P(draw), n, self.ds.img_res[1], P(v["att"]), P(v["rgb"]),
                                                P(v["labels"]), P(v["lines"]), v["lines"].shape[0], P(uv), P(uv_proj), P(rgb), P(lines),
                                                P(labels), P(pixels), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)),
                   "neat_gather_batch")
        sample = {"uv": uv, "uv_proj": uv_proj, "juncs2d": v["juncs2d"], "intrinsics": v["intrinsics"], "pose": v["pose"],
                  "wireframe": [self.ds.wireframes[idx]], "mask": v["mask"], "labels": labels, "lines": lines, "lines_uniq": v["lines_uniq"],
                  "pixels": pixels}
        return torch.LongTensor([idx]), sample, {"rgb": rgb, "lines2d": lines}


class SceneDataset(BlenderDataset):
    """DTU / BlendedMVS scans: <data_dir>/scan<id>/{image/*.png, cameras.npz (world_mat_i, scale_mat_i), <line_detector>/*.json}
    (code/datasets/scene_hawp_dataset.py:16-223).  Differences from the ABC class, as in the reference: cameras from projection
    matrices; wireframes of all images are kept (no empty-wireframe filter); rays of a step are drawn WITHOUT replacement from
    the line support (`torch.randperm`, :182); `n_images` may cut the epoch length; default support distance 5 px."""

    def __init__(self, data_dir, img_res, scan_id=0, n_images=-1, line_detector="hawp", distance_threshold=5.0, data_root="../data"):
        self.instance_dir = os.path.join(data_root, data_dir, "scan{0}".format(scan_id))
        assert os.path.exists(self.instance_dir), "Data directory is empty"
        self.img_res = list(img_res)
        self.total_pixels = img_res[0] * img_res[1]
        self.sampling_idx = None
        self.distance = distance_threshold
        self.score_threshold = 0.05
        paths = []
        for ext in ("*.png", "*.jpg", "*.JPEG", "*.JPG"):
            paths += glob(os.path.join(self.instance_dir, "image", ext))
        paths = sorted(paths)
        self.n_images = len(paths)
        self.cam_file = os.path.join(self.instance_dir, "cameras.npz")
        cams = np.load(self.cam_file)
        self.intrinsics_all, self.pose_all = [], []
        for i in range(self.n_images):
            P = (cams["world_mat_%d" % i].astype(np.float32) @ cams["scale_mat_%d" % i].astype(np.float32))[:3, :4]
            K, pose = load_K_Rt_from_P(P)
            self.intrinsics_all.append(torch.from_numpy(K).float())
            self.pose_all.append(torch.from_numpy(pose).float())
        self.rgb_images, self.wireframes, self.lines = [], [], []
        for path in paths:
            self.rgb_images.append(torch.from_numpy(load_rgb(path).reshape(3, -1).transpose(1, 0).copy()).float())
            wf = WireframeGraph.load_json(os.path.join(self.instance_dir, line_detector, os.path.splitext(os.path.basename(path))[0] + ".json"))
            assert wf.frame_height == img_res[0] and wf.frame_width == img_res[1]
            self.wireframes.append(wf)
            self.lines.append(wf.line_segments(self.score_threshold))
        if n_images > 0:
            self.n_images = n_images
        self.masks, self.labels, self.att_points = [], [], []
        for lines in self.lines:
            m, l, a = compute_point_line_attraction(lines, self.img_res, self.distance)
            self.masks.append(m)
            self.labels.append(l)
            self.att_points.append(a)

    def __getitem__(self, idx):
        H, W = self.img_res
        ys, xs = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
        uv = torch.stack([xs, ys], -1).reshape(-1, 2).float()
        lines, mask, labels = self.lines[idx], self.masks[idx], self.labels[idx]
        sample = {"uv": uv, "uv_proj": self.att_points[idx], "juncs2d": self.wireframes[idx].vertices,
                  "intrinsics": self.intrinsics_all[idx], "pose": self.pose_all[idx], "wireframe": self.wireframes[idx],
                  "mask": mask, "labels": labels, "lines": lines[labels], "lines_uniq": lines}
        gt = {"rgb": self.rgb_images[idx]}
        if self.sampling_idx is not None:      # rays inside the line support, without replacement (:179-182)
            pool = mask.nonzero().flatten()
            pick = pool[torch.randperm(pool.numel())[:len(self.sampling_idx)]]
            gt["rgb"] = self.rgb_images[idx][pick, :]
            gt["lines2d"] = lines[labels[pick]]
            sample.update(lines=lines[labels[pick]], labels=labels[pick], uv=uv[pick, :], uv_proj=self.att_points[idx][pick.to(self.att_points[idx].device)])
        return idx, sample, gt

    def draw_rays(self, npool, n):
        """n DISTINCT draws into the pool, uniformly (the reference takes the first n of `torch.randperm(len(pool))`, :182: O(pool) host
        work per step).  Drawing with replacement and keeping first occurrences until n distinct ones are in hand is the same
        distribution -- a uniformly random n-subset in uniformly random order -- in O(n); it reads torch's CPU stream differently
        from randperm, so the device path picks other pixels than __getitem__ would from the same seed."""
        n = min(n, npool)
        got = np.empty(0, dtype=np.int64)
        while got.size < n:
            cand = np.concatenate([got, torch.randint(npool, (n - got.size + 16,)).numpy()])
            _, first = np.unique(cand, return_index=True)
            got = cand[np.sort(first)]
        return torch.from_numpy(got[:n].copy())

    def get_scale_mat(self):
        return np.load(self.cam_file)["scale_mat_0"]
