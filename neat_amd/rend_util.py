"""Ray helpers with the reference's names (code/utils/rend_util.py).  Only what is on the hot path:
get_camera_params (:55-81, HIP kernel; quaternion poses through quat_to_rot, :111-128), get_psnr (:9-17), get_sphere_intersections (:152-168)."""
import math

import torch

from . import ops


def quat_to_rot(q):
    """Unit quaternions (w, x, y, z) [B,4] -> rotation matrices [B,3,3]  (rend_util.py:111-128: normalised first, same element formulas)."""
    q = torch.nn.functional.normalize(q, dim=1)
    qr, qi, qj, qk = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    rows = [1 - 2 * (qj ** 2 + qk ** 2), 2 * (qj * qi - qk * qr), 2 * (qi * qk + qr * qj),
            2 * (qj * qi + qk * qr), 1 - 2 * (qi ** 2 + qk ** 2), 2 * (qj * qk - qi * qr),
            2 * (qk * qi - qj * qr), 2 * (qj * qk + qi * qr), 1 - 2 * (qi ** 2 + qj ** 2)]
    return torch.stack(rows, dim=1).reshape(-1, 3, 3)


def get_camera_params(uv, pose, intrinsics, normalize=True):
    """pixel (u,v) -> unit ray direction in world space + camera centre.  uv [1,R,2] -> ([1,R,3], [1,3]).
    pose: a 4x4 camera-to-world matrix [1,4,4], or the reference's quaternion form [1,7] = (w, x, y, z, cx, cy, cz) (:56-61), which is
    turned into the matrix on the device (a handful of tiny torch ops: no shipped dataset uses it) before the ray kernel."""
    if not normalize:
        raise NotImplementedError("un-normalised ray directions are not used on the hot path")
    if pose.dim() == 2 and pose.shape[1] == 7:
        p = torch.eye(4, device=pose.device, dtype=pose.dtype).repeat(pose.shape[0], 1, 1)
        p[:, :3, :3] = quat_to_rot(pose[:, :4])
        p[:, :3, 3] = pose[:, 4:]
        pose = p
    return ops.camera_rays(uv, pose, intrinsics)


def get_psnr(img1, img2, normalize_rgb=False):
    if normalize_rgb:
        img1, img2 = (img1 + 1.0) / 2.0, (img2 + 1.0) / 2.0
    mse = torch.mean((img1 - img2) ** 2)
    return -10.0 * torch.log(mse) / math.log(10.0)


def get_sphere_intersections(cam_loc, ray_directions, r=1.0):
    """near/far hits of rays with the radius-r sphere, clamped at 0: [R,2]."""
    b = (ray_directions * cam_loc).sum(-1, keepdim=True)
    disc = b ** 2 - (cam_loc.norm(2, 1, keepdim=True) ** 2 - r ** 2)
    if bool((disc <= 0).any()):
        raise RuntimeError("a ray misses the scene bounding sphere (the reference prints BOUNDING SPHERE PROBLEM and exits)")
    root = torch.sqrt(disc)
    return torch.cat([-root - b, root - b], dim=-1).clamp_min(0.0)
