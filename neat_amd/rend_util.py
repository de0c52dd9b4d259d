"""Ray helpers with the reference's names (code/utils/rend_util.py).  Only what is on the hot path:
get_camera_params (:55-81, HIP kernel), get_psnr (:9-17), get_sphere_intersections (:152-168)."""
import math

import torch

from . import ops


def get_camera_params(uv, pose, intrinsics, normalize=True):
    """pixel (u,v) -> unit ray direction in world space + camera centre.  uv [1,R,2] -> ([1,R,3], [1,3])."""
    if not normalize:
        raise NotImplementedError("un-normalised ray directions are not used on the hot path")
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
