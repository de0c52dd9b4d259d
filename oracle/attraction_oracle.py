"""Brute-force CPU oracle for the dataset attraction field.  TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED: the reference delegates this to `hawp.base._C.encodels` (third-party submodule cherubicXN/hawp,
directory empty in /root/reference, no pinned SHA), so there is nothing to run or compare against.  This restates the
contract its call sites rely on (code/datasets/blender_hawp_dataset.py:93-146): per pixel the nearest segment, the
vector to the closest point on it and to its two endpoints; then the support test of `compute_point_line_attraction`."""
import numpy as np


def encode_lines(lines, H, W):
    ys, xs = np.mgrid[0:H, 0:W].astype(np.float32)
    best = np.full((H, W), np.inf, np.float32)
    lmap = np.zeros((6, H, W), np.float32)
    label = np.zeros((H, W), np.int64)
    for j, (x1, y1, x2, y2) in enumerate(np.asarray(lines, np.float32)[:, :4]):
        dx, dy = np.float32(x2 - x1), np.float32(y2 - y1)
        len2 = dx * dx + dy * dy
        t = ((xs - x1) * dx + (ys - y1) * dy) / len2 if len2 > 0 else np.zeros_like(xs)
        t = np.clip(t, 0.0, 1.0).astype(np.float32)
        qx, qy = x1 + t * dx, y1 + t * dy
        d2 = (qx - xs) ** 2 + (qy - ys) ** 2
        better = d2 < best
        best = np.where(better, d2, best)
        label = np.where(better, j, label)
        for c, v in enumerate((qx - xs, qy - ys, x1 - xs, y1 - ys, x2 - xs, y2 - ys)):
            lmap[c] = np.where(better, v, lmap[c])
    return lmap, label, best


def valid_map(best):
    """The validity map the call sites take from `labels_onehot.max(dim=0)[0]` (:98): the pixel has a nearest segment."""
    return np.isfinite(best)


def support(lmap, distance, valid=None):
    """mask of pixels that support their nearest segment (:104-140; `valid` multiplied in as at :130), and their foot points."""
    H, W = lmap.shape[1:]
    mag = np.sqrt(lmap[0] ** 2 + lmap[1] ** 2)
    md = lmap[:2] / (mag + 1e-6)
    st, ed = lmap[2:4], lmap[4:6]
    rot = lambda v: np.stack([md[0] * v[0] + md[1] * v[1], -md[1] * v[0] + md[0] * v[1]])
    a, b = rot(st), rot(ed)
    swap = (a[1] < 0) & (b[1] > 0)
    pos, neg = np.where(swap, b, a), np.where(swap, a, b)
    pos = np.stack([np.maximum(pos[0], 1e-9), np.maximum(pos[1], 1e-9)])
    neg = np.stack([np.maximum(neg[0], 1e-9), np.minimum(neg[1], -1e-9)])
    mask = (mag <= distance) & (np.arctan2(pos[1], pos[0]) > 0) & (np.arctan2(neg[1], neg[0]) < 0)
    if valid is not None:
        mask &= valid
    ys, xs = np.mgrid[0:H, 0:W].astype(np.float32)
    foot = np.stack([lmap[0] + xs, lmap[1] + ys], -1) * mask[..., None]
    return mask, foot
