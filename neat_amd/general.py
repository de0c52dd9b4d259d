"""Plug-in loader and ray chunking helpers with the reference's names (code/utils/general.py:9-15,23-52)."""
import importlib

import torch


def get_class(path):
    """'pkg.mod.Class' -> class object.  Same contract as the reference's loader (general.py:9-15), so a conf can
    point train.model_class at neat_amd.networks.VolSDFNetwork without touching reference code."""
    module, _, name = path.rpartition(".")
    return getattr(importlib.import_module(module), name)


def split_input(model_input, total_pixels, n_pixels=10000, keys=("uv", "uv_proj")):
    """Chunk the per-pixel entries of a model input into pieces of at most n_pixels rays (general.py:23-38)."""
    chunks = []
    some = model_input[keys[0]]
    for idx in torch.split(torch.arange(total_pixels, device=some.device), n_pixels, dim=0):
        piece = dict(model_input)
        for k in keys:
            piece[k] = model_input[k].index_select(1, idx)
        if "object_mask" in piece:
            piece["object_mask"] = model_input["object_mask"].index_select(1, idx)
        chunks.append(piece)
    return chunks


def merge_output(res, total_pixels, batch_size):
    """Concatenate chunked outputs back to [batch*total_pixels, ...] (general.py:40-52)."""
    merged = {}
    for key, first in res[0].items():
        if first is None or not torch.is_tensor(first):
            continue
        if first.dim() == 1:
            merged[key] = torch.cat([r[key].reshape(batch_size, -1, 1) for r in res], 1).reshape(batch_size * total_pixels)
        else:
            merged[key] = torch.cat([r[key].reshape(batch_size, -1, r[key].shape[-1]) for r in res], 1).reshape(
                batch_size * total_pixels, -1)
    return merged
