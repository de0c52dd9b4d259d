"""Ray/view data parallelism: one process per GPU, one view per rank, ONE flat-bucket gradient all-reduce per step.

The reference has no distributed code.  Rays are independent and each rank renders its own view,
so the only exchange is the gradient mean over ranks: 1 219 274 floats = 4.9 MB per step, sent as a single
RCCL all-reduce over xGMI (torch.distributed backend "nccl" is RCCL on ROCm; "gloo" on CPU for the tests).
With world_size == 1 nothing here touches the data path.
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialise torch.distributed from RANK/WORLD_SIZE/MASTER_* (torchrun) if WORLD_SIZE > 1. Returns (rank, world, local_rank)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def rank_seed(base_seed, rank):
    """Rank r draws its own view and rays with seed base+r."""
    return base_seed + rank


def shard_rays(total_rays, world):
    """C4: a global batch of `total_rays` rays is split evenly; every rank renders total/world rays of its own view."""
    if total_rays % world != 0:
        raise ValueError(f"{total_rays} rays do not divide over {world} ranks")
    return total_rays // world


class FlatGradBucket:
    """Averages the gradients of `params` across ranks with a single all-reduce of one flat fp32 buffer."""

    def __init__(self, params, group=None):
        self.params = [p for p in params if p.requires_grad]
        self.group = group
        self.numel = sum(p.numel() for p in self.params)
        self.flat = None
        self._plan_key, self._plan = None, []

    def _ensure(self):
        p0 = self.params[0]
        if self.flat is None or self.flat.device != p0.device:
            self.flat = torch.zeros(self.numel, device=p0.device, dtype=torch.float32)

    def all_reduce_mean(self, flat_grad=None):
        """grad_i <- mean over ranks of grad_i (a parameter with no gradient on this rank contributes zeros).
        `flat_grad`: the gradients already live in this flat buffer (neat_amd.optim.FlatAdam) -> reduced in place."""
        if not dist.is_initialized() or dist.get_world_size(self.group) == 1:
            return
        world = dist.get_world_size(self.group)
        if flat_grad is not None:
            dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM, group=self.group)
            flat_grad.mul_(1.0 / world)
            return
        # gradients that already tile one contiguous buffer (the 57 weight-normed tensors come back from the HIP backward
        # as views of one flat allocation) are reduced in place; only the rest goes through the packing copies.  The plan
        # depends on this rank's memory layout, so the ranks agree on it (one tiny all-reduce) whenever it changes.
        key = tuple(0 if p.grad is None else p.grad.data_ptr() for p in self.params)
        if key != self._plan_key:
            groups = []
            by_storage = {}
            for i, p in enumerate(self.params):
                g = p.grad
                if g is not None and g.is_contiguous() and g.dtype == torch.float32:
                    by_storage.setdefault(g.untyped_storage().data_ptr(), []).append(i)
            for idx in by_storage.values():
                grads = [self.params[i].grad for i in idx]
                lo = min(g.data_ptr() for g in grads)
                hi = max(g.data_ptr() + 4 * g.numel() for g in grads)
                if len(idx) > 1 and sum(4 * g.numel() for g in grads) == hi - lo:
                    groups.append(tuple(sorted(idx)))
            groups.sort()
            sig = hash(tuple(groups)) % (1 << 40)
            probe = torch.tensor([sig, -sig], dtype=torch.int64, device=self.params[0].grad.device if self.params[0].grad is not None
                                 else self.params[0].device)
            dist.all_reduce(probe, op=dist.ReduceOp.MIN, group=self.group)
            agreed = int(probe[0]) == sig and int(probe[1]) == -sig        # min(sig) == sig == max(sig) on every rank
            agree_all = torch.tensor([1 if agreed else 0], dtype=torch.int64, device=probe.device)
            dist.all_reduce(agree_all, op=dist.ReduceOp.MIN, group=self.group)
            self._plan = groups if int(agree_all[0]) == 1 else []
            self._plan_key = key
        in_place = set()
        for idx in self._plan:
            grads = [self.params[i].grad for i in idx]
            first = min(grads, key=lambda g: g.data_ptr())
            n = sum(g.numel() for g in grads)
            flat = torch.empty(0, dtype=torch.float32, device=first.device).set_(first.untyped_storage(), first.storage_offset(), (n,))
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
            flat.mul_(1.0 / world)
            in_place.update(idx)
        rest = [p for i, p in enumerate(self.params) if i not in in_place]
        if not rest:
            return
        numel = sum(p.numel() for p in rest)
        p0 = rest[0]
        if self.flat is None or self.flat.device != p0.device or self.flat.numel() < numel:
            self.flat = torch.zeros(numel, device=p0.device, dtype=torch.float32)
        flat = self.flat[:numel]
        off = 0
        views = []
        for p in rest:
            views.append(flat[off:off + p.numel()].view(p.shape))
            off += p.numel()
        have = [p.grad is not None for p in rest]
        flat.zero_()
        if any(have):
            torch._foreach_copy_([v for v, h in zip(views, have) if h], [p.grad for p, h in zip(rest, have) if h])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
        flat.mul_(1.0 / world)
        for p, v in zip(rest, views):
            if p.grad is None:
                p.grad = v.clone()
            else:
                p.grad.copy_(v)


def all_reduce_scalars(values, group=None):
    """Mean of a dict of 0-d tensors over ranks (logging only)."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return values
    keys = sorted(values)
    buf = torch.stack([values[k].detach().float().reshape(()) for k in keys])
    dist.all_reduce(buf, group=group)
    buf /= dist.get_world_size(group)
    return {k: buf[i] for i, k in enumerate(keys)}
