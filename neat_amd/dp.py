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
    force = os.environ.get("NEAT_FORCE_DIST") == "1"      # world size 1 with a real process group: exercises RCCL on a single GPU
    if (world > 1 or force) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1 and "MASTER_PORT" not in os.environ:
            # forced single-rank group: nothing else needs to find it, so any free port will do -- a fixed one collides when two such
            # processes share a host (bench.py's RCCL world-1 leg next to the pytest RCCL check)
            import socket
            with socket.socket() as s:
                s.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(s.getsockname()[1])
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        backend = backend or os.environ.get("NEAT_DIST_BACKEND") or None      # gloo on GPUs: functional checks with ranks sharing a device
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local % max(torch.cuda.device_count(), 1))
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
    """Averages the gradients of `params` across ranks with ONE all-reduce of one flat fp32 buffer per step.

    What every rank does is decided by the parameter list alone (same model on every rank), never by properties of this
    rank's memory -- an earlier version reduced contiguous gradient groups in place when their addresses allowed it and
    negotiated that plan whenever THIS rank's addresses changed; a rank-local trigger lets the ranks issue different
    collectives (count, dtype, size) and hang or corrupt RCCL.  Now: pack (missing gradients as zeros) -> all_reduce(SUM)
    -> scale -> the parameters' .grad become views of the flat buffer (no copy back).  A caller that already owns a flat
    gradient buffer passes it as `flat_grad` and skips the packing."""

    def __init__(self, params, group=None):
        self.params = [p for p in params if p.requires_grad]
        self.group = group
        # world size 1 normally skips the exchange; NEAT_FORCE_DIST=1 (or this attribute) sends the flat buffer through the
        # initialised backend anyway: the one-GPU test of the RCCL path (library load, stream ordering against the step's graph)
        self.force_collective = os.environ.get("NEAT_FORCE_DIST") == "1"
        self.numel = sum(p.numel() for p in self.params)
        self.flat = None
        self._views = None

    def _ensure(self):
        p0 = self.params[0]
        if self.flat is None or self.flat.device != p0.device:
            self.flat = torch.zeros(self.numel, device=p0.device, dtype=torch.float32)
            self._views, off = [], 0
            for p in self.params:
                self._views.append(self.flat[off:off + p.numel()].view(p.shape))
                off += p.numel()

    def active(self):
        """True when a step has to exchange gradients (a process group with more than one rank, or the forced single-rank group)."""
        return dist.is_initialized() and (dist.get_world_size(self.group) > 1 or self.force_collective)

    def pack(self):
        """The gradients into the flat buffer, one multi-tensor launch.  Capturable: Trainer.capture() records it at the end of the step's
        HIP graph, so that a replayed step goes from the graph straight into the collective (`reduce_packed`) with no host work between."""
        self._ensure()
        have = [p.grad is not None and p.grad is not v for p, v in zip(self.params, self._views)]
        for p, v in zip(self.params, self._views):
            if p.grad is None:
                v.zero_()               # (rare: a parameter without a gradient on this rank this step)
        src = [p.grad for p, h in zip(self.params, have) if h]
        if src:
            torch._foreach_copy_([v for v, h in zip(self._views, have) if h], src)

    def reduce_packed(self, repoint=True):
        """ONE all-reduce (mean) of the packed flat buffer on the current stream; afterwards the parameters' .grad are views of it."""
        world = dist.get_world_size(self.group)
        if dist.get_backend(self.group) == "nccl":         # RCCL averages in the collective: one launch less than sum + scale
            dist.all_reduce(self.flat, op=dist.ReduceOp.AVG, group=self.group)
        else:
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.group)
            self.flat.mul_(1.0 / world)
        if repoint:
            for p, v in zip(self.params, self._views):
                if p.grad is not v:
                    p.grad = v

    def all_reduce_mean(self, flat_grad=None):
        """grad_i <- mean over ranks of grad_i (a parameter with no gradient on this rank contributes zeros).
        `flat_grad`: the gradients already live in this flat buffer -> reduced in place."""
        if not dist.is_initialized() or (dist.get_world_size(self.group) == 1 and not self.force_collective):
            return
        world = dist.get_world_size(self.group)
        if flat_grad is not None:
            dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM, group=self.group)
            flat_grad.mul_(1.0 / world)
            return
        self.pack()
        self.reduce_packed()


def all_reduce_scalars(values, group=None):
    """Mean of a dict of 0-d tensors over ranks (logging only)."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return values
    keys = sorted(values)
    buf = torch.stack([values[k].detach().float().reshape(()) for k in keys])
    dist.all_reduce(buf, group=group)
    buf /= dist.get_world_size(group)
    return {k: buf[i] for i, k in enumerate(keys)}
