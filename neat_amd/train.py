"""One training step of the hot path as the reference trainer runs it (code/training/volsdf_train.py:361-374,408):
forward -> loss -> zero_grad/backward -> [gradient all-reduce] -> Adam step -> per-iteration ExponentialLR."""
import torch

from . import networks, synth
from .dp import FlatGradBucket
from .loss import VolSDFLoss
from .wireframe import WireframeGraph


class Trainer:
    def __init__(self, model_conf=None, loss_conf=None, lr=5.0e-4, decay_steps=200000, device="cuda:0", state_dict=None):
        self.device = torch.device(device)
        self.model = networks.VolSDFNetwork(model_conf or synth.ABC_NEAT_A_MODEL_CONF)
        if state_dict is not None:
            self.model.load_state_dict(state_dict)
        self.model.to(self.device).train()
        self.loss = VolSDFLoss(**(loss_conf or synth.ABC_NEAT_A_LOSS_CONF))
        if self.device.type == "cuda":
            from .optim import FlatAdam
            self.optimizer = FlatAdam(self.model.parameters(), lr=lr)      # torch.optim.Adam semantics, one HIP launch
        else:
            self.optimizer = torch.optim.Adam(self.model.parameters(), lr=lr)
        self.scheduler = torch.optim.lr_scheduler.ExponentialLR(self.optimizer, 0.1 ** (1.0 / decay_steps))
        self.bucket = FlatGradBucket(self.model.parameters())
        self._graph, self._captured, self._ring_pos, self.capture_error = None, None, 0, None

    def step(self, model_input, ground_truth):
        if self._graph is not None:
            return self._replay(model_input, ground_truth)
        return self.step_eager(model_input, ground_truth)

    def step_eager(self, model_input, ground_truth):
        out = self.model(model_input)
        losses = self.loss(out, ground_truth)
        self.optimizer.zero_grad(set_to_none=True)
        losses["loss"].backward()
        self.bucket.all_reduce_mean()
        self.optimizer.step()
        self.scheduler.step()
        return out, losses

    # ---- HIP-graph mode: forward + loss + backward are captured once and replayed; the gradient all-reduce, the Adam
    # launch and the scheduler stay outside (their arguments change every step).  Valid while the batch keeps its shapes
    # and its wireframe (one graph per view in a real run); the step must be free of host synchronisation, which the
    # hot path is when the depth samples are given or come from a sync-free sampler.
    def capture(self, model_input, ground_truth, warmup=2):
        """Returns True if the step is now replayed from a HIP graph, False if capture was not possible (stays eager)."""
        if self.device.type != "cuda" or self._graph is not None:
            return self._graph is not None
        tensor_keys = [k for k, v in model_input.items() if isinstance(v, torch.Tensor)]
        self._static_in = dict(model_input)
        self._static_gt = dict(ground_truth)
        for k in tensor_keys:
            self._static_in[k] = model_input[k].clone()
        for k, v in ground_truth.items():
            if isinstance(v, torch.Tensor):
                self._static_gt[k] = v.to(self.device).clone()
        self.model.static_randoms = {}
        self.loss.nan_check = "off"
        sampler = getattr(self.model, "ray_sampler", None)
        was_sync_free = getattr(sampler, "sync_free", None)
        if was_sync_free is not None:
            sampler.sync_free = True                        # Algorithm 1 with its control flow on the device: no .item() per round
        try:
            if hasattr(torch.autograd.graph, "set_warn_on_accumulate_grad_stream_mismatch"):
                torch.autograd.graph.set_warn_on_accumulate_grad_stream_mismatch(False)   # warm-up runs on a side stream on purpose
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):                   # warm-up on the side stream: lazy inits, allocator, draw sites
                for _ in range(warmup):
                    self._refill_randoms()
                    self._fwd_bwd()
                    self._finish_step()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            self._refill_randoms()
            self.optimizer.zero_grad(set_to_none=True)
            graph = torch.cuda.CUDAGraph()
            # thread_local: a NCCL/RCCL watchdog thread may touch the runtime while this thread captures
            with torch.cuda.graph(graph, capture_error_mode="thread_local"):
                self._captured = self._fwd_bwd(zero=False)
            self._graph = graph
            self._static_grads = [(p, p.grad) for p in self.model.parameters()]
            self._finish_step()                             # the capture pass itself does not execute: replay it once
            return True
        except Exception as exc:                            # a sync inside the step, an uncapturable op ...
            self.model.static_randoms = None
            self.loss.nan_check = "deferred"
            if was_sync_free is not None:
                sampler.sync_free = was_sync_free
            self._graph = None
            self.capture_error = exc
            torch.cuda.synchronize()
            return False

    def _fwd_bwd(self, zero=True):
        out = self.model(self._static_in)
        losses = self.loss(out, self._static_gt)
        if zero:
            self.optimizer.zero_grad(set_to_none=True)
        losses["loss"].backward()
        return out, losses

    def _finish_step(self):
        if self._graph is not None:
            for p, g in self._static_grads:                 # an eager step in between re-pointed .grad: the graph writes the captured tensors
                if p.grad is not g:
                    p.grad = g
            self._graph.replay()
        self.bucket.all_reduce_mean()
        self.optimizer.step()
        self.scheduler.step()

    def _refill_randoms(self):
        """Fresh CPU draws, in the forward's draw order, into the persistent device tensors (pinned staging ring)."""
        slots = sorted(self.model.static_randoms.values(), key=lambda s: s["order"])
        for slot in slots:
            ring = slot.setdefault("ring", [])
            if len(ring) < 4:
                ring.append((torch.empty_like(slot["dev"], device="cpu").pin_memory(), torch.cuda.Event()))
            pinned, ev = ring[self._ring_pos % len(ring)] if len(ring) == 4 else ring[-1]
            ev.synchronize()                                # the copy that last used this staging buffer is long done
            pinned.copy_(slot["draw"]())
            slot["dev"].copy_(pinned, non_blocking=True)
            ev.record()
        self._ring_pos += 1

    def _same_batch_layout(self, model_input, ground_truth):
        """The captured graph is valid for batches with the captured tensor shapes and the captured non-tensor entries
        (the wireframe of the view: its vertex count shapes the matching)."""
        for static, fresh in ((self._static_in, model_input), (self._static_gt, ground_truth)):
            if set(static) != set(fresh):
                return False
            for k, v in fresh.items():
                if isinstance(v, torch.Tensor):
                    if v.shape != static[k].shape or v.dtype != static[k].dtype:
                        return False
                elif isinstance(v, (list, tuple)):
                    if len(v) != len(static[k]) or any(a is not b for a, b in zip(v, static[k])):
                        return False
                elif v is not static[k] and v != static[k]:
                    return False
        return True

    def check_nan(self):
        """Graph mode keeps the line-loss NaN flag on the device (loss.nan_check == "off"); this reads it (one sync)."""
        flag = self.loss.nan_flag
        if flag is not None and bool(flag.item()):
            raise FloatingPointError("line loss is NaN (the reference drops into pdb here, loss_wfr.py:66-67)")

    def _replay(self, model_input, ground_truth):
        if not self._same_batch_layout(model_input, ground_truth):
            return self.step_eager(model_input, ground_truth)      # another view / batch size: this graph does not apply
        # every tensor of the fresh batch is copied into the captured tensors (a few KB per step).  No "unchanged?" shortcut on
        # (data_ptr, _version): a new batch uploaded with .to(device) usually lands on the block the allocator just freed, with
        # version 0, and would be mistaken for the previous one.
        for static, fresh in ((self._static_in, model_input), (self._static_gt, ground_truth)):
            for k, v in fresh.items():
                if isinstance(v, torch.Tensor) and v is not static[k]:
                    static[k].copy_(v, non_blocking=True)
        self._refill_randoms()
        self._finish_step()
        return self._captured


def synthetic_batch(seed, n_rays, device, view=0):
    """Model input + ground truth in the trainer's layout from neat_amd.synth (no dataset, no network access)."""
    sc = synth.synth_scene(seed=seed, n_rays=n_rays, view=view)
    wf = WireframeGraph(torch.tensor(sc["wf_vertices"]), torch.tensor(sc["wf_vconf"]), torch.tensor(sc["wf_edges"]),
                        torch.tensor(sc["wf_weights"]), sc["res"], sc["res"])
    inp = {k: torch.tensor(sc[k]).to(device) for k in ("intrinsics", "pose", "uv", "uv_proj")}
    inp["wireframe"] = [wf]
    gt = {"rgb": torch.tensor(sc["gt_rgb"]).to(device), "lines2d": torch.tensor(sc["gt_lines2d"]).to(device)}
    return sc, inp, gt
