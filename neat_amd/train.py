"""One training step of the hot path as the reference trainer runs it (code/training/volsdf_train.py:361-374,408):
forward -> loss -> zero_grad/backward -> [gradient all-reduce] -> Adam step -> per-iteration ExponentialLR."""
import os

import torch
import torch.distributed as dist

from . import networks, ops, synth
from .dp import FlatGradBucket
from .loss import VolSDFLoss
from .wireframe import WireframeGraph


GT_KEYS = ("rgb", "lines2d")      # what VolSDFLoss reads from the ground truth (loss_wfr.py:92-110)


def _backward(loss):
    """loss.backward() seeded with a persistent 1.0 on the device: no `ones_like` launch per step, and the loss's own autograd node
    (ops.LossTailFn) is told through ops.unit_seed that the seed is the constant 1 and multiplies nothing."""
    if loss.is_cuda:
        from . import ops
        with ops.unit_seed(loss.device) as one:
            loss.backward(gradient=one)
    else:
        loss.backward()


class Trainer:
    def __init__(self, model_conf=None, loss_conf=None, lr=5.0e-4, decay_steps=200000, device="cuda:0", state_dict=None, parts=None):
        self.device = torch.device(device)
        if self.device.type == "cuda":
            from . import cap_host_threads
            cap_host_threads()             # the host thread pool follows the container's CPU quota (neat_amd/__init__.py)
        if parts is not None:              # an already built model / loss / optimizer / scheduler / bucket (neat_amd.runner)
            self.model, self.loss, self.optimizer, self.scheduler, self.bucket = parts
        else:
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
        self._graphs, self._pool, self._ring_pos, self.capture_error = {}, None, 0, None
        self.auto_capture = 0             # > 0: a batch layout seen this many times eagerly is captured on its next visit
        self._visits, self._uncapturable, self._last = {}, set(), None
        self._capture_fault = None
        self.replays = self.eager_steps = 0
        # round 6: what a captured step holds besides forward + loss + backward: the Adam launch (graph_tail) and, in a data-parallel
        # run, the gradient all-reduce in front of it (graph_collective; off: all-reduce and Adam stay eager behind the replay)
        # graph_collective defaults to the case that could be verified on hardware: the one-rank RCCL group (NEAT_FORCE_DIST=1; same
        # trajectory as without a group, tests/test_dp_gpu.py).  With more than one rank the collective stays eager behind the replay unless
        # NEAT_GRAPH_COLLECTIVE=1 asks for it: no multi-GPU box was available to try a captured multi-rank all-reduce, and a hang cannot
        # be caught the way a refused capture can.
        self.graph_tail = os.environ.get("NEAT_GRAPH_TAIL", "1") != "0"
        gc = os.environ.get("NEAT_GRAPH_COLLECTIVE")
        self.graph_collective = None if gc is None else gc == "1"      # None: decided when a step is captured (_collective_in_graph)

    def step(self, model_input, ground_truth):
        if self._graphs or self.auto_capture:
            key = self._layout_key(model_input, ground_truth)
            entry = self._graphs.get(key)
            if entry is not None:
                return self._replay(entry, model_input, ground_truth)
            if self.auto_capture and self.device.type == "cuda":
                n = self._visits[key] = self._visits.get(key, 0) + 1
                if n > self.auto_capture and self.capture(model_input, ground_truth, warmup=0):
                    return self._graphs[key].outputs          # (capture() ends with one replayed step)
        self.eager_steps += 1
        self._last = None
        return self.step_eager(model_input, ground_truth)

    def step_eager(self, model_input, ground_truth):
        out = self.model(model_input)
        losses = self.loss(out, ground_truth)
        self.optimizer.zero_grad(set_to_none=True)
        _backward(losses["loss"])
        self.bucket.all_reduce_mean()
        self.optimizer.step()
        self.scheduler.step()
        return _detached(out), _detached(losses)

    # ---- HIP-graph mode: forward + loss + backward are captured once PER BATCH LAYOUT and replayed; the gradient all-reduce, the
    # Adam launch and the scheduler stay outside (their arguments change every step).  A layout = the tensor shapes of the batch, its
    # wireframe object (the vertex count shapes the matching and its device tensors are baked into the graph) and the model switches
    # that change what the forward launches.  A real run walks over the views of a scene (volsdf_train.py:361 draws a view per
    # iteration): every view gets its own graph (auto_capture), all graphs share one memory pool -- they never run concurrently, so the
    # 6.6 GB of per-step workspace exists once, not once per view.  The step must be free of host synchronisation: it is when the depth
    # samples are given, or come from the sync-free sampler (ray_sampler.ErrorBoundSampler.get_z_vals_device).
    @property
    def _graph(self):
        """Any captured graph (kept for callers that only ask "is the step replayed?")."""
        return next(iter(self._graphs.values())).graph if self._graphs else None

    def _layout_key(self, model_input, ground_truth):
        key = []
        for tag, batch in (("in", model_input), ("gt", ground_truth)):
            for k in sorted(batch):
                v = batch[k]
                if isinstance(v, torch.Tensor):
                    key.append((tag, k, tuple(v.shape), str(v.dtype)))
                elif isinstance(v, (list, tuple)):
                    key.append((tag, k, tuple(id(x) for x in v)))
                else:
                    key.append((tag, k, v if isinstance(v, (int, float, str, bool, type(None))) else id(v)))
        zo = getattr(self.model, "z_vals_override", None)
        key.append(("z_vals_override", None if zo is None else tuple(zo.shape)))
        key.append(("training", self.model.training))
        return tuple(key)

    def capture(self, model_input, ground_truth, warmup=2):
        """Capture the step for this batch layout.  Returns True if it is now replayed from a HIP graph, False if capture was not
        possible (the layout stays eager; `capture_error` holds the reason).  `warmup` eager optimizer steps on this batch run first
        (lazy initialisations; 0 when the layout has already been stepped eagerly).
        With warmup > 0 the call takes EXACTLY warmup + 1 optimizer steps (= gradient all-reduces) whether the capture succeeds
        (warm-ups + the first replayed step) or fails anywhere on the way (the missing steps are run eagerly): ranks of a data-parallel
        run whose captures end differently stay in lock step, same number of collectives and the same Adam step count."""
        if self.device.type != "cuda":
            return False
        key = self._layout_key(model_input, ground_truth)
        if key in self._graphs:
            return True
        if key in self._uncapturable:
            return False
        entry = _Captured()
        entry.static_in, entry.static_gt = dict(model_input), dict(ground_truth)
        for k, v in model_input.items():      # (host-side entries of a dataset sample -- masks, labels ... -- are not read by the forward)
            if isinstance(v, torch.Tensor) and v.device == self.device:
                entry.static_in[k] = v.clone()
        for k, v in ground_truth.items():
            if isinstance(v, torch.Tensor) and k in GT_KEYS:
                entry.static_gt[k] = v.to(self.device).clone()
        # given depth samples are one more static input (the graph reads THIS tensor; a caller may assign another one later)
        zo = getattr(self.model, "z_vals_override", None)
        entry.static_z = None if zo is None else zo.clone()
        entry.randoms = {}
        # what the captured forward needs from the model, for the duration of the capture only -- eager steps of other layouts keep
        # drawing fresh randoms, checking NaNs their way and using the sampler they were configured with
        sampler = getattr(self.model, "ray_sampler", None)
        eager = (self.loss.nan_check, getattr(sampler, "sync_free", None))
        self.model.static_randoms = entry.randoms          # persistent device tensor per draw site, refilled before every replay
        self.loss.nan_check = "off"                         # the NaN flag stays on the device: check_nan()
        if eager[1] is not None:
            sampler.sync_free = True                        # Algorithm 1 with its control flow on the device: no .item() per round
        if entry.static_z is not None:
            self.model.z_vals_override = entry.static_z
        ok = False
        finished = 0                                        # gradient all-reduces issued so far by this call: what the other ranks
                                                            # count on -- an exception AFTER the exchange of a warm-up step (say, in
                                                            # optimizer.step) must not make this rank issue one collective more

        def count_collective():
            nonlocal finished
            finished += 1
        torch.cuda.synchronize()                            # (captures are rare: start from an idle device)
        try:
            if hasattr(torch.autograd.graph, "set_warn_on_accumulate_grad_stream_mismatch"):
                torch.autograd.graph.set_warn_on_accumulate_grad_stream_mismatch(False)   # warm-up runs on a side stream on purpose
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):                   # warm-up on the side stream: lazy inits, allocator, draw sites
                for _ in range(warmup):                     # (real optimizer steps on this batch)
                    self._refill_randoms(entry)
                    self._fwd_bwd(entry)
                    self._finish_step(None, on_collective=count_collective)
                if not entry.randoms:                       # no warm-up step ran (auto-capture of a layout that already ran eagerly):
                    rng = torch.get_rng_state()             # one forward registers the draw sites, without consuming the CPU stream
                    with torch.no_grad():
                        self.model(entry.static_in)
                    torch.set_rng_state(rng)
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            self._refill_randoms(entry)
            self.optimizer.zero_grad(set_to_none=True)
            # whatever the model caches per parameter version (the packed weights, ops.NetHandle.packed) must be rebuilt INSIDE the
            # graph: a cache filled by the forward just above would be read, never refreshed, by every replay
            torch._C._increment_version(list(self.model.parameters()))
            if self._capture_fault is not None:              # (tests: a capture that fails on this rank only)
                self._capture_fault()
            if self._pool is None:
                self._pool = torch.cuda.graph_pool_handle()
            # process-global device tensors the captured step reads must exist (and be filled) BEFORE the capture: with warmup = 0 no
            # eager backward has run yet, and a tensor first created inside the capture has its fill recorded, not executed, in storage
            # of the graph's private pool -- after an aborted capture it would stay cached with garbage in it
            ops.grad_one(self.device)
            if self.bucket.active():
                self.bucket._ensure()
            graph = torch.cuda.CUDAGraph()
            # thread_local: a NCCL/RCCL watchdog thread may touch the runtime while this thread captures
            with torch.cuda.graph(graph, pool=self._pool, capture_error_mode="thread_local"):
                out, losses = self._fwd_bwd(entry, zero=False)
                entry.outputs = (_detached(out), _detached(losses))
                del out, losses
                # data parallel: the gradients go into the flat bucket INSIDE the graph -- a replayed step then runs graph ->
                # all-reduce -> Adam (reading the flat buffer) with no host work between the replay and the collective
                entry.packed = self.bucket.active() and self.device.type == "cuda" and hasattr(self.optimizer, "flat_param")
                if entry.packed:
                    self.bucket.pack()
                # round 6: the tail of the step inside the graph too -- [the gradient all-reduce on RCCL, when `graph_collective`] and
                # the Adam launch (its step-dependent numbers live in device memory, refreshed with the step's other host-made
                # inputs): a replayed step is then ONE graph launch, nothing is enqueued behind it
                # (gloo's device all-reduce synchronises the stream: not capturable, the tail then stays eager as in round 5)
                coll_ok = self._collective_in_graph()
                tail = self.graph_tail and hasattr(self.optimizer, "capture_step") and (not entry.packed or coll_ok)
                if tail:
                    if entry.packed:
                        self.bucket.reduce_packed(repoint=False)
                    entry.adam_has = self.optimizer.capture_step(flat_grad=self.bucket.flat if entry.packed else None)
            entry.graph = graph
            if entry.adam_has is not None:      # one more host-made input of every replay, staged like the CPU draws
                opt, has = self.optimizer, entry.adam_has
                entry.randoms["adam_coef"] = {"draw": lambda: opt.next_coef(has), "dev": opt.coef, "order": 1 << 30}
            entry.static_grads = [(p, p.grad) for p in self.model.parameters()]
            entry.nan_flag = self.loss.nan_flag
            ok = True
        except Exception as exc:                            # a sync inside the step, an uncapturable op ...
            self.capture_error = exc
            self._uncapturable.add(key)
            torch.cuda.synchronize()
        finally:
            self.model.static_randoms = None
            self.loss.nan_check = eager[0]
            if eager[1] is not None:
                sampler.sync_free = eager[1]
            if entry.static_z is not None:
                self.model.z_vals_override = zo
        if ok:
            self._graphs[key] = entry
            self._load_batch(entry, model_input, ground_truth)      # (same values; loads the multi-tensor copy kernel outside any timed step)
            if entry.adam_has is not None:                  # this replay's Adam numbers (the draws were refilled before the capture)
                slot = entry.randoms["adam_coef"]
                slot["dev"].copy_(slot["draw"]())
            self._finish_step(entry)                        # the capture pass itself does not execute: replay it once
            self.replays += 1
            self._last = entry
        else:
            self._last = None                               # check_nan() reads the loss's own flag again, not the previous layout's
            self.loss.nan_flag = None                       # ... and that flag must not be a tensor of the aborted capture (never executed)
                                                            # or a stale warm-up value: the eager steps below publish their own
        if not ok and entry.packed and self.graph_tail and self._collective_in_graph():
            # the collective may be what the runtime refused to capture: once more with all-reduce and Adam eager behind the replay
            self.graph_collective = False
            self._uncapturable.discard(key)
            first_error = self.capture_error
            ok = self.capture(model_input, ground_truth, warmup=0)
            if ok:
                self.capture_error = RuntimeError(f"captured without the collective (with it: {first_error!r})")
                return True
        if not ok and warmup > 0:
            self.optimizer.zero_grad(set_to_none=True)      # (whatever a half-finished attempt left in .grad)
            for _ in range(warmup + 1 - finished):          # the steps the successful path would have taken
                self.step_eager(model_input, ground_truth)
                self.eager_steps += 1
        return ok

    def _collective_in_graph(self):
        """May the gradient all-reduce be captured with the step?  Only on RCCL (gloo synchronises the stream); by default only for the
        one-rank group, the case verified on hardware (see __init__)."""
        if not (dist.is_initialized() and dist.get_backend(self.bucket.group) == "nccl"):
            return False
        if self.graph_collective is None:
            return dist.get_world_size(self.bucket.group) == 1
        return bool(self.graph_collective)

    def _fwd_bwd(self, entry, zero=True):
        out = self.model(entry.static_in)
        losses = self.loss(out, entry.static_gt)
        if zero:
            self.optimizer.zero_grad(set_to_none=True)
        _backward(losses["loss"])
        return out, losses

    def _finish_step(self, entry, on_collective=None):
        if entry is not None and entry.adam_has is not None:
            for p, g in entry.static_grads:                 # (kept pointing at this graph's tensors for whoever reads .grad afterwards)
                if p.grad is not g:
                    p.grad = g
            entry.graph.replay()                            # forward + loss + backward [+ pack + all-reduce] + Adam: one launch
            if on_collective is not None and entry.packed:
                on_collective()
            self.optimizer.after_replay()
            self.scheduler.step()
            return
        if entry is not None and entry.packed:
            entry.graph.replay()                            # ... ends with the pack of its own gradient tensors into bucket.flat
            self.bucket.reduce_packed()
            if on_collective is not None:
                on_collective()
            self.optimizer.step(flat_grad=self.bucket.flat)
            self.scheduler.step()
            return
        if entry is not None:
            for p, g in entry.static_grads:                 # another graph or an eager step re-pointed .grad: this graph writes ITS tensors
                if p.grad is not g:
                    p.grad = g
            entry.graph.replay()
        self.bucket.all_reduce_mean()
        if on_collective is not None:
            on_collective()
        self.optimizer.step()
        self.scheduler.step()

    def _refill_randoms(self, entry, collect=None):
        """Fresh CPU draws, in the forward's draw order, into the layout's persistent device tensors (pinned staging ring).
        collect (a list): the (device tensor, pinned buffer) pairs are appended for ONE copy_batch launch by the caller, who records the
        returned events after it; None: one asynchronous copy per draw site, here."""
        slots = sorted(entry.randoms.values(), key=lambda s: s["order"])
        events = []
        for slot in slots:
            ring = slot.get("ring")
            if ring is None:
                # all four staging buffers at once, on the slot's first refill (inside capture()): pinned allocations cost
                # milliseconds the first time a process asks for them -- grown one per replay they landed in the first timed
                # replays of a fresh layout (sampler step: 5.2 instead of 4.1 ms over the first 20 replays)
                ring = slot["ring"] = [(torch.empty_like(slot["dev"], device="cpu").pin_memory(), torch.cuda.Event()) for _ in range(4)]
            pinned, ev = ring[self._ring_pos % 4]
            ev.synchronize()                                # the copy that last used this staging buffer is long done
            pinned.copy_(slot["draw"]())
            if collect is not None and ops.copy_batch_ok(slot["dev"], pinned):
                collect.append((slot["dev"], pinned))
                events.append(ev)
            else:
                slot["dev"].copy_(pinned, non_blocking=True)
                ev.record()
        self._ring_pos += 1
        return events

    def check_nan(self):
        """Graph mode keeps the line-loss NaN flag on the device (loss.nan_check == "off"); this reads it (one sync)."""
        deferred = None
        if self._last is None and hasattr(self.loss, "_check_deferred"):
            try:
                self.loss._check_deferred()                 # eager steps publish through the loss's own deferred flag
            except FloatingPointError as exc:               # ... which raises on THIS rank: held until the ranks have agreed below
                deferred = exc
        flag = self._last.nan_flag if self._last is not None else self.loss.nan_flag      # (read after the deferred check has run)
        # (graph mode keeps the line loss itself as the flag -- testing it on the device every step would be one more launch)
        bad = deferred is not None or (flag is not None and bool((torch.isnan(flag).any() if flag.is_floating_point() else flag.any()).item()))
        # data parallel: the flag belongs to THIS rank's batch.  The ranks agree before anyone raises -- a rank that stopped alone would
        # leave the others blocked in the next gradient all-reduce until the watchdog tears the job down.  (Every rank calls this at
        # the same iterations: the runner's log interval and checkpoint epochs.)
        if dist.is_initialized() and dist.get_world_size() > 1:
            t = torch.tensor([1.0 if bad else 0.0], device=self.device if dist.get_backend() == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            bad = bool(t.item() > 0.0)
        if bad:
            raise FloatingPointError("line loss is NaN on at least one rank (the reference drops into pdb here, loss_wfr.py:66-67)") from deferred

    def _load_batch(self, entry, model_input, ground_truth, collect=None):
        """Every tensor of the fresh batch is copied into the captured tensors, in ONE multi-tensor launch (a few KB per step).  No
        "unchanged?" shortcut on (data_ptr, _version): a new batch uploaded with .to(device) usually lands on the block the allocator
        just freed, with version 0, and would be mistaken for the previous one."""
        dst, src = [], []
        for static, fresh, keys in ((entry.static_in, model_input, None), (entry.static_gt, ground_truth, GT_KEYS)):
            for k, v in fresh.items():
                if not isinstance(v, torch.Tensor) or v is static[k] or static[k].device != self.device or (keys and k not in keys):
                    continue
                if v.device == self.device and v.dtype == static[k].dtype:
                    dst.append(static[k]); src.append(v)
                else:
                    static[k].copy_(v, non_blocking=True)
        zo = getattr(self.model, "z_vals_override", None)
        if entry.static_z is not None and zo is not entry.static_z:
            dst.append(entry.static_z); src.append(zo)
        if collect is not None:
            for d, s_ in zip(dst, src):
                if ops.copy_batch_ok(d, s_):
                    collect.append((d, s_))
                else:
                    d.copy_(s_, non_blocking=True)
        elif dst:
            torch._foreach_copy_(dst, src)

    def _replay(self, entry, model_input, ground_truth):
        # the step's prefix as ONE launch: fresh batch tensors and the CPU-drawn randoms (read from their pinned buffers by the kernel)
        pairs = []
        self._load_batch(entry, model_input, ground_truth, collect=pairs)
        events = self._refill_randoms(entry, collect=pairs)
        if pairs:
            ops.copy_batch(pairs)
        for ev in events:
            ev.record()
        self._finish_step(entry)
        self._last = entry
        self.replays += 1
        return entry.outputs


def _detached(d):
    """Outputs / losses without autograd history.  The step has already run backward; a caller that kept the previous step's
    graph alive through them would also keep its AccumulateGrad nodes (bound to the stream of that step) alive, and a later
    capture on the capture stream then records a wait on that other stream: hipStreamEndCapture dies on the unjoined fork."""
    if hasattr(d, "detached"):
        return d.detached()
    return type(d)((k, v.detach() if isinstance(v, torch.Tensor) else v) for k, v in d.items())


class _Captured:
    """One captured batch layout: the graph, its static input tensors, its outputs and the gradient tensors it writes."""
    graph = static_in = static_gt = static_z = outputs = static_grads = randoms = nan_flag = None
    packed = False          # the graph ends with the pack of its gradients into the data-parallel bucket (dp.FlatGradBucket.pack)
    adam_has = None         # round 6: the graph also holds [the all-reduce and] the Adam launch; which parameters that launch steps


def synthetic_batch(seed, n_rays, device, view=0):
    """Model input + ground truth in the trainer's layout from neat_amd.synth (no dataset, no network access)."""
    sc = synth.synth_scene(seed=seed, n_rays=n_rays, view=view)
    wf = WireframeGraph(torch.tensor(sc["wf_vertices"]), torch.tensor(sc["wf_vconf"]), torch.tensor(sc["wf_edges"]),
                        torch.tensor(sc["wf_weights"]), sc["res"], sc["res"])
    inp = {k: torch.tensor(sc[k]).to(device) for k in ("intrinsics", "pose", "uv", "uv_proj")}
    inp["wireframe"] = [wf]
    gt = {"rgb": torch.tensor(sc["gt_rgb"]).to(device), "lines2d": torch.tensor(sc["gt_lines2d"]).to(device)}
    return sc, inp, gt
