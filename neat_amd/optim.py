"""Adam over one flat parameter buffer.  Same arithmetic, hyper-parameters and `state_dict()` layout as
`torch.optim.Adam` (the reference trainer's optimizer, training/volsdf_train.py:177), but the parameters and both
moments live in three flat fp32 buffers and a step is ONE HIP launch (`neat_adam_step`) instead of multi-tensor kernels
over 65 tensors.  Gradients stay wherever autograd put them: the kernel reads them through a pointer table passed as a
kernel argument (so `zero_grad(set_to_none=True)` keeps its meaning and costs nothing)."""
import ctypes

import torch

from . import _lib

MAX_SEGMENTS = 96


class FlatAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        params = [p for p in params if p.requires_grad]
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=0, amsgrad=False, maximize=False))
        if len(self.param_groups) != 1:
            raise ValueError("FlatAdam keeps one flat buffer: pass one parameter group")
        self._params = self.param_groups[0]["params"]
        dev = self._params[0].device
        if dev.type != "cuda" or any(p.dtype != torch.float32 or p.device != dev for p in self._params):
            raise RuntimeError("FlatAdam needs float32 CUDA parameters on one device (no CPU path)")
        if len(self._params) > MAX_SEGMENTS:
            raise ValueError(f"FlatAdam handles up to {MAX_SEGMENTS} parameter tensors")
        sizes = [p.numel() for p in self._params]
        self.numel = sum(sizes)
        self._offsets = (ctypes.c_longlong * (len(sizes) + 1))()
        for i, n in enumerate(sizes):
            self._offsets[i + 1] = self._offsets[i] + n
        self.flat_param = torch.empty(self.numel, device=dev)
        self.exp_avg = torch.zeros(self.numel, device=dev)
        self.exp_avg_sq = torch.zeros(self.numel, device=dev)
        self._steps = (ctypes.c_int * len(sizes))()
        # step-dependent numbers of a CAPTURED launch (capture_step): (lr / bias_correction1, 1 / sqrt(bias_correction2)) per tensor, in
        # device memory; refreshed by the trainer before every replay with next_coef()
        self.coef = torch.zeros(2 * len(sizes), device=dev)
        for i, p in enumerate(self._params):
            off, n = self._offsets[i], sizes[i]
            view = self.flat_param[off:off + n].view(p.shape)
            view.copy_(p.data)
            p.data = view                      # the module's parameters now alias the flat buffer
            self.state[p] = {"step": torch.tensor(0.0), "exp_avg": self.exp_avg[off:off + n].view(p.shape),
                             "exp_avg_sq": self.exp_avg_sq[off:off + n].view(p.shape)}

    @torch.no_grad()
    def step(self, closure=None, flat_grad=None):
        """flat_grad: ALL gradients live in this flat fp32 buffer, laid out like the parameters (dp.FlatGradBucket.flat after the
        all-reduce): the pointer table is built once per buffer, no per-parameter host work in the step.
        Semantics of this (data-parallel) form = DistributedDataParallel's: EVERY parameter steps, a parameter that had no gradient on
        any rank steps with the zeros the bucket packed for it (its moments decay).  That is deliberate -- which parameters have a
        gradient is a property of one rank's batch (e.g. no matched junction on this view), and ranks that skipped different
        parameters would drift apart; the pointer-table form below keeps torch.optim.Adam's single-process rule (no gradient: no
        step, no step count)."""
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        group = self.param_groups[0]
        self._realias()
        keep = []                                  # non-contiguous / non-fp32 gradients are normalised first (not the usual case)
        if flat_grad is not None:
            if flat_grad.numel() != self.numel or flat_grad.dtype != torch.float32 or flat_grad.device != self.flat_param.device:
                raise ValueError("FlatAdam.step(flat_grad=...): one fp32 buffer with the parameters' layout on their device")
            cache = getattr(self, "_flat_ptrs", None)
            if cache is None or cache[0] != flat_grad.data_ptr():
                tab = (ctypes.c_void_p * len(self._params))()
                for i in range(len(self._params)):
                    tab[i] = flat_grad.data_ptr() + 4 * self._offsets[i]
                cache = self._flat_ptrs = (flat_grad.data_ptr(), tab)
            ptrs = cache[1]
            for i in range(len(self._params)):
                self._steps[i] += 1
        else:
            ptrs = (ctypes.c_void_p * len(self._params))()
            for i, p in enumerate(self._params):
                g = p.grad
                if g is None:
                    continue
                if g.dtype != torch.float32 or not g.is_contiguous():
                    g = g.float().contiguous()
                    keep.append(g)
                ptrs[i] = g.data_ptr()
                self._steps[i] += 1
        b1, b2 = group["betas"]
        P = lambda t: ctypes.c_void_p(t.data_ptr())
        _lib.check(_lib.lib().neat_adam_step(P(self.flat_param), ptrs, self._offsets, self._steps, len(self._params), P(self.exp_avg),
                                             P(self.exp_avg_sq), float(group["lr"]), float(b1), float(b2), float(group["eps"]),
                                             ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "neat_adam_step")
        # the parameters changed behind autograd's back: bump their version counters (packed-weight caches key on them)
        torch._C._increment_version(self._params)
        self._steps_dirty = True                   # state[p]["step"] is refreshed when somebody reads it (state_dict), not 65 tensors per step
        return loss

    @torch.no_grad()
    def capture_step(self, flat_grad=None):
        """The Adam launch as a node of the step's HIP graph (call INSIDE the capture): gradients through a pointer table -- the
        parameters' .grad tensors as they are now (the graph's own static tensors), or one flat buffer -- and the step-dependent
        numbers from `self.coef` (device).  Returns `has`: which parameters the captured launch updates; the caller refreshes
        `self.coef` with next_coef(has) before every replay.  Nothing is stepped by this call (a capture does not execute)."""
        group = self.param_groups[0]
        self._realias()
        tab = (ctypes.c_void_p * len(self._params))()
        has = []
        for i, p in enumerate(self._params):
            if flat_grad is not None:
                tab[i] = flat_grad.data_ptr() + 4 * self._offsets[i]
                has.append(True)
            else:
                g = p.grad
                ok = g is not None and g.dtype == torch.float32 and g.is_contiguous()
                if g is not None and not ok:
                    raise RuntimeError("FlatAdam.capture_step: contiguous float32 gradients only")
                if ok:
                    tab[i] = g.data_ptr()
                has.append(ok)
        b1, b2 = group["betas"]
        P = lambda t: ctypes.c_void_p(t.data_ptr())
        _lib.check(_lib.lib().neat_adam_step_coef(P(self.flat_param), tab, self._offsets, len(self._params), P(self.exp_avg), P(self.exp_avg_sq),
                                                  P(self.coef), float(b1), float(b2), float(group["eps"]),
                                                  ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "neat_adam_step_coef")
        return has

    def next_coef(self, has):
        """Counts one step for the parameters of `has` and returns the captured launch's numbers for it as a CPU float32 tensor
        [2 n] (the arithmetic of neat_adam_step: double precision, rounded once)."""
        import math
        import numpy as np
        group = self.param_groups[0]
        # (neat_adam_step receives lr and the betas as C floats and widens them: the same float32-rounded values here, so that a
        # replayed step and an eager one update the parameters to the same bits)
        f32 = lambda x: ctypes.c_float(float(x)).value
        b1, b2 = (f32(b) for b in group["betas"])
        lr = f32(group["lr"])
        steps = np.ctypeslib.as_array(self._steps)               # (a view: the increments land in the ctypes array)
        mask = getattr(has, "_mask", None)
        if mask is None:
            mask = np.asarray(has, dtype=bool)
        steps[mask] += 1
        out = np.zeros((len(self._params), 2), dtype=np.float32)
        for t in np.unique(steps[mask]):                          # one value in a normal run: every parameter has taken every step
            t = int(t)
            sel = mask & (steps == t)
            out[sel, 0] = lr / (1.0 - math.pow(b1, t))
            out[sel, 1] = 1.0 / math.sqrt(1.0 - math.pow(b2, t))
        self._steps_dirty = True
        return torch.from_numpy(out.reshape(-1))

    def after_replay(self):
        """The parameters changed behind autograd's back (a replayed graph): bump their version counters."""
        torch._C._increment_version(self._params)

    def _sync_steps(self):
        if getattr(self, "_steps_dirty", False):
            for i, p in enumerate(self._params):
                self.state[p]["step"] = torch.tensor(float(self._steps[i]))
            self._steps_dirty = False

    def state_dict(self):
        self._sync_steps()
        return super().state_dict()

    def _realias(self):
        """The kernel updates `flat_param`; the module reads its parameters.  They are the same memory as long as nobody rebinds
        a parameter's storage (model.to(), load_state_dict(assign=True), deepcopy ...): if that happened, adopt the new values
        and point the parameter back into the flat buffer instead of silently training a buffer the model no longer reads."""
        base = self.flat_param.data_ptr()
        for i, p in enumerate(self._params):
            off, n = self._offsets[i], p.numel()
            if p.data_ptr() != base + 4 * off:
                if p.device != self.flat_param.device or p.dtype != torch.float32:
                    raise RuntimeError("FlatAdam: a parameter was moved off the optimizer's device / dtype after construction")
                view = self.flat_param[off:off + n].view(p.shape)
                view.copy_(p.data)
                p.data = view

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._steps_dirty = False
        for i, p in enumerate(self._params):
            st = self.state[p]
            off, n = self._offsets[i], p.numel()
            for key, flat in (("exp_avg", self.exp_avg), ("exp_avg_sq", self.exp_avg_sq)):
                view = flat[off:off + n].view(p.shape)
                view.copy_(st[key])
                st[key] = view
            self._steps[i] = int(float(st["step"]))
