"""`Adam`: a `torch.optim.Optimizer` with torch.optim.Adam's update rule (no weight decay, no amsgrad) that runs ALL parameter
groups of the refine-time Gaussian model in ONE HIP launch (`igs_adam_step_multi`, include/igs_rast.h).

The reference builds `torch.optim.Adam(l, lr=0.0, eps=1e-15)` over five groups (igs/models/gaussian_model.py:303-348); PyTorch's
implementation walks them with a dozen elementwise multi-tensor kernels (~0.6 ms per step over 11.8 M parameters on this GPU,
more than the whole render + backward).  A caller replaces one constructor:

    from igs_amd.optim import Adam
    self.optimizer = Adam(l, lr=0.0, eps=1e-15)

State layout and names are torch.optim.Adam's (`state[p] = {"step", "exp_avg", "exp_avg_sq"}`), so `state_dict()` / the densification
code that edits `optimizer.state` (gaussian_model.py:466-557) keep working.  There is no CPU path: parameters must be float32 GPU tensors.

`capturable=True` (torch.optim.Adam's name for the same thing): `state[p]["step"]` is a float32 scalar ON THE GPU, advanced by the
launch itself (`igs_adam_step_multi_dev`: still ONE launch -- the workgroup that finishes last advances the counts), so `step()` reads nothing from the host that changes between steps and a whole refine
iteration -- render, loss, backward, step -- can be captured into one hipGraph (`torch.cuda.graph`) and replayed.  The state must
exist before the capture (one ordinary step on a side stream, as for any captured PyTorch optimizer, or `init_state()`).
"""
import math

import torch

from . import _cabi


class Adam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, capturable=False):
        if lr < 0.0 or eps < 0.0 or not (0.0 <= betas[0] < 1.0) or not (0.0 <= betas[1] < 1.0):
            raise ValueError("igs_amd.optim.Adam: invalid hyper-parameter")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))
        self.capturable = bool(capturable)
        self._done = {}                 # capturable: per device, the self-resetting done-counters of the one-launch update
        self._ext = _cabi.ext()
        # torch.optim.Optimizer wraps the class's `step` in a profiler / hook trampoline that costs ~35 us of host time per call -- more than
        # the launch it guards, in a loop whose iteration is bound by host time (tools/profile_dropin_loop_host.py).  Instances call the
        # plain method unless somebody has registered step hooks.
        self._wrapped_step = super(Adam, self).__getattribute__("step")
        self.step = self._step_dispatch

    def _step_dispatch(self, closure=None):
        if self._optimizer_step_pre_hooks or self._optimizer_step_post_hooks:
            return self._wrapped_step(closure)
        return self._step_impl(closure)

    def zero_grad(self, set_to_none=True):
        """torch.optim.Optimizer.zero_grad without its per-call bookkeeping (profiler range, foreach grouping): ~2 us instead of ~20."""
        for group in self.param_groups:
            for p in group["params"]:
                if p.grad is not None:
                    if set_to_none:
                        p.grad = None
                    else:
                        p.grad.detach_().zero_()

    def step(self, closure=None):
        return self._step_impl(closure)

    def _init_state(self, p):
        if not p.is_cuda or p.dtype != torch.float32 or not p.is_contiguous():
            raise RuntimeError("igs_amd.optim.Adam: parameters must be dense, contiguous float32 GPU tensors (no CPU fallback)")
        st = self.state[p]
        st["step"] = torch.zeros((), dtype=torch.float32, device=p.device) if self.capturable else 0
        st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
        st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
        return st

    def init_state(self):
        """Create the state of every parameter now (what the first step() would do): call it before capturing a step into a graph."""
        for group in self.param_groups:
            for p in group["params"]:
                if len(self.state[p]) == 0:
                    self._init_state(p)
                if self.capturable and p.device not in self._done:
                    self._done[p.device] = torch.zeros(self._ext.adam_dev_scratch_words(), dtype=torch.int32, device=p.device)

    @torch.no_grad()
    def _step_impl(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        Cx = self._ext
        # batches of up to 8 tensors that share (device, betas, eps): one launch each
        batches = {}
        for group in self.param_groups:
            b1, b2 = group["betas"]
            lr, eps = group["lr"], group["eps"]
            for p in group["params"]:
                g = p.grad
                if g is None:
                    continue
                st = self.state[p]
                if len(st) == 0:
                    if g.is_sparse:
                        raise RuntimeError("igs_amd.optim.Adam: parameters must be dense, contiguous float32 GPU tensors (no CPU fallback)")
                    if p.is_cuda and torch.cuda.is_current_stream_capturing():
                        raise RuntimeError("igs_amd.optim.Adam: the optimizer state must exist before a step is captured (init_state())")
                    st = self._init_state(p)
                key = (p.device, b1, b2, eps)
                b = batches.get(key)
                if b is None:
                    b = batches[key] = ([], [], [], [], [], [], [], [])
                b[0].append(p); b[1].append(g); b[2].append(st["exp_avg"]); b[3].append(st["exp_avg_sq"]); b[4].append(lr)
                if self.capturable:
                    if not torch.is_tensor(st["step"]):          # (state loaded from a non-capturable optimizer)
                        st["step"] = torch.full((), float(st["step"]), dtype=torch.float32, device=p.device)
                    b[7].append(st["step"])
                else:
                    t = st["step"] = int(st["step"]) + 1
                    b[5].append(1.0 - b1 ** t); b[6].append(math.sqrt(1.0 - b2 ** t))
        for (dev, b1, b2, eps), b in batches.items():
            done = None
            if self.capturable:
                done = self._done.get(dev)
                if done is None:
                    if torch.cuda.is_current_stream_capturing():
                        raise RuntimeError("igs_amd.optim.Adam: the optimizer state must exist before a step is captured (init_state())")
                    done = self._done[dev] = torch.zeros(Cx.adam_dev_scratch_words(), dtype=torch.int32, device=dev)
            for i in range(0, len(b[0]), 8):
                Cx.adam_step_multi(b[0][i:i + 8], b[1][i:i + 8], b[2][i:i + 8], b[3][i:i + 8], b[4][i:i + 8], b[5][i:i + 8], b[6][i:i + 8],
                                   b1, b2, eps, b[7][i:i + 8], done)
        return loss
