"""`Adam`: a `torch.optim.Optimizer` with torch.optim.Adam's update rule (no weight decay, no amsgrad) that runs ALL parameter
groups of the refine-time Gaussian model in ONE HIP launch (`igs_adam_step_multi`, include/igs_rast.h).

The reference builds `torch.optim.Adam(l, lr=0.0, eps=1e-15)` over five groups (igs/models/gaussian_model.py:303-348); PyTorch's
implementation walks them with a dozen elementwise multi-tensor kernels (~0.6 ms per step over 11.8 M parameters on this GPU,
more than the whole render + backward).  A caller replaces one constructor:

    from igs_amd.optim import Adam
    self.optimizer = Adam(l, lr=0.0, eps=1e-15)

State layout and names are torch.optim.Adam's (`state[p] = {"step", "exp_avg", "exp_avg_sq"}`), so `state_dict()` / the densification
code that edits `optimizer.state` (gaussian_model.py:466-557) keep working.  There is no CPU path: parameters must be float32 GPU tensors.
"""
import math

import torch

from . import _cabi


class Adam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        if lr < 0.0 or eps < 0.0 or not (0.0 <= betas[0] < 1.0) or not (0.0 <= betas[1] < 1.0):
            raise ValueError("igs_amd.optim.Adam: invalid hyper-parameter")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        Cx = _cabi.ext()
        # batches of up to 8 tensors that share (device, betas, eps): one launch each
        batches = {}
        for group in self.param_groups:
            b1, b2 = group["betas"]
            for p in group["params"]:
                if p.grad is None:
                    continue
                if not p.is_cuda or p.dtype != torch.float32 or p.grad.is_sparse:
                    raise RuntimeError("igs_amd.optim.Adam: parameters must be dense float32 GPU tensors (no CPU fallback)")
                st = self.state[p]
                if len(st) == 0:
                    st["step"] = 0
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["step"] = int(st["step"]) + 1
                t = st["step"]
                g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                if not (p.is_contiguous() and st["exp_avg"].is_contiguous() and st["exp_avg_sq"].is_contiguous()):
                    raise RuntimeError("igs_amd.optim.Adam: parameters and their state must be contiguous")
                batches.setdefault((p.device, b1, b2, group["eps"]), []).append(
                    (p, g, st["exp_avg"], st["exp_avg_sq"], group["lr"], 1.0 - b1 ** t, math.sqrt(1.0 - b2 ** t)))
        for (dev, b1, b2, eps), items in batches.items():
            for i in range(0, len(items), 8):
                chunk = items[i:i + 8]
                col = lambda j: [c[j] for c in chunk]
                Cx.adam_step_multi(col(0), col(1), col(2), col(3), col(4), col(5), col(6), b1, b2, eps)
        return loss
