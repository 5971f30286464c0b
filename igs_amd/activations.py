"""Fused activations of the refine-time Gaussian model as ONE autograd Function (one HIP launch forward, one backward).

The reference applies them with separate PyTorch ops outside the rasterizer (`igs/models/gaussian_model.py:90-127`:
`get_opacity = sigmoid(_opacity)`, `get_scaling = exp(_scaling)`, `get_rotation = F.normalize(_rotation)`), which costs about a dozen
tiny elementwise / reduction kernels per step forward and backward -- on this GPU more host time than device time
(INTEGRATION.md).  A caller who is willing to change those three property bodies gets the same values and gradients from
`igs_activate_fwd` / `igs_activate_bwd` (include/igs_rast.h):

    opacity, scaling, rotation = igs_amd.activations.activate(self._opacity, self._scaling, self._rotation)
"""
import torch

from . import _cabi


class _Activate(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logit, log_scale, rot):
        if not (logit.is_cuda and log_scale.is_cuda and rot.is_cuda):
            raise RuntimeError("igs_amd.activations: tensors must be on a GPU (no CPU fallback)")
        L = _cabi.lib()
        dev = logit.device
        P = rot.shape[0]
        lg, ls, rt = logit.contiguous().float(), log_scale.contiguous().float(), rot.contiguous().float()
        opacity, scale, rot_n = torch.empty_like(lg), torch.empty_like(ls), torch.empty_like(rt)
        with torch.cuda.device(dev):
            rc = L.igs_activate_fwd(torch.cuda.current_stream(dev).cuda_stream, P, lg.data_ptr(), ls.data_ptr(), rt.data_ptr(),
                                    opacity.data_ptr(), scale.data_ptr(), rot_n.data_ptr())
        if rc != 0:
            raise RuntimeError("igs_activate_fwd failed: %d" % rc)
        ctx.save_for_backward(opacity, scale, rt)
        return opacity, scale, rot_n

    @staticmethod
    def backward(ctx, d_opacity, d_scale, d_rot):
        opacity, scale, rt = ctx.saved_tensors
        L = _cabi.lib()
        dev = opacity.device
        P = rt.shape[0]
        z = torch.zeros_like
        d_opacity = z(opacity) if d_opacity is None else d_opacity.contiguous()
        d_scale = z(scale) if d_scale is None else d_scale.contiguous()
        d_rot = z(rt) if d_rot is None else d_rot.contiguous()
        g_logit, g_log_scale, g_rot = torch.empty_like(opacity), torch.empty_like(scale), torch.empty_like(rt)
        with torch.cuda.device(dev):
            rc = L.igs_activate_bwd(torch.cuda.current_stream(dev).cuda_stream, P, opacity.data_ptr(), scale.data_ptr(), rt.data_ptr(),
                                    d_opacity.data_ptr(), d_scale.data_ptr(), d_rot.data_ptr(), g_logit.data_ptr(), g_log_scale.data_ptr(),
                                    g_rot.data_ptr())
        if rc != 0:
            raise RuntimeError("igs_activate_bwd failed: %d" % rc)
        return g_logit, g_log_scale, g_rot


def activate(opacity_logit, log_scale, rotation):
    """(sigmoid(opacity_logit) [P,1], exp(log_scale) [P,3], F.normalize(rotation) [P,4]) with their gradients, fused."""
    return _Activate.apply(opacity_logit, log_scale, rotation)
