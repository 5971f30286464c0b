"""The UNCHANGED caller: one refine iteration exactly as `infer_batch.py:279-324` drives the rasterizer package -- used by bench.py's
`dropin` leg, tools/trace_dropin_loop.sh / dropin_bench.py / profile_dropin_loop_host.py and the GPU tests; caller-side code, not part of the product package.

What is kept from the reference's loop, statement for statement in behaviour (not in text):
  * the model's parameters are five `nn.Parameter`s (xyz, shs, opacity logit, log-scale, rotation) in a `torch.optim.Adam(l, lr=0.0,
    eps=1e-15)` with one group each (`gaussian_model.py:303-348`), activations are separate PyTorch ops in the property getters
    (`:90-127`: sigmoid / exp / F.normalize);
  * `forward_single_view` (`infer_batch.py:39-124`): a fresh zeros+0 `screenspace_points` with `retain_grad()`, a fresh
    `GaussianRasterizationSettings` / `GaussianRasterizer` per call, keyword arguments, the eight outputs;
  * the per-iteration PSNR of the rendered image (`:300`), `l1_loss`, `ssim(render, gt.unsqueeze(0), size_average=False)`,
    `loss.backward()`, `optimizer.step()`, `optimizer.zero_grad(set_to_none=True)` (`:301-324`).
Not reproduced: building the Camera from a c2w matrix per iteration (`Camera.from_c2w`, `:296` -- caller-side matrix algebra on the
host, nothing the package under test sees) and moving the image to the GPU (`:290-291`); cameras and images are resident.

`optimizer`: "torch" = torch.optim.Adam as the reference constructs it; "fused" = igs_amd.optim.Adam, the one-line replacement
that runs all groups in one HIP launch (same update rule); "fused_capturable" / "torch_capturable" = either with its step counts on the
GPU, for a loop body replayed from hipGraphs (igs_amd/graphs.py).
`losses`: "igs" = `from igs_amd.losses import l1_loss, ssim` (the one-line import change of INTEGRATION.md); "torch" = plain
PyTorch l1 (abs / mean) and whatever SSIM function the caller passes in.
"""
import math

import torch
import torch.nn as nn


class CallerModel:
    """Refine-time parameter container in the shape of `GaussianModel.load_fromstream` (gaussian_model.py:265-348)."""

    def __init__(self, raw, device, lrs, optimizer="torch", fused_activations=False):
        self.fused_activations = fused_activations      # a caller who changes the three activation properties to igs_amd.activations.activate
        mk = lambda t: nn.Parameter(t.detach().clone().to(device).contiguous().requires_grad_(True))
        self._xyz, self._shs = mk(raw["xyz"]), mk(raw["shs"])
        self._opacity, self._scaling, self._rotation = mk(raw["opacity"]), mk(raw["scaling"]), mk(raw["rotation"])
        l = [{"params": [self._xyz], "lr": lrs["xyz"], "name": "xyz"},
             {"params": [self._rotation], "lr": lrs["rotation"], "name": "rotation"},
             {"params": [self._shs], "lr": lrs["shs"], "name": "shs"},
             {"params": [self._opacity], "lr": lrs["opacity"], "name": "opacity"},
             {"params": [self._scaling], "lr": lrs["scaling"], "name": "scaling"}]
        if optimizer == "torch":
            self.optimizer = torch.optim.Adam(l, lr=0.0, eps=1e-15)
        elif optimizer == "torch_fused":
            self.optimizer = torch.optim.Adam(l, lr=0.0, eps=1e-15, fused=True)
        elif optimizer == "torch_capturable":
            self.optimizer = torch.optim.Adam(l, lr=0.0, eps=1e-15, capturable=True)
        else:
            from igs_amd.optim import Adam
            self.optimizer = Adam(l, lr=0.0, eps=1e-15, capturable=(optimizer == "fused_capturable"))

    get_xyz = property(lambda self: self._xyz)
    get_features = property(lambda self: self._shs)
    def _act(self, i):
        if not self.fused_activations:          # gaussian_model.py:90-127: every getter applies ITS activation only
            if i == 0:
                return torch.sigmoid(self._opacity)
            if i == 1:
                return torch.exp(self._scaling)
            return torch.nn.functional.normalize(self._rotation)
        cur = getattr(self, "_act_cache", None)          # one fused launch serves the three getters of an iteration
        key = (self._opacity._version, self._scaling._version, self._rotation._version)
        if cur is None or cur[0] != key or cur[2] >= 3:
            from igs_amd.activations import activate
            cur = self._act_cache = [key, activate(self._opacity, self._scaling, self._rotation), 0]
        cur[2] += 1
        return cur[1][i]

    get_opacity = property(lambda self: self._act(0))
    get_scaling = property(lambda self: self._act(1))
    get_rotation = property(lambda self: self._act(2))

    def raw(self):
        return dict(xyz=self._xyz.detach(), shs=self._shs.detach(), opacity=self._opacity.detach(), scaling=self._scaling.detach(),
                    rotation=self._rotation.detach())


def forward_single_view(gs, cam, bg, sh_degree=3, package=None):
    """The operations of `infer_batch.py:39-124`, in its order, against `package` (default: diff_gaussian_rasterization_rade):
    a zeros + 0 gradient sink for the screen-space means with retain_grad(), a settings tuple built by keyword, a fresh rasterizer module,
    keyword call with SHs (no precomputed colours / covariances), the visibility mask from the radii."""
    if package is None:
        import diff_gaussian_rasterization_rade as package
    sink = torch.zeros_like(gs.get_xyz, dtype=gs.get_xyz.dtype, requires_grad=True, device="cuda") + 0
    try:
        sink.retain_grad()
    except Exception:  # noqa: BLE001
        pass
    cfg = dict(image_height=int(cam.height), image_width=int(cam.width), tanfovx=math.tan(cam.FoVx * 0.5), tanfovy=math.tan(cam.FoVy * 0.5),
               bg=bg, scale_modifier=1.0, viewmatrix=cam.world_view_transform, projmatrix=cam.full_proj_transform.float(),
               sh_degree=sh_degree, campos=cam.camera_center, prefiltered=False, debug=False,
               kernel_size=0.0, require_coord=True, require_depth=True)          # (the RaDe-GS defaults the reference passes)
    render = package.GaussianRasterizer(raster_settings=package.GaussianRasterizationSettings(**cfg))
    outs = render(means3D=gs.get_xyz, means2D=sink.contiguous().float(), shs=gs.get_features, colors_precomp=None,
                  opacities=gs.get_opacity, scales=gs.get_scaling, rotations=gs.get_rotation, cov3D_precomp=None)
    image, radii, coord, mcoord, depth, mdepth, alpha, normal = outs
    return {"images_pred": image, "bg_color": bg, "depth_pred": depth, "radii": radii, "visibility_filter": radii > 0,
            "viewspace_points": sink, "mdepth": mdepth, "normal": normal, "alpha": alpha, "coord": coord, "mcoord": mcoord}


def make_losses(kind, ssim_fn=None):
    """(l1_loss, ssim) pair: "igs" = igs_amd.losses (the one-line import change); "torch" = PyTorch's own abs / mean for L1 and
    `ssim_fn` (a restatement of loss_utils.py:34-63 the caller supplies -- tests pass the one in oracle/torch_losses.py) for SSIM."""
    if kind == "igs":
        from igs_amd.losses import l1_loss, ssim
        return l1_loss, ssim
    return (lambda a, b: torch.abs((a - b)).mean()), ssim_fn


def refine_iteration(gs, cam, gt_image, bg, loss="l1_ssim", lambda_l1=0.8, losses=None, package=None, psnr_line=True):
    """One pass of the loop body `infer_batch.py:296-324` (densification off).  Returns the render package and the loss tensor."""
    l1_loss, ssim = losses
    pkg = forward_single_view(gs, cam, bg, sh_degree=3, package=package)
    render_image = pkg["images_pred"]
    if psnr_line:
        pkg["psnr"] = -10 * torch.log10(torch.mean((render_image.detach() - gt_image) ** 2))
    Ll1 = l1_loss(render_image, gt_image)
    if loss == "l1":
        total = Ll1                                  # BASELINE configs[2]: L1 only
    else:
        total = lambda_l1 * Ll1 + (1 - lambda_l1) * (1.0 - ssim(render_image, gt_image.unsqueeze(0), size_average=False))
    total.backward()
    with torch.no_grad():
        gs.optimizer.step()
        gs.optimizer.zero_grad(set_to_none=True)
    return pkg, total
