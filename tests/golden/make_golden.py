"""Generates tests/golden/ref_helpers.npz from the reference's own importable pure-python helpers.

Run in the build container only (needs /root/reference):  python tests/golden/make_golden.py
The reference has no tests/fixtures for the rasterizer path; the only reference code that
can be executed here are igs/utils/sh_utils.py (eval_sh: the SH basis and sign conventions that
computeColorFromSH, forward.cu:23-74, must reproduce) and igs/utils/graphics_utils.py
(getProjectionMatrix, getWorld2View2, fov2focal, focal2fov: the matrix conventions of the callers).
They are loaded by file path (the `igs` package itself cannot be imported: jaxtyping/omegaconf absent).
Only inputs and outputs (data) are stored, never reference source.

Round 2 adds tests/golden/ref_torch_only.npz from the RaDe-GS python files that import with torch alone
(census: loss_utils, general_utils, image_utils, sh_utils, depth_utils import; graphics_utils -- home of
depth_double_to_normal -- needs cv2, igs/utils/loss_utils.py needs icecream: both stay unpinned):
  * submodules/RaDe-GS/utils/loss_utils.py  l1_loss, ssim (the same functions as igs/utils/loss_utils.py:17-63): values and the
    autograd gradient of 0.8 L1 + 0.2 (1 - SSIM) w.r.t. the rendered image, on two image sizes;
  * submodules/RaDe-GS/utils/general_utils.py  build_rotation, build_scaling_rotation + strip_symmetric (the 3-D covariance of
    gaussian_model.py's build_covariance_from_scaling_rotation = what computeCov3D, forward.cu:270-304, must produce), inverse_sigmoid;
  * submodules/RaDe-GS/utils/image_utils.py  psnr.
"""
import importlib.util
import math
import os

import numpy as np
import torch

REF = "/root/reference/igs/utils"
HERE = os.path.dirname(os.path.abspath(__file__))


def load(name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, name + ".py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


RADE = "/root/reference/submodules/RaDe-GS/utils"


def load_rade(name):
    spec = importlib.util.spec_from_file_location("rade_" + name, os.path.join(RADE, name + ".py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def torch_only():
    lu, gu, iu = load_rade("loss_utils"), load_rade("general_utils"), load_rade("image_utils")
    out = {}
    g = torch.Generator().manual_seed(4321)
    for tag, (H, W) in (("a", (40, 52)), ("b", (67, 35))):
        yy, xx = torch.meshgrid(torch.arange(H).float(), torch.arange(W).float(), indexing="ij")
        gt = torch.stack([0.5 + 0.4 * torch.sin(xx / 5.0 + c) * torch.cos(yy / (7.0 + c)) for c in range(3)]) \
            + 0.05 * torch.randn(3, H, W, generator=g)
        gt = gt.clamp(0, 1)
        img = (gt + 0.08 * torch.randn(3, H, W, generator=g)).clamp(0, 1)
        x = img.clone().requires_grad_(True)
        l1 = lu.l1_loss(x, gt)
        s_avg = lu.ssim(x, gt)                                           # size_average=True
        s_call = lu.ssim(x, gt.unsqueeze(0), size_average=False)         # the call shape of infer_batch.py:302
        loss = 0.8 * l1 + 0.2 * (1.0 - s_call)
        loss.sum().backward()
        gx = torch.autograd.grad(lu.l1_loss(x, gt), x)[0]
        gs = torch.autograd.grad(lu.ssim(x, gt), x)[0]
        out["loss_%s_img" % tag], out["loss_%s_gt" % tag] = img.numpy(), gt.numpy()
        out["loss_%s_l1" % tag], out["loss_%s_ssim" % tag] = l1.detach().numpy(), s_avg.detach().numpy()
        out["loss_%s_ssim_call" % tag] = s_call.detach().numpy()
        out["loss_%s_total" % tag], out["loss_%s_grad" % tag] = loss.detach().numpy(), x.grad.numpy()
        out["loss_%s_grad_l1" % tag], out["loss_%s_grad_ssim" % tag] = gx.numpy(), gs.numpy()
        out["psnr_%s" % tag] = iu.psnr(img.unsqueeze(0), gt.unsqueeze(0)).numpy()
    N = 48
    q = torch.randn(N, 4, generator=g)
    s = torch.exp(torch.rand(N, 3, generator=g) * 3.0 - 3.5)
    out["rot_q"], out["rot_scales"] = q.numpy(), s.numpy()
    orig_zeros = torch.zeros
    torch.zeros = lambda *a, **k: orig_zeros(*a, **{kk: vv for kk, vv in k.items() if kk != "device"})     # build_rotation says device='cuda'
    try:
        out["rot_R"] = gu.build_rotation(q).numpy()
        L = gu.build_scaling_rotation(1.0 * s, q)
        out["rot_cov6"] = gu.strip_symmetric(L @ L.transpose(1, 2)).numpy()
        L2 = gu.build_scaling_rotation(1.7 * s, q)
        out["rot_cov6_mod17"] = gu.strip_symmetric(L2 @ L2.transpose(1, 2)).numpy()
    finally:
        torch.zeros = orig_zeros
    out["inv_sigmoid_x"] = np.array([0.01, 0.1, 0.5, 0.9, 0.995], dtype=np.float32)
    out["inv_sigmoid_y"] = gu.inverse_sigmoid(torch.tensor(out["inv_sigmoid_x"])).numpy()
    np.savez(os.path.join(HERE, "ref_torch_only.npz"), **out)
    print("wrote ref_torch_only.npz:", {k: v.shape for k, v in out.items()})


def main():
    torch_only()
    sh_utils = load("sh_utils")
    gu = load("graphics_utils")
    g = torch.Generator().manual_seed(1234)
    N = 64
    means = torch.randn(N, 3, generator=g) * 2.0
    campos = torch.tensor([0.3, -0.2, -4.0])
    dirs = means - campos
    dirs = dirs / dirs.norm(dim=1, keepdim=True)
    sh = torch.randn(N, 16, 3, generator=g) * 0.5
    out = {"sh_means": means.numpy(), "sh_campos": campos.numpy(), "sh_coeffs": sh.numpy()}
    for deg in range(4):
        # eval_sh wants [..., C, (deg+1)^2]
        rgb = sh_utils.eval_sh(deg, sh[:, :(deg + 1) ** 2, :].transpose(1, 2), dirs) + 0.5
        out["sh_rgb_deg%d" % deg] = rgb.numpy()
    out["rgb2sh"] = sh_utils.RGB2SH(torch.tensor([0.0, 0.25, 1.0])).numpy()
    fovs = [(math.radians(50.0), math.radians(50.0)), (1.4944, 1.2138), (0.6, 0.9)]
    out["proj_fovs"] = np.array(fovs, dtype=np.float64)
    out["proj_mats"] = np.stack([gu.getProjectionMatrix(0.01, 100.0, fx, fy).numpy() for fx, fy in fovs])
    R = np.array([[0.9, -0.1, 0.42], [0.2, 0.95, -0.2], [-0.38, 0.27, 0.88]], dtype=np.float64)
    q, _ = np.linalg.qr(R)
    t = np.array([0.5, -1.0, 3.0])
    out["w2v_R"], out["w2v_t"] = q, t
    out["w2v_mat"] = gu.getWorld2View2(q, t)
    out["w2v_mat_ts"] = gu.getWorld2View2(q, t, np.array([0.1, 0.2, -0.3]), 2.0)
    out["fov2focal"] = np.array([gu.fov2focal(1.2, 1352), gu.fov2focal(0.9, 1014)])
    out["focal2fov"] = np.array([gu.focal2fov(730.0, 1352), gu.focal2fov(730.0, 1014)])
    np.savez(os.path.join(HERE, "ref_helpers.npz"), **out)
    print("wrote ref_helpers.npz:", {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
