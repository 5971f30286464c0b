"""Pins the CPU oracle and the host-side maths against the only reference code that is runnable in the build container:
igs/utils/sh_utils.py, igs/utils/graphics_utils.py and the torch-only RaDe-GS helpers loss_utils.py / general_utils.py /
image_utils.py (fixtures made by tests/golden/make_golden.py; data only)."""
import math

import numpy as np
import torch

from igs_amd import camera
from oracle import c_oracle as co


def test_sh_basis_matches_reference_eval_sh(golden):
    co.set_precision("float32")
    means, campos, sh = golden["sh_means"], golden["sh_campos"], golden["sh_coeffs"]
    for deg in range(4):
        want = golden["sh_rgb_deg%d" % deg]
        for i in range(means.shape[0]):
            rgb, clamped = co.sh_to_rgb(deg, sh[i], means[i], campos)
            np.testing.assert_allclose(rgb, np.maximum(want[i], 0.0), atol=2e-6)
            assert (clamped == (want[i] < 0)).all() or np.abs(want[i]).min() < 1e-6


def test_projection_matrix_matches_reference(golden):
    for (fx, fy), want in zip(golden["proj_fovs"], golden["proj_mats"]):
        got = camera.get_projection_matrix(0.01, 100.0, float(fx), float(fy)).numpy()
        np.testing.assert_allclose(got, want, rtol=0, atol=0)


def test_fov_focal_helpers(golden):
    np.testing.assert_allclose([camera.fov2focal(1.2, 1352), camera.fov2focal(0.9, 1014)], golden["fov2focal"], rtol=1e-12)
    np.testing.assert_allclose([camera.focal2fov(730.0, 1352), camera.focal2fov(730.0, 1014)], golden["focal2fov"], rtol=1e-12)


def test_camera_view_matrix_convention(golden):
    """Camera.world_view_transform = w2c.T, same matrix getWorld2View2(R, t) builds (R = c2w rotation, t = w2c translation)."""
    R, t = golden["w2v_R"], golden["w2v_t"]
    w2c = torch.tensor(golden["w2v_mat"], dtype=torch.float32)
    c2w = torch.inverse(w2c)
    cam = camera.Camera.from_c2w(c2w, (1.0, 0.8), (64, 96))
    np.testing.assert_allclose(cam.world_view_transform.numpy(), golden["w2v_mat"].T, atol=1e-5)
    np.testing.assert_allclose(cam.world_view_transform.numpy()[:3, :3], R, atol=1e-5)   # rows of w2c.T[:3,:3] = R
    np.testing.assert_allclose(cam.camera_center.numpy(), c2w[:3, 3].numpy(), atol=1e-5)
    assert cam.height == 64 and cam.width == 96
    # full projection = view_T @ proj_T
    P = camera.get_projection_matrix(0.01, 100.0, 1.0, 0.8).t()
    np.testing.assert_allclose(cam.full_proj_transform.numpy(), (cam.world_view_transform @ P).numpy(), atol=1e-6)


def test_cov3d_matches_reference_build_covariance(golden_torch_only):
    """computeCov3D (forward.cu:270-304) as restated by the oracle == strip_symmetric(L L^T), L = build_scaling_rotation(mod * s, q)
    of the reference's own python (RaDe-GS utils/general_utils.py:66-112, what gaussian_model.py's covariance activation calls)."""
    g = golden_torch_only
    q, s = g["rot_q"], g["rot_scales"]
    qn = q / np.linalg.norm(q, axis=1, keepdims=True)          # the kernel does not normalise (forward.cu:279); build_rotation does
    P = q.shape[0]
    means = np.zeros((P, 3), np.float32); means[:, 2] = 3.0
    eye = np.eye(4, dtype=np.float32)
    proj = camera.get_projection_matrix(0.01, 100.0, 1.0, 1.0).t().numpy()
    co.set_precision("float32")
    for mod, key in ((1.0, "rot_cov6"), (1.7, "rot_cov6_mod17")):
        nr, out, st = co.rasterize_forward(np.zeros(3, np.float32), means, None, np.full((P, 1), 0.5, np.float32), s, qn.astype(np.float32), mod, None,
                                           eye, proj, 0.5463, 0.5463, 0.0, 32, 32, np.zeros((P, 16, 3), np.float32), 3, np.zeros(3, np.float32))
        cov = st.intermediates()["cov3D"]
        np.testing.assert_allclose(cov, g[key], rtol=2e-5, atol=1e-7 * float(np.abs(g[key]).max()))


def test_densify_rotation_and_logit_helpers_match_reference(golden_torch_only):
    from igs_amd.densify import build_rotation
    g = golden_torch_only
    R = build_rotation(torch.from_numpy(g["rot_q"])).numpy()
    np.testing.assert_allclose(R, g["rot_R"], rtol=0, atol=2e-6)
    x = torch.from_numpy(g["inv_sigmoid_x"])
    np.testing.assert_allclose(torch.log(x / (1 - x)).numpy(), g["inv_sigmoid_y"], rtol=1e-6)


def test_loss_restatements_match_reference_loss_utils(golden_torch_only):
    """The PyTorch restatements the GPU loss kernels are compared with elsewhere (oracle/torch_losses.py: l1_mean,
    ssim_reference_call / ssim_mean, psnr) against values and autograd gradients produced by the reference's own loss_utils.py /
    image_utils.py -- so the chain  HIP kernel == restatement == reference  is closed by data, not by reading."""
    from oracle import torch_losses as tl
    from igs_amd import refine
    g = golden_torch_only
    for tag in ("a", "b"):
        img, gt = torch.from_numpy(g["loss_%s_img" % tag]), torch.from_numpy(g["loss_%s_gt" % tag])
        x = img.clone().requires_grad_(True)
        l1 = tl.l1_mean(x, gt)
        s_call = tl.ssim_reference_call(x, gt.unsqueeze(0), size_average=False)
        np.testing.assert_allclose(l1.item(), g["loss_%s_l1" % tag], rtol=1e-6)
        np.testing.assert_allclose(s_call.detach().numpy(), g["loss_%s_ssim_call" % tag], rtol=1e-5)
        loss = 0.8 * l1 + 0.2 * (1.0 - s_call)
        loss.sum().backward()
        np.testing.assert_allclose(loss.detach().numpy(), g["loss_%s_total" % tag], rtol=1e-5)
        G = g["loss_%s_grad" % tag]
        np.testing.assert_allclose(x.grad.numpy(), G, rtol=0, atol=2e-5 * float(np.abs(G).max()))
        m, ssim_map = tl.ssim_reference_call(img, gt, size_average=True)
        np.testing.assert_allclose(m.item(), g["loss_%s_ssim" % tag], rtol=1e-5)
        np.testing.assert_allclose(tl.ssim_mean(img, gt).item(), g["loss_%s_ssim" % tag], rtol=1e-5)
        # infer_batch.py:350-353's PSNR == image_utils.psnr when the image is already inside [0, 1]
        np.testing.assert_allclose(refine.psnr(img, gt).item(), g["psnr_%s" % tag].item(), rtol=1e-5)
        np.testing.assert_allclose(tl.psnr(img, gt).item(), g["psnr_%s" % tag].item(), rtol=1e-5)
